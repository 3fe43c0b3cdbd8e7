// simd_probe.hip -- which SIMD does each wave of a 512-thread workgroup land on?  (measurement helper for the paired-wave
// kernels: wave w and wave w + 4 are meant to share a SIMD.)  Build: hipcc --offload-arch=gfx950 -O2 simd_probe.hip -o simd_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>

__global__ void __launch_bounds__(512) probe(uint32_t *out, int spin)
{
  extern __shared__ uint32_t lds[];
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); // HW_REG_HW_ID
  lds[threadIdx.x] = hw;
  uint32_t x = hw;
  for (int i = 0; i < spin; i++)
    x = x * 1664525u + 1013904223u;
  if ((threadIdx.x & 63u) == 0)
    out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw | (x == 12345u ? 1u << 31 : 0u);
}

int main(int argc, char **argv)
{
  const int blocks = argc > 1 ? atoi(argv[1]) : 512;
  const size_t lds_bytes = argc > 2 ? (size_t)atoi(argv[2]) : 155 * 1024;
  uint32_t *d;
  hipMalloc(&d, blocks * 8 * sizeof(uint32_t));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), lds_bytes, 0, d, 20000);
  if (hipDeviceSynchronize() != hipSuccess)
  {
    printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
    return 1;
  }
  uint32_t *h = (uint32_t *)malloc(blocks * 8 * sizeof(uint32_t));
  hipMemcpy(h, d, blocks * 8 * sizeof(uint32_t), hipMemcpyDeviceToHost);
  std::map<std::string, int> patterns;
  int paired = 0;
  for (int b = 0; b < blocks; b++)
  {
    char s[64];
    int n = 0, ok = 1;
    for (int w = 0; w < 8; w++)
    {
      const uint32_t simd = (h[b * 8 + w] >> 4) & 3u;
      n += snprintf(s + n, sizeof(s) - n, "%u", simd);
      if (w >= 4 && simd != ((h[b * 8 + w - 4] >> 4) & 3u))
        ok = 0;
    }
    patterns[s]++;
    paired += ok;
  }
  printf("blocks %d, lds %zu: wave w and w+4 on the same SIMD in %d blocks\n", blocks, lds_bytes, paired);
  for (auto &p : patterns)
    printf("  simd of waves 0..7 = %s : %d blocks\n", p.first.c_str(), p.second);
  for (int w = 0; w < 8; w++)
    printf("  block 0 wave %d: hw_id %08x (wave_id %u simd %u cu %u se %u)\n", w, h[w], h[w] & 15u, (h[w] >> 4) & 3u, (h[w] >> 8) & 15u, (h[w] >> 13) & 7u);
  return 0;
}
