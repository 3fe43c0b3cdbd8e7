// ubench.hip -- instruction issue-cost microbenchmark for gfx950, in the regime the DEGA coder runs in:
// integer/bit VALU work, ONE wave per SIMD (64 Ki channels = 1024 waves = 1 per SIMD) and two waves per SIMD.
// Prints ns per instruction per wave and the same in cycles at the clock measured with s_memtime/s_memrealtime.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench && tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)

constexpr int ITERS = 2000;

// DEP: one dependent chain on v0 ; IND: four independent chains v0..v3 (16 groups of 4)
#define KERNEL_DEP(name, INSTR)                                                               \
  __global__ void name(uint32_t *out, uint32_t seed)                                          \
  {                                                                                           \
    uint32_t a = threadIdx.x + seed, b = seed | 3u, c = seed + 77u;                            \
    uint64_t w = ((uint64_t)a << 32) | b, z = ((uint64_t)c << 20) | 5u;                        \
    double d = (double)a, e = 1.000001, f = 0.5;                                              \
    float g = (float)a + 1.0f;                                                                \
    for (int i = 0; i < ITERS; i++)                                                           \
    {                                                                                         \
      asm volatile(R64(INSTR "\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(w), "+v"(z), "+v"(d), "+v"(e), "+v"(f), "+v"(g)::"vcc", "s4", "s5"); \
    }                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + (uint32_t)w + (uint32_t)z + (uint32_t)d + (uint32_t)g; \
  }

// operand names: %0=a %1=b %2=c (u32) %3=w %4=z (u64) %5=d %6=e %7=f (f64) %8=g (f32)
KERNEL_DEP(k_add, "v_add_u32 %0, %0, %1")
KERNEL_DEP(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL_DEP(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL_DEP(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL_DEP(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL_DEP(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL_DEP(k_mulhi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
KERNEL_DEP(k_mad_u64, "v_mad_u64_u32 %3, vcc, %0, %1, %3")
KERNEL_DEP(k_lshl64, "v_lshlrev_b64 %3, %1, %3")
KERNEL_DEP(k_lshr64, "v_lshrrev_b64 %3, %1, %3")
KERNEL_DEP(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL_DEP(k_bfe, "v_bfe_u32 %0, %0, %1, %2")
KERNEL_DEP(k_ffbh, "v_ffbh_u32 %0, %0")
KERNEL_DEP(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL_DEP(k_lshl_or, "v_lshl_or_b32 %0, %0, %1, %2")
KERNEL_DEP(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %2")
KERNEL_DEP(k_xnor, "v_xnor_b32 %0, %0, %1")
KERNEL_DEP(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL_DEP(k_addco, "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %2, vcc, %2, %1, vcc")
KERNEL_DEP(k_cmp_cnd, "v_cmp_ne_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL_DEP(k_cmp_addc, "v_cmp_ne_u32 vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
KERNEL_DEP(k_fma64, "v_fma_f64 %5, %5, %6, %7")
KERNEL_DEP(k_mul64, "v_mul_f64 %5, %5, %6")
KERNEL_DEP(k_add64, "v_add_f64 %5, %5, %7")
KERNEL_DEP(k_cvt_f64_u32, "v_cvt_f64_u32 %5, %0\n\tv_cvt_u32_f64 %0, %5")
KERNEL_DEP(k_rcp32, "v_rcp_f32 %8, %8")
KERNEL_DEP(k_rcp64, "v_rcp_f64 %5, %5")
KERNEL_DEP(k_cvt_f32_u32, "v_cvt_f32_u32 %8, %0\n\tv_cvt_u32_f32 %0, %8")
KERNEL_DEP(k_fma32, "v_fma_f32 %8, %8, %8, %8")
KERNEL_DEP(k_readlane, "v_readfirstlane_b32 s4, %0\n\tv_add_u32 %0, s4, %1")
KERNEL_DEP(k_salu_mix, "s_add_u32 s4, s4, 1\n\tv_add_u32 %0, %0, %1")
KERNEL_DEP(k_salu2_mix, "s_add_u32 s4, s4, 1\n\ts_and_b32 s5, s5, s4\n\tv_add_u32 %0, %0, %1")
KERNEL_DEP(k_ind4_add, "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_xor_b32 %1, %1, %1\n\tv_add_u32 %0, %0, %2")
KERNEL_DEP(k_branch, "v_cmp_eq_u32 vcc, %0, %1\n\ts_cbranch_vccnz 1f\n\tv_add_u32 %0, %0, %2\n\t1:")

__global__ void k_lds_b64(uint32_t *out, uint32_t seed)
{
  __shared__ uint64_t tab[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t idx = (threadIdx.x * 977u + seed) & 16383u;
  uint32_t acc = 0;
  for (int i = 0; i < ITERS; i++)
  {
#pragma unroll
    for (int j = 0; j < 64; j++)
    {
      const uint64_t v = tab[idx];
      idx = (uint32_t)(v >> 7) & 16383u; // dependent: latency of ds_read_b64 + 2 valu
      acc += (uint32_t)v;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

__global__ void k_lds_b64_tp(uint32_t *out, uint32_t seed)
{
  __shared__ uint64_t tab[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t idx = (threadIdx.x * 977u + seed) & 16383u;
  uint32_t acc = 0;
  for (int i = 0; i < ITERS; i++)
  {
#pragma unroll
    for (int j = 0; j < 64; j++)
    {
      const uint64_t v = tab[(idx + j * 131u) & 16383u]; // independent: throughput incl. address VALU (2 ops)
      acc += (uint32_t)v;
    }
    idx += acc & 7u;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

__global__ void k_clock(uint64_t *out)
{
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t a = threadIdx.x;
  for (int i = 0; i < 200000; i++)
    asm volatile(R16("v_add_u32 %0, %0, %0\n\t") : "+v"(a));
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
    out[2] = a;
  }
}

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Test { const char *name; kern_t k; int instr_per_rep; };

int main()
{
  setvbuf(stdout, NULL, _IONBF, 0);
  uint32_t *out;
  uint64_t *clk;
  CHECK(hipMalloc(&out, 4096 * 256 * 4));
  CHECK(hipMalloc(&clk, 64));
  hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, clk);
  CHECK(hipDeviceSynchronize());
  uint64_t h[3];
  CHECK(hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] * 10.0); // memrealtime ticks at 100 MHz
  printf("shader clock under integer VALU load: %.3f GHz (memtime %llu / memrealtime %llu)\n", ghz, (unsigned long long)h[0], (unsigned long long)h[1]);

  std::vector<Test> tests = {
    {"v_add_u32 (dep)", k_add, 1}, {"v_add3_u32", k_add3, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
    {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mul_hi_u32_u24", k_mulhi_u24, 1}, {"v_mad_u64_u32", k_mad_u64, 1},
    {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_bfe_u32", k_bfe, 1},
    {"v_ffbh_u32", k_ffbh, 1}, {"v_perm_b32", k_perm, 1}, {"v_lshl_or_b32", k_lshl_or, 1}, {"v_lshl_add_u32", k_lshl_add, 1},
    {"v_xnor_b32", k_xnor, 1}, {"v_bfi_b32", k_bfi, 1}, {"add_co+addc (2)", k_addco, 2}, {"cmp+cndmask (2)", k_cmp_cnd, 2},
    {"cmp+addc (2)", k_cmp_addc, 2}, {"v_fma_f64", k_fma64, 1}, {"v_mul_f64", k_mul64, 1}, {"v_add_f64", k_add64, 1},
    {"cvt f64<->u32 (2)", k_cvt_f64_u32, 2}, {"v_rcp_f32", k_rcp32, 1}, {"v_rcp_f64", k_rcp64, 1}, {"cvt f32<->u32 (2)", k_cvt_f32_u32, 2},
    {"v_fma_f32", k_fma32, 1}, {"readfirstlane+add (2)", k_readlane, 2}, {"s_add + v_add (2)", k_salu_mix, 2}, {"2 salu + v_add (3)", k_salu2_mix, 3},
    {"4 valu, 2 chains (4)", k_ind4_add, 4}, {"cmp+branch(not taken)+add (3)", k_branch, 3},
    {"ds_read_b64 dep (+2 valu)", k_lds_b64, 1}, {"ds_read_b64 indep (+addr valu)", k_lds_b64_tp, 1},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-34s %14s %14s %14s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  for (auto &t : tests)
  {
    double res[3];
    for (int m = 0; m < 3; m++)
    {
      const int blocks = 256 << m; // 256 blocks x 256 threads = 1 wave per SIMD
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1u); // warm
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 2u);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      res[m] = (double)ms * 1e6 / ((double)ITERS * 64.0 * t.instr_per_rep); // ns per instruction per wave
    }
    printf("%-34s %6.2f ns %4.1fc %6.2f ns %4.1fc %6.2f ns %4.1fc\n", t.name, res[0], res[0] * ghz, res[1], res[1] * ghz, res[2], res[2] * ghz);
  }
  return 0;
}
