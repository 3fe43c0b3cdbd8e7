// pcie_bench.hip -- what the host-pointer entry points of libdega_hip.so can expect from this box's PCIe link and host
// memory: pageable vs pinned hipMemcpy, hipHostRegister cost, host memcpy rates (1 and N threads), 2-D copies.
//   hipcc --offload-arch=gfx950 -O2 tools/pcie_bench.hip -o tools/pcie_bench -lpthread && tools/pcie_bench [MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char *dst, const char *src, size_t n, int threads)
{
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~(size_t)4095;
  for (int i = 0; i < threads; i++)
  {
    const size_t o = (size_t)i * per;
    if (o >= n)
      break;
    const size_t len = o + per <= n ? per : n - o;
    th.emplace_back([=] { memcpy(dst + o, src + o, len); });
  }
  for (auto &t : th)
    t.join();
}

int main(int argc, char **argv)
{
  const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2048;
  const size_t n = mib << 20;
  printf("buffer %zu MiB, hardware threads %u\n", mib, std::thread::hardware_concurrency());
  char *pageable = (char *)malloc(n), *pageable2 = (char *)malloc(n), *pinned = nullptr, *dev = nullptr;
  memset(pageable, 1, n);
  memset(pageable2, 2, n);
  double t0 = now();
  CHECK(hipHostMalloc((void **)&pinned, n, hipHostMallocDefault));
  printf("hipHostMalloc            %8.1f ms\n", (now() - t0) * 1e3);
  memset(pinned, 3, n);
  CHECK(hipMalloc((void **)&dev, n));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  for (int rep = 0; rep < 2; rep++)
  {
    t0 = now();
    CHECK(hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice));
    printf("H2D pageable hipMemcpy   %8.2f GB/s\n", n / (now() - t0) / 1e9);
    t0 = now();
    CHECK(hipMemcpy(pageable2, dev, n, hipMemcpyDeviceToHost));
    printf("D2H pageable hipMemcpy   %8.2f GB/s\n", n / (now() - t0) / 1e9);
    t0 = now();
    CHECK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    printf("H2D pinned async         %8.2f GB/s\n", n / (now() - t0) / 1e9);
    t0 = now();
    CHECK(hipMemcpyAsync(pinned, dev, n, hipMemcpyDeviceToHost, s));
    CHECK(hipStreamSynchronize(s));
    printf("D2H pinned async         %8.2f GB/s\n", n / (now() - t0) / 1e9);
  }
  // both directions at once on two streams
  {
    hipStream_t s2;
    CHECK(hipStreamCreate(&s2));
    char *dev2 = nullptr, *pinned2 = nullptr;
    CHECK(hipMalloc((void **)&dev2, n));
    CHECK(hipHostMalloc((void **)&pinned2, n, hipHostMallocDefault));
    t0 = now();
    CHECK(hipMemcpyAsync(dev, pinned, n, hipMemcpyHostToDevice, s));
    CHECK(hipMemcpyAsync(pinned2, dev2, n, hipMemcpyDeviceToHost, s2));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipStreamSynchronize(s2));
    printf("H2D + D2H concurrently   %8.2f GB/s each direction\n", n / (now() - t0) / 1e9);
    CHECK(hipFree(dev2));
    CHECK(hipHostFree(pinned2));
  }
  // 2-D copy: rows of 32 KiB out of a 256 KiB pitch (a channel chunk of a [T][ld] array)
  {
    const size_t width = 32768, pitch = 262144, rows = n / pitch;
    t0 = now();
    CHECK(hipMemcpy2DAsync(dev, width, pinned, pitch, width, rows, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    printf("H2D pinned 2-D 32K/256K  %8.2f GB/s (%zu rows)\n", width * rows / (now() - t0) / 1e9, rows);
    t0 = now();
    CHECK(hipMemcpy2DAsync(pinned, pitch, dev, width, width, rows, hipMemcpyDeviceToHost, s));
    CHECK(hipStreamSynchronize(s));
    printf("D2H pinned 2-D 32K/256K  %8.2f GB/s\n", width * rows / (now() - t0) / 1e9);
  }
  // the same with wider pieces, with two and four 2-D copies side by side on their own streams (adjacent column ranges),
  // and with the copy done by a kernel reading the pinned rows (what the encode kernel's filling waves would do)
  {
    const size_t pitch = 262144, rows = n / pitch;
    for (size_t width : {(size_t)65536, (size_t)131072})
    {
      t0 = now();
      CHECK(hipMemcpy2DAsync(dev, width, pinned, pitch, width, rows, hipMemcpyHostToDevice, s));
      CHECK(hipStreamSynchronize(s));
      printf("H2D pinned 2-D %3zuK/256K %8.2f GB/s\n", width >> 10, width * rows / (now() - t0) / 1e9);
    }
    hipStream_t ss[4];
    for (auto &x : ss)
      CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (int k : {2, 4})
    {
      const size_t width = 32768;
      t0 = now();
      for (int i = 0; i < k; i++)
        CHECK(hipMemcpy2DAsync(dev + (size_t)i * width * rows, width, pinned + (size_t)i * width, pitch, width, rows, hipMemcpyHostToDevice, ss[i]));
      for (int i = 0; i < k; i++)
        CHECK(hipStreamSynchronize(ss[i]));
      printf("H2D pinned 2-D 32K/256K x %d streams %8.2f GB/s in all\n", k, k * width * rows / (now() - t0) / 1e9);
    }
    // one chunk's rows cut into 4 row ranges on 4 streams (the same columns)
    {
      const size_t width = 32768, part = rows / 4;
      t0 = now();
      for (int i = 0; i < 4; i++)
        CHECK(hipMemcpy2DAsync(dev + (size_t)i * part * width, width, pinned + (size_t)i * part * pitch, pitch, width, part, hipMemcpyHostToDevice, ss[i]));
      for (int i = 0; i < 4; i++)
        CHECK(hipStreamSynchronize(ss[i]));
      printf("H2D pinned 2-D 32K/256K rows in 4 ranges on 4 streams %8.2f GB/s\n", 4 * part * width / (now() - t0) / 1e9);
    }
    for (auto &x : ss)
      CHECK(hipStreamDestroy(x));
  }
  // chunked pinned copies (64 MiB pieces)
  {
    const size_t piece = 64u << 20;
    t0 = now();
    for (size_t o = 0; o < n; o += piece)
      CHECK(hipMemcpyAsync(dev + o, pinned + o, o + piece <= n ? piece : n - o, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    printf("H2D pinned 64 MiB pieces %8.2f GB/s\n", n / (now() - t0) / 1e9);
  }
  // host memcpy rates
  for (int th : {1, 2, 4, 8, 16})
  {
    t0 = now();
    par_memcpy(pinned, pageable, n, th);
    printf("memcpy pageable->pinned, %2d threads %8.2f GB/s\n", th, n / (now() - t0) / 1e9);
  }
  // staged H2D: memcpy into 4 pinned 32 MiB buffers by N threads, DMA from there, overlapped
  for (int th : {1, 4, 8})
  {
    const size_t piece = 32u << 20;
    const int NB = 4;
    hipEvent_t ev[NB];
    for (int i = 0; i < NB; i++)
      CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    t0 = now();
    int k = 0;
    for (size_t o = 0; o < n; o += piece, k++)
    {
      const int b = k % NB;
      const size_t len = o + piece <= n ? piece : n - o;
      if (k >= NB)
        CHECK(hipEventSynchronize(ev[b]));
      par_memcpy(pinned + (size_t)b * piece, pageable + o, len, th);
      CHECK(hipMemcpyAsync(dev + o, pinned + (size_t)b * piece, len, hipMemcpyHostToDevice, s));
      CHECK(hipEventRecord(ev[b], s));
    }
    CHECK(hipStreamSynchronize(s));
    printf("H2D staged (4 x 32 MiB pinned ring), %d copy threads %8.2f GB/s\n", th, n / (now() - t0) / 1e9);
  }
  // registering the caller's pageable buffer in place
  {
    t0 = now();
    hipError_t e = hipHostRegister(pageable, n, hipHostRegisterDefault);
    const double treg = now() - t0;
    if (e == hipSuccess)
    {
      printf("hipHostRegister          %8.1f ms (%.2f GB/s)\n", treg * 1e3, n / treg / 1e9);
      t0 = now();
      CHECK(hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, s));
      CHECK(hipStreamSynchronize(s));
      printf("H2D registered           %8.2f GB/s\n", n / (now() - t0) / 1e9);
      t0 = now();
      CHECK(hipHostUnregister(pageable));
      printf("hipHostUnregister        %8.1f ms\n", (now() - t0) * 1e3);
    }
    else
      printf("hipHostRegister failed: %s\n", hipGetErrorString(e));
  }
  // allocation costs
  {
    void *p;
    t0 = now();
    CHECK(hipMalloc(&p, n));
    const double ta = now() - t0;
    t0 = now();
    CHECK(hipFree(p));
    printf("hipMalloc %zu MiB %8.2f ms, hipFree %8.2f ms\n", mib, ta * 1e3, (now() - t0) * 1e3);
  }
  return 0;
}
