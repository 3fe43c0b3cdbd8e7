// dega_hip.hip -- host side of libdega_hip.so: the C ABI declared in include/dega_hip.h.
//
// Owns one context per device (division table in HBM, scratch buffers for the host-pointer entry points, optional
// hipEvent timing of the hot kernels) and launches the kernels of dega_kernels.hpp.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC dega_hip.hip -o libdega_hip.so      (see csrc/Makefile)
#include <hip/hip_runtime.h>

#include "../../include/dega_hip.h"
#include "dega_kernels.hpp"
#include "lzmh_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

using namespace dg;

struct dega_hip_ctx
{
  int device;
  uint32_t *div_magic; // device, DIV_TABLE_SIZE entries
  char last_error[256];
  bool profile;
  std::vector<hipEvent_t> ev[4]; // start/stop pairs per kernel kind (0 encode, 1 decode, 2 lzmh encode, 3 lzmh decode)
  std::vector<hipEvent_t> ev_pool;
  int force_waves; // 0 = choose by batch size; 4 / 8 = DEGA_WAVES_PER_WORKGROUP (measurement knob)
};

static int fail(dega_hip_ctx *ctx, int code, const char *what, hipError_t e)
{
  if (ctx != nullptr)
    snprintf(ctx->last_error, sizeof(ctx->last_error), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
  return code;
}

#define HIP_TRY(ctx, expr, code)                      \
  do                                                  \
  {                                                   \
    const hipError_t e_ = (expr);                     \
    if (e_ != hipSuccess)                             \
      return fail((ctx), (code), #expr, e_);          \
  } while (0)

static void build_div_table(std::vector<uint32_t> &tab)
{
  tab.assign(DIV_TABLE_SIZE, 0u);
  for (uint32_t t = 3; t < DIV_TABLE_SIZE; t++)
  {
    uint32_t L = 0;
    while ((1u << L) < t)
      L++;
    const unsigned __int128 num = (unsigned __int128)1 << (30 + L);
    tab[t] = (uint32_t)((num + t - 1) / t); // ceil(2^(30+L) / t); the shift L - 2 is recomputed from t (div_shift)
  }
}

extern "C" int dega_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

extern "C" const char *dega_hip_version(void)
{
  return "dega-hip 0.1 (gfx950)";
}

extern "C" size_t dega_hip_worst_case_bytes(size_t T)
{
  // seg: <= 65 bits per sample (seg.c:18-19).  bac: a coded bit costs at most log2(cum[0]/freq) <= 14 bits in the worst
  // model state, but over a whole stream the adaptive coder stays within a few percent of 1 bit/bit plus the counts'
  // learning cost; the static coder (frequencies 1:1:1) costs log2(3) bits per bit.  2 output bits per seg bit covers both.
  const size_t bits = T * 65 * 2 + 64;
  return ((bits + 7) / 8 + 16 + 3) & ~(size_t)3;
}

extern "C" size_t dega_hip_worst_case_bytes64(size_t T)
{
  const size_t bits = T * 127 * 2 + 64; // as above with the 127-bit worst-case codeword of 64-bit values
  return ((bits + 7) / 8 + 16 + 3) & ~(size_t)3;
}

extern "C" int dega_hip_create(int device, dega_hip_ctx **out)
{
  if (out == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    return DEGA_ERROR_LIBRARY_INIT; // no CPU fallback: fail loudly
  dega_hip_ctx *ctx = new dega_hip_ctx();
  ctx->device = device;
  ctx->div_magic = nullptr;
  ctx->last_error[0] = '\0';
  ctx->profile = false;
  {
    const char *w = getenv("DEGA_WAVES_PER_WORKGROUP");
    const int v = w != nullptr ? atoi(w) : 0;
    ctx->force_waves = (v == 4 || v == 8) ? v : 0;
  }
  if (hipSetDevice(device) != hipSuccess)
  {
    delete ctx;
    return DEGA_ERROR_LIBRARY_INIT;
  }
  std::vector<uint32_t> tab;
  build_div_table(tab);
  if (hipMalloc((void **)&ctx->div_magic, tab.size() * sizeof(uint32_t)) != hipSuccess)
  {
    delete ctx;
    return DEGA_ERROR_MEMORY;
  }
  if (hipMemcpy(ctx->div_magic, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
  {
    (void)hipFree(ctx->div_magic);
    delete ctx;
    return DEGA_ERROR_LIBRARY_INIT;
  }
  *out = ctx;
  return DEGA_OK;
}

extern "C" void dega_hip_destroy(dega_hip_ctx *ctx)
{
  if (ctx == nullptr)
    return;
  (void)hipSetDevice(ctx->device);
  for (int k = 0; k < 2; k++)
    for (hipEvent_t e : ctx->ev[k])
      (void)hipEventDestroy(e);
  (void)hipFree(ctx->div_magic);
  delete ctx;
}

extern "C" const char *dega_hip_last_error(const dega_hip_ctx *ctx)
{
  return ctx == nullptr ? "no context" : ctx->last_error;
}

extern "C" int dega_hip_profile(dega_hip_ctx *ctx, int enable)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  ctx->profile = enable != 0;
  return DEGA_OK;
}

extern "C" int dega_hip_profile_read(dega_hip_ctx *ctx, int which, double *avg_ms, int reset)
{
  if (ctx == nullptr || which < 0 || which > 3 || avg_ms == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  std::vector<hipEvent_t> &v = ctx->ev[which];
  double sum = 0.0;
  int n = 0;
  for (size_t i = 0; i + 1 < v.size(); i += 2)
  {
    float ms = 0.f;
    if (hipEventSynchronize(v[i + 1]) == hipSuccess && hipEventElapsedTime(&ms, v[i], v[i + 1]) == hipSuccess)
    {
      sum += ms;
      n++;
    }
  }
  *avg_ms = n ? sum / n : 0.0;
  if (reset)
  {
    for (hipEvent_t e : v)
      (void)hipEventDestroy(e);
    v.clear();
  }
  return n;
}

// RAII-free helper: brackets a launch with events on the launch's own stream when profiling is on
struct LaunchTimer
{
  dega_hip_ctx *ctx;
  int which;
  hipStream_t s;
  LaunchTimer(dega_hip_ctx *c, int w, hipStream_t st) : ctx(c), which(w), s(st)
  {
    if (ctx->profile)
    {
      hipEvent_t e;
      if (hipEventCreate(&e) == hipSuccess)
      {
        (void)hipEventRecord(e, s);
        ctx->ev[which].push_back(e);
      }
    }
  }
  ~LaunchTimer()
  {
    if (ctx->profile && (ctx->ev[which].size() & 1u))
    {
      hipEvent_t e;
      if (hipEventCreate(&e) == hipSuccess)
      {
        (void)hipEventRecord(e, s);
        ctx->ev[which].push_back(e);
      }
      else
      {
        (void)hipEventDestroy(ctx->ev[which].back());
        ctx->ev[which].pop_back();
      }
    }
  }
};

static int check_shape(dega_hip_ctx *ctx, size_t C, size_t T, size_t ld, size_t cap, int valuesize)
{
  (void)T;
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (valuesize < 1 || valuesize > 32)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "valuesize must be 1..32 for the [T][C] int32 layout", hipSuccess);
  if (ld < C)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "ld < C", hipSuccess);
  if (cap % 4 != 0 || cap > ((size_t)1 << 29))
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "cap must be a multiple of 4 and at most 512 MiB", hipSuccess);
  if (T > ((size_t)1 << 25))
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "at most 2^25 samples per channel and call", hipSuccess);
  return DEGA_OK;
}

extern "C" int dega_hip_encode_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                   uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  EncodeArgs a;
  a.x = x_tc;
  a.C = C;
  a.T = T;
  a.ld = ld;
  a.out = out;
  a.cap = cap;
  a.out_bits = out_bits;
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.valuesize = (uint32_t)valuesize;
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 0, s);
    if (ctx->force_waves == 8 || (ctx->force_waves == 0 && C > 65536)) // more than one wave per SIMD of work: 8-wave workgroups with smaller rings, two waves per SIMD
    {
      const dim3 grid((unsigned)((C + 511) / 512)), block(512);
      if (valuesize < 32)
      {
        if (adaptive)
          hipLaunchKernelGGL((dega_encode_kernel<true, true, 8, 4, 16, 24>), grid, block, 0, s, a);
        else
          hipLaunchKernelGGL((dega_encode_kernel<false, true, 8, 4, 16, 24>), grid, block, 0, s, a);
      }
      else if (adaptive)
        hipLaunchKernelGGL((dega_encode_kernel<true, false, 8, 4, 16, 24>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((dega_encode_kernel<false, false, 8, 4, 16, 24>), grid, block, 0, s, a);
    }
    else
    {
      const dim3 grid((unsigned)((C + BLOCK - 1) / BLOCK));
      if (valuesize < 32) // the narrow variants mask the samples and range check against the value size
      {
        if (adaptive)
          hipLaunchKernelGGL((dega_encode_kernel<true, true>), grid, dim3(BLOCK), 0, s, a);
        else
          hipLaunchKernelGGL((dega_encode_kernel<false, true>), grid, dim3(BLOCK), 0, s, a);
      }
      else if (adaptive)
        hipLaunchKernelGGL(dega_encode_kernel<true>, grid, dim3(BLOCK), 0, s, a);
      else
        hipLaunchKernelGGL(dega_encode_kernel<false>, grid, dim3(BLOCK), 0, s, a);
    }
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

static int launch_decode(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                         int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DecodeArgs a;
  a.in = in;
  a.cap = cap;
  a.in_bits = in_bits;
  a.C = C;
  a.T = T;
  a.ld = ld;
  a.x = x_tc;
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.out_count = out_count;
  a.valuesize = (uint32_t)valuesize;
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 1, s);
    if (ctx->force_waves == 8 || (ctx->force_waves == 0 && C > 65536)) // more than one wave per SIMD of work: 8-wave workgroups, two waves per SIMD
    {
      const dim3 grid((unsigned)((C + 511) / 512)), block(512);
      if (valuesize < 32)
      {
        if (adaptive)
          hipLaunchKernelGGL((dega_decode_kernel<true, true, 8>), grid, block, 0, s, a);
        else
          hipLaunchKernelGGL((dega_decode_kernel<false, true, 8>), grid, block, 0, s, a);
      }
      else if (adaptive)
        hipLaunchKernelGGL((dega_decode_kernel<true, false, 8>), grid, block, 0, s, a);
      else
        hipLaunchKernelGGL((dega_decode_kernel<false, false, 8>), grid, block, 0, s, a);
    }
    else
    {
      const dim3 grid((unsigned)((C + BLOCK - 1) / BLOCK));
      if (valuesize < 32)
      {
        if (adaptive)
          hipLaunchKernelGGL((dega_decode_kernel<true, true>), grid, dim3(BLOCK), 0, s, a);
        else
          hipLaunchKernelGGL((dega_decode_kernel<false, true>), grid, dim3(BLOCK), 0, s, a);
      }
      else if (adaptive)
        hipLaunchKernelGGL(dega_decode_kernel<true>, grid, dim3(BLOCK), 0, s, a);
      else
        hipLaunchKernelGGL(dega_decode_kernel<false>, grid, dim3(BLOCK), 0, s, a);
    }
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_decode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                   int adaptive, int valuesize, int32_t *x_tc, int32_t *err, void *stream)
{
  return launch_decode(ctx, in, cap, in_bits, C, T, ld, adaptive, valuesize, x_tc, nullptr, err, stream);
}

extern "C" int dega_hip_decode_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                       int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return launch_decode(ctx, in, cap, in_bits, C, max_T, ld, adaptive, valuesize, x_tc, out_count, err, stream);
}

static dim3 rowsplit_grid(size_t C, size_t T)
{
  // enough blocks to fill the chip: columns x row-chunks
  const unsigned gx = (unsigned)((C + BLOCK - 1) / BLOCK);
  unsigned gy = 1;
  while ((size_t)gx * gy < 2048 && gy < 1024 && (size_t)gy * 64 < T)
    gy *= 2;
  return dim3(gx, gy);
}

extern "C" int dega_hip_normalize_dev(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int valuesize,
                                      int32_t *x_tc, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(ctx, hipMemsetAsync(err, 0, C * sizeof(int32_t), s), DEGA_ERROR_LIBRARY_CALL);
  if (T == 0)
    return DEGA_OK;
  // the bounds of normalize.c:21, rounded to float by the host compiler exactly as the reference's are
  const float lo = -(float)((uint64_t)1 << (valuesize - 1)), hi = (float)(((uint64_t)1 << (valuesize - 1)) - 1);
  NormalizeArgs a{v_tc, x_tc, C, T, ld, factor, err, lo, hi, valuesize >= 32 ? 0xFFFFFFFFu : (1u << valuesize) - 1u};
  hipLaunchKernelGGL(dega_normalize_kernel, rowsplit_grid(C, T), dim3(BLOCK), 0, s, a);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_denormalize_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, float factor, int valuesize,
                                        float *v_tc, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, 0, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0 || T == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DenormalizeArgs a{x_tc, v_tc, C, T, ld, factor, (uint32_t)(32 - valuesize)};
  hipLaunchKernelGGL(dega_denormalize_kernel, rowsplit_grid(C, T), dim3(BLOCK), 0, (hipStream_t)stream, a);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

// exclusive prefix sum of ceil(bits/8) over channels: one block, chunked (C is at most a few million; not a hot path)
__global__ void __launch_bounds__(1024) dega_offsets_kernel(const uint64_t *bits, size_t C, uint64_t *offsets)
{
  __shared__ uint64_t part[1024];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0)
    carry = 0;
  __syncthreads();
  for (size_t base = 0; base < C; base += 1024)
  {
    const size_t i = base + threadIdx.x;
    const uint64_t v = i < C ? (bits[i] + 7) / 8 : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1)
    {
      const uint64_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < C)
      offsets[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023)
      carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0)
    offsets[C] = carry;
}

extern "C" int dega_hip_compact_offsets_dev(dega_hip_ctx *ctx, const uint64_t *bits, size_t C, uint64_t *offsets, void *stream)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  hipLaunchKernelGGL(dega_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bits, C, offsets);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_compact_gather_dev(dega_hip_ctx *ctx, const uint8_t *slabs, size_t cap, const uint64_t *offsets, size_t C,
                                           uint8_t *packed, void *stream)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  GatherArgs a{slabs, cap, offsets, C, packed};
  hipLaunchKernelGGL(dega_gather_kernel, dim3((unsigned)((C + WAVES - 1) / WAVES)), dim3(BLOCK), 0, (hipStream_t)stream, a);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_synth_dev(dega_hip_ctx *ctx, int32_t *x_tc, size_t C, size_t T, size_t ld, uint64_t seed, uint64_t c0, uint32_t S, void *stream)
{
  if (ctx == nullptr || ld < C)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0 || T == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  SynthArgs a{x_tc, C, T, ld, seed, c0, S};
  hipLaunchKernelGGL(dega_synth_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, (hipStream_t)stream, a);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

// ---- valuesize 33..64: int64 containers ----------------------------------------------------------------------------------

static int check_shape64(dega_hip_ctx *ctx, size_t C, size_t T, size_t ld, size_t cap, int valuesize)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (valuesize < 33 || valuesize > 64)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "valuesize must be 33..64 for the [T][C] int64 layout", hipSuccess);
  return check_shape(ctx, C, T, ld, cap, 32);
}

extern "C" int dega_hip_encode64_dev(dega_hip_ctx *ctx, const int64_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                     uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape64(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  EncodeArgs a;
  a.x = reinterpret_cast<const int32_t *>(x_tc);
  a.C = C;
  a.T = T;
  a.ld = ld;
  a.out = out;
  a.cap = cap;
  a.out_bits = out_bits;
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.valuesize = (uint32_t)valuesize;
  const dim3 grid((unsigned)((C + BLOCK - 1) / BLOCK));
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 0, s);
    if (adaptive)
      hipLaunchKernelGGL((dega_encode_kernel<true, false, 4, 4, 32, 32, true>), grid, dim3(BLOCK), 0, s, a);
    else
      hipLaunchKernelGGL((dega_encode_kernel<false, false, 4, 4, 32, 32, true>), grid, dim3(BLOCK), 0, s, a);
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

static int launch_decode64(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                           int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape64(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DecodeArgs a;
  a.in = in;
  a.cap = cap;
  a.in_bits = in_bits;
  a.C = C;
  a.T = T;
  a.ld = ld;
  a.x = reinterpret_cast<int32_t *>(x_tc);
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.out_count = out_count;
  a.valuesize = (uint32_t)valuesize;
  const dim3 grid((unsigned)((C + BLOCK - 1) / BLOCK));
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 1, s);
    if (adaptive)
      hipLaunchKernelGGL((dega_decode_kernel<true, false, 4, true>), grid, dim3(BLOCK), 0, s, a);
    else
      hipLaunchKernelGGL((dega_decode_kernel<false, false, 4, true>), grid, dim3(BLOCK), 0, s, a);
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_decode64_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                     int adaptive, int valuesize, int64_t *x_tc, int32_t *err, void *stream)
{
  return launch_decode64(ctx, in, cap, in_bits, C, T, ld, adaptive, valuesize, x_tc, nullptr, err, stream);
}

extern "C" int dega_hip_decode64_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                         int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return launch_decode64(ctx, in, cap, in_bits, C, max_T, ld, adaptive, valuesize, x_tc, out_count, err, stream);
}

// ---- LZMH (BASELINE config 4) ----------------------------------------------------------------------------------------

extern "C" size_t dega_hip_lzmh_worst_case_bytes(size_t n)
{
  return ((n * 10u + 7u) / 8u + 32u + 15u) / 16u * 16u; // every byte a 10-bit literal + the room the kernel keeps for its last words
}

static bool aligned16(const void *p)
{
  return ((uintptr_t)p & 15u) == 0;
}

extern "C" int dega_hip_lzmh_encode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out,
                                        size_t cap, uint64_t *out_bits, int32_t *err, void *stream)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (stride == 0 || (stride & 15u) != 0 || (cap & 15u) != 0 || cap < 48 || stride > 0x7FFFFFF0u || !aligned16(in) || !aligned16(out))
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "lzmh encode: stride/cap must be multiples of 16 (cap >= 48), buffers 16-byte aligned", hipSuccess);
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  LzmhEncodeArgs a{in, stride, in_len, C, out, cap, out_bits, err};
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 2, s);
    hipLaunchKernelGGL(lzmh_encode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZ_BLOCK), 0, s, a);
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_lzmh_decode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out,
                                        size_t stride, uint64_t *out_len, int32_t *err, void *stream)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (cap < 4 || (cap & 3u) != 0 || stride < 8 || (stride & 7u) != 0 || ((uintptr_t)in & 3u) != 0 || ((uintptr_t)out & 7u) != 0)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "lzmh decode: cap must be a multiple of 4, stride a multiple of 8", hipSuccess);
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  LzmhDecodeArgs a{in, cap, in_bits, C, out, stride, out_len, err};
  hipStream_t s = (hipStream_t)stream;
  {
    LaunchTimer lt(ctx, 3, s);
    if (C <= 65536 && ctx->force_waves != 4) // no more waves than SIMDs: half-filled waves, two per SIMD (see the kernel)
      hipLaunchKernelGGL(lzmh_decode_kernel<32>, dim3((unsigned)((C + 127) / 128)), dim3(LZ_BLOCK), 0, s, a);
    else
      hipLaunchKernelGGL(lzmh_decode_kernel<64>, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZ_BLOCK), 0, s, a);
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_lzmh_render_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, uint8_t *out, size_t stride,
                                        uint64_t *out_len, int32_t *err, void *stream)
{
  if (ctx == nullptr || ld < C || stride < 16 || (stride & 15u) != 0 || !aligned16(out))
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  RenderArgs a{x_tc, C, T, ld, out, stride, out_len, err};
  hipLaunchKernelGGL(lzmh_render_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

// ---- host-pointer entry points ---------------------------------------------------------------------------------------

struct DevBuf
{
  void *p = nullptr;
  ~DevBuf()
  {
    if (p != nullptr)
      (void)hipFree(p);
  }
  hipError_t alloc(size_t n)
  {
    return hipMalloc(&p, n > 0 ? n : 1);
  }
};

extern "C" int dega_hip_encode_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                    uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dx, dout, dbits, derr;
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(dx.p, x_tc, T * ld * sizeof(int32_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dout.p, 0, C * cap), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_encode_dev(ctx, (const int32_t *)dx.p, C, T, ld, adaptive, valuesize, (uint8_t *)dout.p, cap, (uint64_t *)dbits.p, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out, dout.p, C * cap, hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_bits, dbits.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

static int decode_host_impl(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                            int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err);

extern "C" int dega_hip_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                    int adaptive, int valuesize, int32_t *x_tc, int32_t *err)
{
  return decode_host_impl(ctx, in, cap, in_bits, C, T, ld, adaptive, valuesize, x_tc, nullptr, err);
}

extern "C" int dega_hip_decode_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                        int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err)
{
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return decode_host_impl(ctx, in, cap, in_bits, C, max_T, ld, adaptive, valuesize, x_tc, out_count, err);
}

static int decode_host_impl(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                            int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dx, din, dbits, derr;
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, din.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(din.p, in, C * cap, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dbits.p, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dx.p, 0, T * ld * sizeof(int32_t)), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dcount;
  if (out_count != nullptr)
    HIP_TRY(ctx, dcount.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  if ((ret = launch_decode(ctx, (const uint8_t *)din.p, cap, (const uint64_t *)dbits.p, C, T, ld, adaptive, valuesize, (int32_t *)dx.p,
                           out_count != nullptr ? (uint64_t *)dcount.p : nullptr, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(x_tc, dx.p, T * ld * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  if (out_count != nullptr)
    HIP_TRY(ctx, hipMemcpy(out_count, dcount.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_encode_f32_host(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int adaptive, int valuesize,
                                        uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dv, dx, dout, dbits, derr, dnerr;
  HIP_TRY(ctx, dv.alloc(T * ld * sizeof(float)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dnerr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(dv.p, v_tc, T * ld * sizeof(float), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dout.p, 0, C * cap), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_normalize_dev(ctx, (const float *)dv.p, C, T, ld, factor, valuesize, (int32_t *)dx.p, (int32_t *)dnerr.p, nullptr)) != DEGA_OK)
    return ret;
  if ((ret = dega_hip_encode_dev(ctx, (const int32_t *)dx.p, C, T, ld, adaptive, valuesize, (uint8_t *)dout.p, cap, (uint64_t *)dbits.p, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  std::vector<int32_t> nerr(C);
  HIP_TRY(ctx, hipMemcpy(out, dout.p, C * cap, hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_bits, dbits.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(nerr.data(), dnerr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  for (size_t c = 0; c < C; c++) // the first failing stage of the chain reports (normalize runs before diff)
    if (nerr[c] != DEGA_OK)
      err[c] = nerr[c];
  return DEGA_OK;
}

static int decode_f32_host_impl(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err);

extern "C" int dega_hip_decode_f32_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                        float factor, int adaptive, int valuesize, float *v_tc, int32_t *err)
{
  return decode_f32_host_impl(ctx, in, cap, in_bits, C, T, ld, factor, adaptive, valuesize, v_tc, nullptr, err);
}

extern "C" int dega_hip_decode_f32_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                            float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err)
{
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  return decode_f32_host_impl(ctx, in, cap, in_bits, C, max_T, ld, factor, adaptive, valuesize, v_tc, out_count, err);
}

static int decode_f32_host_impl(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dv, dx, din, dbits, derr;
  HIP_TRY(ctx, dv.alloc(T * ld * sizeof(float)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, din.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(din.p, in, C * cap, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dbits.p, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dx.p, 0, T * ld * sizeof(int32_t)), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dcount;
  if (out_count != nullptr)
    HIP_TRY(ctx, dcount.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  if ((ret = launch_decode(ctx, (const uint8_t *)din.p, cap, (const uint64_t *)dbits.p, C, T, ld, adaptive, valuesize, (int32_t *)dx.p,
                           out_count != nullptr ? (uint64_t *)dcount.p : nullptr, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  if (out_count != nullptr)
    HIP_TRY(ctx, hipMemcpy(out_count, dcount.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_denormalize_dev(ctx, (const int32_t *)dx.p, C, T, ld, factor, valuesize, (float *)dv.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(v_tc, dv.p, T * ld * sizeof(float), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_lzmh_encode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out,
                                         size_t cap, uint64_t *out_bits, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf din, dlen, dout, dbits, derr;
  HIP_TRY(ctx, din.alloc(C * stride), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dlen.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(din.p, in, C * stride, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dlen.p, in_len, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dout.p, 0, C * cap), DEGA_ERROR_LIBRARY_CALL);
  int ret;
  if ((ret = dega_hip_lzmh_encode_dev(ctx, (const uint8_t *)din.p, stride, (const uint64_t *)dlen.p, C, (uint8_t *)dout.p, cap, (uint64_t *)dbits.p,
                                      (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out, dout.p, C * cap, hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_bits, dbits.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_lzmh_decode_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out,
                                         size_t stride, uint64_t *out_len, int32_t *err)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf din, dbits, dout, dlen, derr;
  HIP_TRY(ctx, din.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * stride), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dlen.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(din.p, in, C * cap, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dbits.p, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  int ret;
  if ((ret = dega_hip_lzmh_decode_dev(ctx, (const uint8_t *)din.p, cap, (const uint64_t *)dbits.p, C, (uint8_t *)dout.p, stride, (uint64_t *)dlen.p,
                                      (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out, dout.p, C * stride, hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_len, dlen.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_encode64_host(dega_hip_ctx *ctx, const int64_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                      uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err)
{
  int ret;
  if ((ret = check_shape64(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dx, dout, dbits, derr;
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(dx.p, x_tc, T * ld * sizeof(int64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dout.p, 0, C * cap), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_encode64_dev(ctx, (const int64_t *)dx.p, C, T, ld, adaptive, valuesize, (uint8_t *)dout.p, cap, (uint64_t *)dbits.p, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out, dout.p, C * cap, hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_bits, dbits.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_decode64_var_host(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                          int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape64(ctx, C, max_T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf din, dbits, dx, dcnt, derr;
  HIP_TRY(ctx, din.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dx.alloc(max_T * ld * sizeof(int64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dcnt.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(din.p, in, C * cap, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dbits.p, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dx.p, 0, max_T * ld * sizeof(int64_t)), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_decode64_var_dev(ctx, (const uint8_t *)din.p, cap, (const uint64_t *)dbits.p, C, max_T, ld, adaptive, valuesize, (int64_t *)dx.p,
                                       (uint64_t *)dcnt.p, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(x_tc, dx.p, max_T * ld * sizeof(int64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_count, dcnt.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_encode_packed_host(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                           uint8_t *packed, size_t packed_cap, uint64_t *offsets, uint64_t *out_bits, int32_t *err)
{
  int ret;
  const size_t cap = dega_hip_worst_case_bytes(T);
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (offsets == nullptr || out_bits == nullptr || err == nullptr || (packed == nullptr && packed_cap != 0))
    return DEGA_ERROR_INVALID_VALUE;
  offsets[0] = 0;
  if (C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  DevBuf dx, dout, dbits, derr, doff, dpacked;
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dout.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, doff.alloc((C + 1) * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, hipMemcpy(dx.p, x_tc, T * ld * sizeof(int32_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  if ((ret = dega_hip_encode_dev(ctx, (const int32_t *)dx.p, C, T, ld, adaptive, valuesize, (uint8_t *)dout.p, cap, (uint64_t *)dbits.p, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  if ((ret = dega_hip_compact_offsets_dev(ctx, (const uint64_t *)dbits.p, C, (uint64_t *)doff.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipMemcpy(offsets, doff.p, (C + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(out_bits, dbits.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  if (offsets[C] > packed_cap)
    return fail(ctx, DEGA_ERROR_MEMORY, "packed buffer too small: offsets[C] holds the size needed", hipSuccess);
  if (offsets[C] == 0)
    return DEGA_OK;
  HIP_TRY(ctx, dpacked.alloc((size_t)offsets[C]), DEGA_ERROR_MEMORY);
  if ((ret = dega_hip_compact_gather_dev(ctx, (const uint8_t *)dout.p, cap, (const uint64_t *)doff.p, C, (uint8_t *)dpacked.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipMemcpy(packed, dpacked.p, (size_t)offsets[C], hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

extern "C" int dega_hip_decode_packed_host(dega_hip_ctx *ctx, const uint8_t *packed, const uint64_t *offsets, const uint64_t *in_bits, size_t C, size_t T,
                                           size_t ld, int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err)
{
  int ret;
  if (ctx == nullptr || offsets == nullptr || in_bits == nullptr || x_tc == nullptr || err == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (C == 0)
    return DEGA_OK;
  if (packed == nullptr && offsets[C] != 0)
    return DEGA_ERROR_INVALID_VALUE;
  uint64_t longest = 0;
  for (size_t c = 0; c < C; c++)
  {
    if (offsets[c + 1] < offsets[c] || (in_bits[c] + 7) / 8 > offsets[c + 1] - offsets[c])
      return fail(ctx, DEGA_ERROR_INVALID_VALUE, "offsets must grow and hold ceil(bits / 8) bytes per channel", hipSuccess);
    longest = offsets[c + 1] - offsets[c] > longest ? offsets[c + 1] - offsets[c] : longest;
  }
  const size_t cap = ((size_t)longest + 16 + 3) & ~(size_t)3; // room for the decoder's word look-ahead
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  const size_t total = (size_t)offsets[C];
  DevBuf dpacked, doff, dslabs, dbits, dx, dcnt, derr;
  HIP_TRY(ctx, dpacked.alloc(total), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, doff.alloc((C + 1) * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dslabs.alloc(C * cap), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dbits.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dx.alloc(T * ld * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, dcnt.alloc(C * sizeof(uint64_t)), DEGA_ERROR_MEMORY);
  HIP_TRY(ctx, derr.alloc(C * sizeof(int32_t)), DEGA_ERROR_MEMORY);
  if (total > 0)
    HIP_TRY(ctx, hipMemcpy(dpacked.p, packed, total, hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(doff.p, offsets, (C + 1) * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(dbits.p, in_bits, C * sizeof(uint64_t), hipMemcpyHostToDevice), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemset(dx.p, 0, T * ld * sizeof(int32_t)), DEGA_ERROR_LIBRARY_CALL);
  {
    GatherArgs g{(const uint8_t *)dslabs.p, cap, (const uint64_t *)doff.p, C, (uint8_t *)dpacked.p};
    hipLaunchKernelGGL(dega_scatter_kernel, dim3((unsigned)((C + WAVES - 1) / WAVES)), dim3(BLOCK), 0, nullptr, g);
    HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  }
  if ((ret = launch_decode(ctx, (const uint8_t *)dslabs.p, cap, (const uint64_t *)dbits.p, C, T, ld, adaptive, valuesize, (int32_t *)dx.p,
                           out_count != nullptr ? (uint64_t *)dcnt.p : nullptr, (int32_t *)derr.p, nullptr)) != DEGA_OK)
    return ret;
  HIP_TRY(ctx, hipDeviceSynchronize(), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(x_tc, dx.p, T * ld * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  if (out_count != nullptr)
    HIP_TRY(ctx, hipMemcpy(out_count, dcnt.p, C * sizeof(uint64_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  HIP_TRY(ctx, hipMemcpy(err, derr.p, C * sizeof(int32_t), hipMemcpyDeviceToHost), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}
