// dega_hip.hip -- host side of libdega_hip.so: the C ABI declared in include/dega_hip.h.
//
// One context per device: the division table in HBM, pooled hipEvents for the optional kernel timing, and -- for the
// host-pointer entry points -- a pipeline of streams with grow-only device and pinned-host buffers (dega_pipeline.hpp).
// A group (dega_hip_group) is a set of contexts, one host thread per device, channel ranges per device, host-side
// concatenation of the packed streams; no collective.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC dega_hip.hip -o libdega_hip.so      (see csrc/Makefile)
#include <hip/hip_runtime.h>

#include "../../include/dega_hip.h"
#include "dega_kernels.hpp"
#include "lzmh_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using namespace dg;

struct Pipeline;
static void pipeline_destroy(Pipeline *p);

struct dega_hip_ctx
{
  int device;
  uint32_t *div_magic; // device, DIV_TABLE_SIZE entries
  char last_error[256];
  bool profile;
  std::vector<hipEvent_t> ev[4]; // start/stop pairs per kernel kind (0 encode, 1 decode, 2 lzmh encode, 3 lzmh decode)
  std::vector<hipEvent_t> ev_pool; // events handed back by profile_read, reused by the next timed launches
  int force_waves; // 0 = choose by batch size; 4 / 8 = pairs of waves per workgroup, DEGA_WAVES_PER_WORKGROUP (measurement knob)
  Pipeline *pipe;  // streams and buffers of the host-pointer entry points, created on first use
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
  return "dega-hip 0.2 (gfx950)";
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
  ctx->pipe = nullptr;
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
  (void)hipDeviceSynchronize();
  pipeline_destroy(ctx->pipe);
  for (int k = 0; k < 4; k++)
    for (hipEvent_t e : ctx->ev[k])
      (void)hipEventDestroy(e);
  for (hipEvent_t e : ctx->ev_pool)
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
  (void)hipSetDevice(ctx->device);
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
    ctx->ev_pool.insert(ctx->ev_pool.end(), v.begin(), v.end()); // kept for the next timed launches
    v.clear();
  }
  return n;
}

// Brackets a launch with events on the launch's own stream when profiling is on.  Events come from the context's pool;
// a context never holds more than DEGA_MAX_TIMED launches' worth (later launches go untimed until profile_read resets).
constexpr size_t DEGA_MAX_TIMED = 4096;
struct LaunchTimer
{
  dega_hip_ctx *ctx;
  int which;
  hipStream_t s;
  bool armed;
  static bool take(dega_hip_ctx *ctx, hipEvent_t *e)
  {
    if (!ctx->ev_pool.empty())
    {
      *e = ctx->ev_pool.back();
      ctx->ev_pool.pop_back();
      return true;
    }
    return hipEventCreate(e) == hipSuccess;
  }
  LaunchTimer(dega_hip_ctx *c, int w, hipStream_t st) : ctx(c), which(w), s(st), armed(false)
  {
    hipEvent_t e;
    if (ctx->profile && ctx->ev[which].size() < 2 * DEGA_MAX_TIMED && take(ctx, &e))
    {
      (void)hipEventRecord(e, s);
      ctx->ev[which].push_back(e);
      armed = true;
    }
  }
  ~LaunchTimer()
  {
    if (!armed)
      return;
    hipEvent_t e;
    if (take(ctx, &e))
    {
      (void)hipEventRecord(e, s);
      ctx->ev[which].push_back(e);
    }
    else
    {
      ctx->ev_pool.push_back(ctx->ev[which].back());
      ctx->ev[which].pop_back();
    }
  }
};

constexpr size_t DEGA_MAX_T = (size_t)1 << 25; // samples per channel and call: T * 65 seg bits are counted in 32 bits

static int check_shape(dega_hip_ctx *ctx, size_t C, size_t T, size_t ld, size_t cap, int valuesize)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if (valuesize < 1 || valuesize > 32)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "valuesize must be 1..32 for the [T][C] int32 layout", hipSuccess);
  if (ld < C)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "ld < C", hipSuccess);
  if (cap % 4 != 0 || cap > ((size_t)1 << 29))
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "cap must be a multiple of 4 and at most 512 MiB", hipSuccess);
  if (T > DEGA_MAX_T)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "at most 2^25 samples per channel and call", hipSuccess);
  return DEGA_OK;
}

// ---- kernel dispatch ---------------------------------------------------------------------------------------------------
// One batch: its size and what its samples are (include/dega_hip.h: DEGA_SAMPLES_*)
struct Shape
{
  size_t C, T, ld;
  int adaptive, valuesize, samples;
  float factor;
};

static int check_job_shape(dega_hip_ctx *ctx, const Shape &j, size_t cap)
{
  if (ctx == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  const bool wide = j.valuesize > 32;
  if (j.samples < DEGA_SAMPLES_I32 || j.samples > DEGA_SAMPLES_F32 || j.valuesize < 1 || j.valuesize > 64 ||
      ((j.samples == DEGA_SAMPLES_I32 || j.samples == DEGA_SAMPLES_BE32) && wide) || (j.samples == DEGA_SAMPLES_I64 && !wide))
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "sample type and valuesize do not go together (int32 / big-endian: 1..32, int64: 33..64, float32: 1..64)",
                hipSuccess);
  return check_shape(ctx, j.C, j.T, j.ld, cap, 32);
}

template <bool AD, bool NARROW, bool F32>
static void encode_launch(size_t C, hipStream_t s, const EncodeArgs &a)
{
  const dim3 grid((unsigned)((C + ENC_CHANNELS - 1) / ENC_CHANNELS));
  // short channels (and enough of them to fill the chip twice): half the table, small rings, two workgroups per CU
  if (AD && a.T <= ENC_SHORT_T && a.seg_state == nullptr && C > 65536)
    hipLaunchKernelGGL((dega_encode_kernel<AD, NARROW, 4, 16, 8, 16, false, F32, ENC_PAIRS, ENC_SHORT_TABLE>), grid, dim3(ENC_BLOCK), 0, s, a);
  else
    hipLaunchKernelGGL((dega_encode_kernel<AD, NARROW, ENC_ROWS, ENC_RING, ENC_RAW, ENC_ORING, false, F32>), grid, dim3(ENC_BLOCK), 0, s, a);
}

// `batch_C`: the channel count the workgroup shape is chosen by (the whole batch's when this launch is one chunk of it)
static int launch_encode(dega_hip_ctx *ctx, const void *x, const Shape &j, size_t batch_C, uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err,
                         hipStream_t s, uint32_t *seg_state = nullptr, uint32_t seg_flags = 0)
{
  int ret;
  if ((ret = check_job_shape(ctx, j, cap)) != DEGA_OK)
    return ret;
  if (j.C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  const int vs = j.valuesize;
  EncodeArgs a;
  a.x = reinterpret_cast<const int32_t *>(x);
  a.C = j.C;
  a.T = j.T;
  a.ld = j.ld;
  a.out = out;
  a.cap = cap;
  a.out_bits = out_bits;
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.valuesize = (uint32_t)vs;
  a.big_endian = j.samples == DEGA_SAMPLES_BE32 ? 1u : 0u;
  a.factor = j.factor;
  a.seg_state = seg_state;
  a.seg_flags = seg_flags;
  // the bounds of normalize.c:21, rounded to float by the host compiler exactly as the reference's are
  a.lo = -(float)((uint64_t)1 << (vs - 1));
  a.hi = (float)(((uint64_t)1 << (vs - 1)) - 1);
  const bool f32 = j.samples == DEGA_SAMPLES_F32, ad = j.adaptive != 0;
  (void)batch_C; // (one workgroup shape for every batch size)
  {
    LaunchTimer lt(ctx, 0, s);
    if (vs > 32) // 64-bit values
    {
      const dim3 grid((unsigned)((j.C + ENC_CHANNELS - 1) / ENC_CHANNELS));
      if (f32)
      {
        if (ad)
          hipLaunchKernelGGL((dega_encode_kernel<true, false, 4, 32, 16, 32, true, true>), grid, dim3(ENC_BLOCK), 0, s, a);
        else
          hipLaunchKernelGGL((dega_encode_kernel<false, false, 4, 32, 16, 32, true, true>), grid, dim3(ENC_BLOCK), 0, s, a);
      }
      else if (ad)
        hipLaunchKernelGGL((dega_encode_kernel<true, false, 4, 32, 16, 32, true, false>), grid, dim3(ENC_BLOCK), 0, s, a);
      else
        hipLaunchKernelGGL((dega_encode_kernel<false, false, 4, 32, 16, 32, true, false>), grid, dim3(ENC_BLOCK), 0, s, a);
    }
    else
    {
      const bool narrow = vs < 32; // the narrow variants mask the samples and range check against the value size
      const int sel = (ad ? 4 : 0) | (narrow ? 2 : 0) | (f32 ? 1 : 0);
      switch (sel)
      {
        case 0: encode_launch<false, false, false>(j.C, s, a); break;
        case 1: encode_launch<false, false, true>(j.C, s, a); break;
        case 2: encode_launch<false, true, false>(j.C, s, a); break;
        case 3: encode_launch<false, true, true>(j.C, s, a); break;
        case 4: encode_launch<true, false, false>(j.C, s, a); break;
        case 5: encode_launch<true, false, true>(j.C, s, a); break;
        case 6: encode_launch<true, true, false>(j.C, s, a); break;
        default: encode_launch<true, true, true>(j.C, s, a); break;
      }
    }
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

template <bool AD, bool NARROW, bool F32>
static void decode_launch(bool wide, size_t C, hipStream_t s, const DecodeArgs &a)
{
  if (wide)
    hipLaunchKernelGGL((dega_decode_kernel<AD, NARROW, false, F32, 8, false>), dim3((unsigned)((C + 511) / 512)), dim3(1024), 0, s, a);
  else
    hipLaunchKernelGGL((dega_decode_kernel<AD, NARROW, false, F32>), dim3((unsigned)((C + DEC_CHANNELS - 1) / DEC_CHANNELS)), dim3(DEC_BLOCK), 0, s, a);
}

static int launch_decode(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, const Shape &j, size_t batch_C, void *x,
                         uint64_t *out_count, int32_t *err, hipStream_t s, uint32_t *rows_done = nullptr, uint32_t band_rows = 0)
{
  int ret;
  if ((ret = check_job_shape(ctx, j, cap)) != DEGA_OK)
    return ret;
  if (j.C == 0)
    return DEGA_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device), DEGA_ERROR_LIBRARY_CALL);
  const int vs = j.valuesize;
  DecodeArgs a;
  a.in = in;
  a.cap = cap;
  a.in_bits = in_bits;
  a.C = j.C;
  a.T = j.T;
  a.ld = j.ld;
  a.x = reinterpret_cast<int32_t *>(x);
  a.err = err;
  a.div_magic = ctx->div_magic;
  a.out_count = out_count;
  a.valuesize = (uint32_t)vs;
  a.big_endian = j.samples == DEGA_SAMPLES_BE32 ? 1u : 0u;
  a.factor = j.factor;
  a.rows_done = band_rows != 0 ? rows_done : nullptr;
  a.band_rows = band_rows;
  const bool f32 = j.samples == DEGA_SAMPLES_F32, ad = j.adaptive != 0;
  const bool wide = ctx->force_waves == 8 || (ctx->force_waves == 0 && batch_C > 65536);
  {
    LaunchTimer lt(ctx, 1, s);
    if (vs > 32)
    {
      const dim3 grid((unsigned)((j.C + DEC_CHANNELS - 1) / DEC_CHANNELS));
      if (f32)
      {
        if (ad)
          hipLaunchKernelGGL((dega_decode_kernel<true, false, true, true>), grid, dim3(DEC_BLOCK), 0, s, a);
        else
          hipLaunchKernelGGL((dega_decode_kernel<false, false, true, true>), grid, dim3(DEC_BLOCK), 0, s, a);
      }
      else if (ad)
        hipLaunchKernelGGL((dega_decode_kernel<true, false, true, false>), grid, dim3(DEC_BLOCK), 0, s, a);
      else
        hipLaunchKernelGGL((dega_decode_kernel<false, false, true, false>), grid, dim3(DEC_BLOCK), 0, s, a);
    }
    else
    {
      const bool narrow = vs < 32;
      const int sel = (ad ? 4 : 0) | (narrow ? 2 : 0) | (f32 ? 1 : 0);
      switch (sel)
      {
        case 0: decode_launch<false, false, false>(wide, j.C, s, a); break;
        case 1: decode_launch<false, false, true>(wide, j.C, s, a); break;
        case 2: decode_launch<false, true, false>(wide, j.C, s, a); break;
        case 3: decode_launch<false, true, true>(wide, j.C, s, a); break;
        case 4: decode_launch<true, false, false>(wide, j.C, s, a); break;
        case 5: decode_launch<true, false, true>(wide, j.C, s, a); break;
        case 6: decode_launch<true, true, false>(wide, j.C, s, a); break;
        default: decode_launch<true, true, true>(wide, j.C, s, a); break;
      }
    }
  }
  HIP_TRY(ctx, hipGetLastError(), DEGA_ERROR_LIBRARY_CALL);
  return DEGA_OK;
}

static Shape shape_of(size_t C, size_t T, size_t ld, int adaptive, int valuesize, int samples, float factor = 0.0f)
{
  Shape j;
  j.C = C;
  j.T = T;
  j.ld = ld;
  j.adaptive = adaptive;
  j.valuesize = valuesize;
  j.samples = samples;
  j.factor = factor;
  return j;
}

// ---- device-pointer entry points ---------------------------------------------------------------------------------------

extern "C" int dega_hip_encode_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T, size_t ld, int adaptive, int valuesize,
                                   uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  return launch_encode(ctx, x_tc, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), C, out, cap, out_bits, err, (hipStream_t)stream);
}

extern "C" size_t dega_hip_encode_state_bytes(size_t C)
{
  return (size_t)ENC_STATE_WORDS * C * sizeof(uint32_t);
}

extern "C" int dega_hip_encode_segment_dev(dega_hip_ctx *ctx, const int32_t *x_tc, size_t C, size_t T_seg, size_t ld, int adaptive, int valuesize,
                                           uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *state, unsigned flags, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T_seg, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  if (state == nullptr || (flags & ~(unsigned)(DEGA_SEGMENT_CONTINUES | DEGA_SEGMENT_MORE)) != 0u || ((uintptr_t)state & 3u) != 0)
    return fail(ctx, DEGA_ERROR_INVALID_VALUE, "encode segment: state (device memory, dega_hip_encode_state_bytes) and flags DEGA_SEGMENT_*", hipSuccess);
  static_assert(DEGA_SEGMENT_CONTINUES == ENC_SEG_CONTINUES && DEGA_SEGMENT_MORE == ENC_SEG_MORE, "the flags of the header are the kernel's");
  return launch_encode(ctx, x_tc, shape_of(C, T_seg, ld, adaptive, valuesize, DEGA_SAMPLES_I32), C, out, cap, out_bits, err, (hipStream_t)stream,
                       (uint32_t *)state, flags);
}

extern "C" int dega_hip_decode_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                   int adaptive, int valuesize, int32_t *x_tc, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  return launch_decode(ctx, in, cap, in_bits, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), C, x_tc, nullptr, err, (hipStream_t)stream);
}

extern "C" int dega_hip_decode_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                       int adaptive, int valuesize, int32_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  int ret;
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape(ctx, C, max_T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  return launch_decode(ctx, in, cap, in_bits, shape_of(C, max_T, ld, adaptive, valuesize, DEGA_SAMPLES_I32), C, x_tc, out_count, err, (hipStream_t)stream);
}

extern "C" int dega_hip_encode_f32_dev(dega_hip_ctx *ctx, const float *v_tc, size_t C, size_t T, size_t ld, float factor, int adaptive, int valuesize,
                                       uint8_t *out, size_t cap, uint64_t *out_bits, int32_t *err, void *stream)
{
  return launch_encode(ctx, v_tc, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_F32, factor), C, out, cap, out_bits, err, (hipStream_t)stream);
}

extern "C" int dega_hip_decode_f32_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                       float factor, int adaptive, int valuesize, float *v_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  return launch_decode(ctx, in, cap, in_bits, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_F32, factor), C, v_tc, out_count, err, (hipStream_t)stream);
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
  return launch_encode(ctx, x_tc, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I64), C, out, cap, out_bits, err, (hipStream_t)stream);
}

extern "C" int dega_hip_decode64_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld,
                                     int adaptive, int valuesize, int64_t *x_tc, int32_t *err, void *stream)
{
  int ret;
  if ((ret = check_shape64(ctx, C, T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  return launch_decode(ctx, in, cap, in_bits, shape_of(C, T, ld, adaptive, valuesize, DEGA_SAMPLES_I64), C, x_tc, nullptr, err, (hipStream_t)stream);
}

extern "C" int dega_hip_decode64_var_dev(dega_hip_ctx *ctx, const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t max_T, size_t ld,
                                         int adaptive, int valuesize, int64_t *x_tc, uint64_t *out_count, int32_t *err, void *stream)
{
  int ret;
  if (out_count == nullptr)
    return DEGA_ERROR_INVALID_VALUE;
  if ((ret = check_shape64(ctx, C, max_T, ld, cap, valuesize)) != DEGA_OK)
    return ret;
  return launch_decode(ctx, in, cap, in_bits, shape_of(C, max_T, ld, adaptive, valuesize, DEGA_SAMPLES_I64), C, x_tc, out_count, err, (hipStream_t)stream);
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
    hipLaunchKernelGGL(lzmh_encode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZ_ENC_THREADS), 0, s, a);
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
    hipLaunchKernelGGL(lzmh_decode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZD_THREADS), 0, s, a);
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

// ---- host-pointer entry points: the pipeline and the multi-device group ------------------------------------------------------
#include "dega_pipeline.hpp"
