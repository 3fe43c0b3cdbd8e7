// dega_kernels.hpp -- the HIP kernels of the DEGA path for gfx950 (MI355X).
//
// Mapping: one lane = one meter channel, one wave = 64 adjacent channels, one workgroup = 4 waves (one per SIMD of a
// CU) sharing the 128 KiB division table in LDS.  Samples are [T][ld] (time-major), so a wave's row read is one
// coalesced 256-byte segment and every input byte is fetched exactly once.
//
// Encode (diff -> seg -> bac fused, nothing but the final stream ever goes back to HBM):
//   phase F ("fill", row lockstep)   every lane takes the same rows t..t+3, forms the delta, its signed exp-Golomb
//                                    codeword and appends it to its private bit queue; finished 32-bit words go to the
//                                    lane's column of an LDS ring.
//   phase C ("code", word lockstep)  every lane that has a whole word queued takes it and codes its 32 bits with the
//                                    adaptive binary arithmetic coder -- 32 unrolled, divergence-free symbol steps.
// Channels need different numbers of coded bits per row; the ring decouples the two lockstep domains, so the wave
// makes exactly max-over-lanes(words) phase-C steps: the slowest channel of a wave sets its time, as it must
// (the coder is serial per channel), and nothing is wasted on codeword-length divergence inside a row.
//
// This header is compiled by hipcc (dega_hip.hip) and, for offline debugging only, by g++ under tests/sim/.
#pragma once

#include "dega_lane.hpp"

#include <stddef.h>

namespace dg
{

#if !defined(DEGA_SIM)
DG_DEV bool wave_any(bool p)
{
  return __any((int)p) != 0;
}
DG_DEV bool wave_all(bool p)
{
  return __all((int)p) != 0;
}
#endif

#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
// diagnostic build: per-wave cycle totals per section of the encode loop (s_memtime), dumped over out_bits[] / err[]
#define DG_STAMP_DECL uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DG_STAMP(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); stamp_sum[k] += now_ - stamp_t0; stamp_cnt[k]++; stamp_t0 = now_; } while (0)
#else
#define DG_STAMP_DECL
#define DG_STAMP(k)
#endif

constexpr uint32_t BLOCK = 256;
constexpr uint32_t WAVES = BLOCK / 64;

constexpr uint32_t ENC_RING = 16;                               // queued words per lane (LDS: 4 KiB per wave)
constexpr uint32_t ENC_ROWS = 4;                                // rows per fill batch
constexpr uint32_t ENC_FILL_WORDS = (31 + 65 * ENC_ROWS) / 32;  // most words a batch can add (65-bit worst-case codewords)
static_assert(ENC_FILL_WORDS < ENC_RING, "ring too small");

struct EncodeArgs
{
  const int32_t *x; // [T][ld]
  size_t C, T, ld;
  uint8_t *out;     // [C][cap]
  size_t cap;       // bytes per channel, multiple of 4
  uint64_t *out_bits;
  int32_t *err;
  const uint32_t *div_magic; // DIV_TABLE_SIZE division magics (dega_lane.hpp), global memory
};

template <bool ADAPTIVE>
DG_DEV void load_div_table(uint32_t *tab, const uint32_t *gtab)
{
  if (ADAPTIVE)
  {
    for (uint32_t i = threadIdx.x; i < DIV_TABLE_SIZE; i += BLOCK)
      tab[i] = gtab[i];
  }
  else if (threadIdx.x < 4)
    tab[threadIdx.x] = gtab[threadIdx.x]; // the static model never leaves cum[0] = 3
  __syncthreads();
}

template <bool ADAPTIVE>
__global__ void __launch_bounds__(256) dega_encode_kernel(const EncodeArgs a)
{
  __shared__ uint32_t tab[ADAPTIVE ? DIV_TABLE_SIZE : 4]; // 64 KiB
  __shared__ uint32_t ring[WAVES * ENC_RING * 64];   // seg bits waiting to be coded, per lane
  __shared__ uint32_t oring[WAVES * ENC_ORING * 64]; // coded words waiting to be stored, per lane
  __shared__ uint32_t xrows[WAVES * ENC_ROWS * 64];  // the next input rows, per lane

  load_div_table<ADAPTIVE>(tab, a.div_magic);

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool live = c < a.C;
  uint32_t *const ring_col = &ring[wave * ENC_RING * 64 + lane];
  const int32_t *const col = a.x + (live ? c : 0);

  BacEncoder<ADAPTIVE> enc;
  enc.init(live ? reinterpret_cast<uint32_t *>(a.out + c * a.cap) : nullptr, live ? (uint32_t)(a.cap / 4) : 0u,
           &oring[wave * ENC_ORING * 64 + lane]);
  BitQueue q;
  q.init();
  uint32_t last = 0; // diff.c:11
  int32_t lane_err = OK;

  // Input rows travel  HBM -> registers (loads issued right after the fill step of every iteration) -> LDS (parked
  // there after the code step, by which time the loads have long landed) -> the next fill.  The loads and the parking
  // are unconditional (a step without a fill re-reads the same rows, from L2), so that on every path nothing is
  // pending at the loop's back edge: hipcc's wait-count pass is path insensitive and would otherwise put a vmcnt(0)
  // in the fill -- behind the drain's stores, which retire in order with the loads (measured: 26 % of all cycles).
  uint32_t *const rows_col = &xrows[wave * ENC_ROWS * 64 + lane];
  size_t t = 0; // rows consumed by fills, wave uniform; LDS always holds rows [t, t + ENC_ROWS)
  const size_t t_last = a.T > 0 ? a.T - 1 : 0;
#pragma unroll
  for (uint32_t i = 0; i < ENC_ROWS; i++)
    rows_col[i * 64u] = (live && a.T > 0) ? (uint32_t)col[(i < a.T ? i : t_last) * a.ld] : 0u;

  DG_STAMP_DECL;
  for (;;)
  {
    DG_STAMP(7);
    // ---- phase F: the same ENC_ROWS rows for every lane ----------------------------------------------------------
    const bool room = (q.wr - q.rd) + ENC_FILL_WORDS <= ENC_RING;
    const bool fill = t < a.T && wave_all(room);
    DG_STAMP(0);
    if (fill)
    {
      const size_t left = a.T - t;
      uint32_t xr[ENC_ROWS];
#pragma unroll
      for (uint32_t i = 0; i < ENC_ROWS; i++)
        xr[i] = rows_col[i * 64u];
#pragma unroll
      for (uint32_t i = 0; i < ENC_ROWS; i++)
      {
        if (i < left && live)
        {
          const SegWord s = diff_seg(xr[i], last);
          if (!s.ok && lane_err == OK)
            lane_err = ERR_INVALID_VALUE;
          q.put_codeword<ENC_RING>(s, ring_col);
        }
      }
      t += left < ENC_ROWS ? left : ENC_ROWS;
      DG_STAMP(1);
    }
    uint32_t xt[ENC_ROWS]; // rows [t, t + ENC_ROWS), clamped to the last row: in flight while phase C runs
#pragma unroll
    for (uint32_t i = 0; i < ENC_ROWS; i++)
    {
      const size_t row = t + i < a.T ? t + i : t_last;
#if defined(DEGA_DIAG) && (DEGA_DIAG & 2)
      xt[i] = (uint32_t)(row * 7u + lane); // diagnostic build: no input loads
#else
      xt[i] = live ? (uint32_t)col[row * a.ld] : 0u;
#endif
    }
    // ---- phase C: one queued word (32 symbols) for every lane that has one ---------------------------------------
    const bool has = q.wr != q.rd;
    const bool any_has = wave_any(has);
    if (!any_has && !fill)
      break; // all rows consumed and every queue drained
    DG_STAMP(2);
    if (any_has)
    {
      const bool fast = wave_all(!has || enc.fast_ok()); // wave uniform: the unrolled branch-free word, or bit by bit
      if (has)
      {
        const uint32_t word = ring_col[(q.rd % ENC_RING) * 64u];
        q.rd++;
        bool done = false;
        if (fast)
        {
          const BacEncoder<ADAPTIVE> checkpoint = enc;
          done = enc.encode_word_fast(word, tab);
          if (!done)
            enc = checkpoint; // a carry ran past the held-back word (33+ pending bits): redo exactly
        }
        if (!done)
        {
#pragma unroll 1
          for (uint32_t i = 0; i < 32; i++)
            enc.encode_bit((word >> (31u - i)) & 1u, tab);
        }
      }
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
      if (fast)
        DG_STAMP(3);
      else
        DG_STAMP(4);
#endif
    }
    // ---- park the rows loaded during this step (the compiler's vmcnt wait lands here, ahead of the drain's stores) --
#pragma unroll
    for (uint32_t i = 0; i < ENC_ROWS; i++)
      rows_col[i * 64u] = xt[i];
    DG_STAMP(5);
    // ---- drain: staged words -> slabs, all lanes in lockstep -----------------------------------------------------
    {
      const uint32_t n = enc.staged;
      const uint32_t w0 = enc.oring[0], w1 = enc.oring[64]; // nearly always 0..2 words per step: fetch both at once
#if defined(DEGA_DIAG) && (DEGA_DIAG & 1)
      if (n > 100) // diagnostic build: no output stores
        enc.put_word(enc.drained, w0 + w1);
#else
      if (n > 0)
        enc.put_word(enc.drained, w0);
      if (n > 1)
        enc.put_word(enc.drained + 1u, w1);
      for (uint32_t s = 2; wave_any(s < n); s++)
        if (s < n)
          enc.put_word(enc.drained + s, enc.oring[s * 64u]);
#endif
      enc.drained += n;
      enc.staged = 0;
    }
    DG_STAMP(6);
  }

  if (live)
  {
    // the last, partial word of the seg stream, then EOF + flush (bac.c:163-164)
    const uint32_t tail = q.cnt;
    const uint32_t word = tail ? (uint32_t)(q.acc << (32u - tail)) : 0u;
    for (uint32_t i = 0; i < tail; i++)
      enc.encode_bit((word >> (31u - i)) & 1u, tab);
    a.out_bits[c] = enc.finish(tab);
    a.err[c] = lane_err != OK ? lane_err : enc.err;
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
    if (lane < 8)
      a.out_bits[c] = stamp_sum[lane];
    else if (lane < 16)
      a.out_bits[c] = stamp_cnt[lane - 8];
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// normalize / denormalize (DCLib/src/normalize.c), elementwise, HBM bound.  Float parity rules: the
// multiply and the +-0.5 are two separately rounded operations (no FMA), truncating convert, IEEE division.
// ---------------------------------------------------------------------------------------------------------------------
DG_DEV bool normalize_value(float v, float factor, int32_t &out)
{
#if defined(DEGA_SIM)
  volatile float prod;
  if (v > 0.0f)
  {
    prod = v * factor;
    v = prod + 0.5f;
  }
  else if (v < 0.0f)
  {
    prod = v * factor;
    v = prod - 0.5f;
  }
#else
  if (v > 0.0f)
    v = __fadd_rn(__fmul_rn(v, factor), 0.5f); // normalize.c:17-18
  else if (v < 0.0f)
    v = __fsub_rn(__fmul_rn(v, factor), 0.5f); // :19-20
#endif
  const bool ok = !(v < -2147483648.0f || v > 2147483648.0f); // :21 -- (float)(2^31-1) is 2^31, so exactly 2^31 passes
  out = v >= 2147483648.0f ? (int32_t)0x80000000u : (int32_t)v; // :23 (int64) truncation, low 32 bits
  return ok;
}

struct NormalizeArgs
{
  const float *v;
  int32_t *x;
  size_t C, T, ld;
  float factor;
  int32_t *err; // [C], pre-zeroed
};

__global__ void __launch_bounds__(256) dega_normalize_kernel(const NormalizeArgs a)
{
  const size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= a.C)
    return;
  const size_t rows_per_block = (a.T + gridDim.y - 1) / gridDim.y;
  const size_t t0 = (size_t)blockIdx.y * rows_per_block;
  const size_t t1 = t0 + rows_per_block < a.T ? t0 + rows_per_block : a.T;
  bool bad = false;
  for (size_t t = t0; t < t1; t++)
  {
    int32_t n;
    bad |= !normalize_value(a.v[t * a.ld + c], a.factor, n);
    a.x[t * a.ld + c] = n;
  }
  if (bad)
    a.err[c] = ERR_INVALID_VALUE;
}

struct DenormalizeArgs
{
  const int32_t *x;
  float *v;
  size_t C, T, ld;
  float factor;
};

__global__ void __launch_bounds__(256) dega_denormalize_kernel(const DenormalizeArgs a)
{
  const size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= a.C)
    return;
  const size_t rows_per_block = (a.T + gridDim.y - 1) / gridDim.y;
  const size_t t0 = (size_t)blockIdx.y * rows_per_block;
  const size_t t1 = t0 + rows_per_block < a.T ? t0 + rows_per_block : a.T;
  for (size_t t = t0; t < t1; t++)
  {
#if defined(DEGA_SIM)
    a.v[t * a.ld + c] = (float)a.x[t * a.ld + c] / a.factor;
#else
    a.v[t * a.ld + c] = __fdiv_rn((float)a.x[t * a.ld + c], a.factor); // normalize.c:38, true division
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// synthetic load profiles
// ---------------------------------------------------------------------------------------------------------------------
DG_DEV uint64_t mix64(uint64_t z) // splitmix64 finaliser
{
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct SynthArgs
{
  int32_t *x;
  size_t C, T, ld;
  uint64_t seed, c0;
  uint32_t S;
};

__global__ void __launch_bounds__(256) dega_synth_kernel(const SynthArgs a)
{
  const size_t c = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (c >= a.C)
    return;
  const uint64_t ch = a.c0 + c;
  const uint64_t key = mix64(a.seed ^ (ch * 0xD1342543DE82EF95ull));
  int64_t v = 10000 + (int64_t)(mix64(key) % 50000u);
  const uint32_t span = 2u * a.S + 1u;
  for (size_t t = 0; t < a.T; t++)
  {
    if (t > 0)
    {
      v += (int64_t)(mix64(key + t) % span) - (int64_t)a.S;
      v = v < 0 ? 0 : (v > 2147483647ll ? 2147483647ll : v);
    }
    a.x[t * a.ld + c] = (int32_t)v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// stream compaction
// ---------------------------------------------------------------------------------------------------------------------
struct GatherArgs
{
  const uint8_t *slabs;
  size_t cap;
  const uint64_t *offsets; // [C+1]
  size_t C;
  uint8_t *packed;
};

// one wave per channel, byte granular (offsets are arbitrary), coalesced over the channel's bytes
__global__ void __launch_bounds__(256) dega_gather_kernel(const GatherArgs a)
{
  const size_t c = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (c >= a.C)
    return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t o0 = a.offsets[c], o1 = a.offsets[c + 1];
  const uint8_t *src = a.slabs + c * a.cap;
  uint8_t *dst = a.packed + o0;
  for (uint64_t i = lane; i < o1 - o0; i += 64)
    dst[i] = src[i];
}

} // namespace dg
