// dega_kernels.hpp -- the HIP kernels of the DEGA path for gfx950 (MI355X).
//
// Mapping: one lane = one meter channel, one wave = 64 adjacent channels; a workgroup = 4 (or 8) PAIRS of waves sharing
// the 64 KiB table of division magics in LDS -- of each pair one wave does the serial arithmetic, its partner on the same
// SIMD the work around it (see the encode and decode kernels).  Samples are [T][ld] (time-major), so a wave's row read is
// one coalesced 256-byte segment and every input byte is fetched exactly once.
//
// Encode (diff -> seg -> bac fused, nothing but the final stream ever goes back to HBM):
//   filling wave (row lockstep)   every lane takes the same rows t..t+7, forms the delta, its signed exp-Golomb codeword
//                                 and appends it to its private bit queue; finished 32-bit words go to the lane's column
//                                 of an LDS ring.
//   coding wave (word lockstep)   every lane that has a whole word queued takes it and codes its 32 bits with the
//                                 adaptive binary arithmetic coder -- 32 unrolled, divergence-free symbol steps.
// Channels need different numbers of coded bits per row; the ring decouples the two lockstep domains, so the coding wave
// makes exactly max-over-lanes(words) steps: the slowest channel of a wave sets its time, as it must (the coder is serial
// per channel), and nothing is wasted on codeword-length divergence inside a row.
//
// This header is compiled by hipcc (dega_hip.hip) and, for offline debugging only, by g++ under tests/sim/.
#pragma once

#include "dega_lane.hpp"

#include <stddef.h>

#include <type_traits>

namespace dg
{

// lo = -(float)2^(valuesize-1), hi = (float)(2^(valuesize-1) - 1) as the host's C compiler rounds them (normalize.c:21;
// hi rounds up to 2^(valuesize-1) from valuesize 26 on), mask = the low valuesize bits written (normalize.c:24)
DG_DEV bool normalize_value(float v, float factor, int32_t &out, float lo = -2147483648.0f, float hi = 2147483648.0f, uint32_t mask = 0xFFFFFFFFu)
{
  if (v > 0.0f)
    v = fadd_once(fmul_once(v, factor), 0.5f); // normalize.c:17-18: two operations, two roundings
  else if (v < 0.0f)
    v = fadd_once(fmul_once(v, factor), -0.5f); // :19-20
  const bool ok = !(v < lo || v > hi); // :21 -- for valuesize 32 (float)(2^31-1) is 2^31, so exactly 2^31 passes
  out = (int32_t)((v >= 2147483648.0f ? 0x80000000u : (uint32_t)(int32_t)v) & mask); // :23-24 (int64) truncation, low valuesize bits
  return ok;
}


// The same for valuesize 33..64: the normalized value is an io_int_t = int64 (normalize.c:23), written as its low
// valuesize bits.  v >= 2^63 (reachable only through the rounded upper bound at valuesize 64) converts like x86's
// cvttss2si does: to the "integer indefinite" 0x8000000000000000.
DG_DEV bool normalize_value64(float v, float factor, uint64_t &out, float lo, float hi, uint64_t mask)
{
  if (v > 0.0f)
    v = fadd_once(fmul_once(v, factor), 0.5f);
  else if (v < 0.0f)
    v = fadd_once(fmul_once(v, factor), -0.5f);
  const bool ok = !(v < lo || v > hi);
  const bool big = v >= 9223372036854775808.0f || v < -9223372036854775808.0f || v != v;
  out = (big ? 0x8000000000000000ull : (uint64_t)(int64_t)v) & mask;
  return ok;
}

DG_DEV float denormalize_value(float n, float factor) // normalize.c:38: true IEEE division, no reciprocal
{
  return fdiv_once(n, factor);
}

constexpr uint32_t BLOCK = 256;
constexpr uint32_t WAVES = BLOCK / 64;

constexpr uint32_t ENC_RING = 32;                               // queued words per lane (LDS: 8 KiB per pair)
constexpr uint32_t ENC_ROWS = 8;                                // rows per fill batch
constexpr uint32_t ENC_FILL_WORDS = (31 + 65 * ENC_ROWS) / 32;  // most words a batch can add (65-bit worst-case codewords)
static_assert(ENC_FILL_WORDS < ENC_RING, "ring too small");

// A channel can be coded over several launches, each taking the next range of rows (the host pipeline uploads a batch of
// few, long channels in bands and codes every band as it lands; a stream longer than one call's 2^25 samples): the state
// of a lane's three state machines between two launches, ENC_STATE_WORDS dwords per channel, [word][channel].
constexpr uint32_t ENC_STATE_WORDS = 18;
constexpr uint32_t ENC_SEG_CONTINUES = 1; // the launch goes on from saved state
constexpr uint32_t ENC_SEG_MORE = 2;      // more rows follow in a later launch: no EOF symbol, state saved

struct EncodeArgs
{
  const int32_t *x; // [T][ld]: the rows of this launch
  size_t C, T, ld;
  uint8_t *out;     // [C][cap]
  size_t cap;       // bytes per channel, multiple of 4
  uint64_t *out_bits;
  int32_t *err;
  const uint32_t *div_magic; // DIV_TABLE_SIZE division magics (dega_lane.hpp), global memory
  uint32_t valuesize;        // 1..32: samples are the low valuesize bits of x, unsigned (diff.c:15)
  uint32_t big_endian;       // 32-bit samples arrive byte swapped (the big-endian words `encode normalize` writes, bit_file_buffer.c:297-308)
  // F32IN variants (normalize fused into the fill phase, normalize.c:9-27): x holds raw float32 bits
  float factor, lo, hi;      // normalization factor; range of normalize.c:21 for the value size, rounded to float by the host compiler
  uint32_t *seg_state;       // NULL: whole channels in one launch.  Else [ENC_STATE_WORDS][C], see above
  uint32_t seg_flags;        // ENC_SEG_*
};

// `symbols`: no channel of the batch codes more symbols than this (cum[0] starts at 3 and grows by one per symbol until
// it halves at MAX_FREQUENCY): short channels need only the beginning of the table
template <bool ADAPTIVE>
DG_DEV void load_div_table(uint32_t *tab, const uint32_t *gtab, uint64_t symbols)
{
  if (ADAPTIVE)
  {
    const uint32_t words = symbols < DIV_TABLE_SIZE - 68u ? (uint32_t)symbols + 68u : DIV_TABLE_SIZE; // + the fetch-ahead of a word
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x)
      tab[i] = gtab[i];
  }
  else if (threadIdx.x < 4)
    tab[threadIdx.x] = gtab[threadIdx.x]; // the static model never leaves cum[0] = 3
  __syncthreads();
}

// 64 channels are the work of THREE waves that the CU places on the same SIMD (waves g, g + 4, g + 8 of a workgroup of
// twelve):
//   the coding wave   seg-bit words through the arithmetic coder, nothing else;
//   the filling wave  rows in (LDS-DMA), normalize, diff, seg -> bit words into the lane's column of an LDS ring;
//   the writing wave  the coder's raw entries -> 32-bit words -> the slab.
// A channel's coder is serial, and what bounds it is the instruction stream of ITS wave: one wave issues an instruction
// every 4.1 - 4.6 cycles whatever shares the SIMD with it, while two waves together get ~3.2 cycles per instruction out of
// the SIMD, three ~2.9, four ~2.5 (profiles/r03_ubench2_issue_cost.txt).  64 Ki channels are one coding wave per SIMD:
// every instruction that is not the arithmetic itself is worth moving into a helper, whose instructions go into issue
// slots the coder cannot use -- up to the point where the SIMD as a whole is full, which is where the kernel is now
// (VALUBusy 100 %: every helper instruction costs the coder a little).
// The waves talk through one published word per lane each (peer_store / peer_load); a wave with nothing to do sleeps.
//   filler publishes: ring words written (mod 2^16) | all rows done << 24 | a value was out of range << 25 |
//                     bits of the final, partial word << 27
//   writer publishes: raw entries absorbed (mod 2^8) << 16
//   coder publishes:  ring words consumed (mod 2^16) | raw entries written (mod 2^8) << 16 | all entries written << 24 |
//                     the filler's verdict << 25
// ROWS rows per fill batch; RING / RAW / ORING: seg-bit words, raw entries, staged output words per lane.
// W64: valuesize 33..64 -- a.x is int64 [T][ld]; rows travel as two dwords per lane, the fill step takes the general
// writer (127-bit worst-case codewords), everything behind the bit queue is the same.  Instantiated with ROWS = 4.
// F32IN: the rows are float32 readings; Normalize (normalize.c:16-24) runs on each value as it leaves LDS, in front of
// the difference -- one launch, no int32 intermediate in HBM.  With W64 the rows stay one dword per lane (floats) and
// the normalized value is 64 bits wide (valuesize 33..64, normalize.c:21-24 with io_int_t = int64).
// How the two waves of a pair share their SIMD: how long a wave with nothing to do sleeps (units of 64 cycles) and the
// issue priority of the coding waves.  Compile-time, so that tools/tunebench.py can time variants side by side.
#ifndef DG_ENC_FILL_SLEEP
#define DG_ENC_FILL_SLEEP 12
#endif
#ifndef DG_ENC_WRITE_SLEEP
#define DG_ENC_WRITE_SLEEP 6
#endif
#ifndef DG_ENC_CODE_SLEEP
#define DG_ENC_CODE_SLEEP 1
#endif
#ifndef DG_ENC_FILL_PRIO
#define DG_ENC_FILL_PRIO 0
#endif
#ifndef DG_ENC_WRITE_PRIO
#define DG_ENC_WRITE_PRIO 1
#endif
#ifndef DG_ENC_CODE_PRIO
#define DG_ENC_CODE_PRIO 3
#endif
#ifndef DG_DEC_PARSE_SLEEP
#define DG_DEC_PARSE_SLEEP 6
#endif
#ifndef DG_DEC_CODE_SLEEP
#define DG_DEC_CODE_SLEEP 1
#endif
#ifndef DG_DEC_LOAD_SLEEP
#define DG_DEC_LOAD_SLEEP 8
#endif
#ifndef DG_DEC_PARSE_PRIO
#define DG_DEC_PARSE_PRIO 0
#endif
#ifndef DG_DEC_LOAD_PRIO
#define DG_DEC_LOAD_PRIO 0
#endif
#ifndef DG_DEC_CODE_PRIO
#define DG_DEC_CODE_PRIO 0
#endif
#ifndef DG_DEC_TAKES // unconditional short-codeword takes per pass of the parsing wave
#define DG_DEC_TAKES 4
#endif
constexpr uint32_t ENC_PAIRS = 4;                 // groups of three waves per workgroup
constexpr uint32_t ENC_BLOCK = ENC_PAIRS * 192;   // threads per workgroup: a filling, a coding and a writing wave per 64 channels
constexpr uint32_t ENC_CHANNELS = ENC_PAIRS * 64; // channels per workgroup
constexpr uint32_t ENC_PUB_DONE = 1u << 24, ENC_PUB_BAD = 1u << 25;

// ---- the helping wave(s): fill and write -------------------------------------------------------------------------------
// FILLS / WRITES: which of the two this wave is.  (A wave each: the coding wave of a SIMD must never wait for either, and
// three waves get more instructions per cycle out of a SIMD than two -- one wave doing both left the coder waiting for
// room a sixth of its time.)
template <bool NARROW, uint32_t ROWS, uint32_t RING, uint32_t RAW, uint32_t ORING, bool W64, bool F32IN, bool FILLS, bool WRITES>
DG_DEV void encode_helping_wave(const EncodeArgs &a, uint32_t *ring_col, const uint32_t *raw_col, uint32_t *oring_col, uint32_t *rows_wave, uint32_t *pub_mine,
                                const uint32_t *pub_peer, uint32_t lane, size_t c, bool live, size_t c_wave0)
{
  static_assert(FILLS != WRITES, "a helper fills or writes");
  // NARROW: valuesize < 32 -- the samples are masked to valuesize bits and the difference is range checked against it
  constexpr uint32_t FILL_WORDS = (31 + (W64 ? 127 : 65) * ROWS) / 32; // most words a batch can add (worst-case codewords)
  constexpr bool ROWS64 = W64 && !F32IN; // rows of two dwords per lane
  constexpr uint32_t GROUP = ORING / 2;  // staged words stored together: 64 bytes for the narrow batches' 32-word staging ring
  static_assert(FILL_WORDS < RING, "ring too small");
  static_assert(GROUP % 4 == 0 && GROUP >= 4, "whole 16-byte stores");
  const uint32_t vmask = NARROW ? (1u << (a.valuesize & 31u)) - 1u : 0xFFFFFFFFu, vhalf = NARROW ? 1u << ((a.valuesize - 1u) & 31u) : 0x80000000u;
  const bool continues = a.seg_state != nullptr && (a.seg_flags & ENC_SEG_CONTINUES) != 0u;
  const bool more = a.seg_state != nullptr && (a.seg_flags & ENC_SEG_MORE) != 0u;
  uint32_t *const state = a.seg_state != nullptr && live ? a.seg_state + c : nullptr; // word k at state[k * C]

  BitQueue q;
  q.init();
  uint32_t last = 0; // diff.c:11
  uint64_t last64 = 0;
  int32_t lane_err = OK;
  BacWriter<ORING> wr;
  wr.init(live ? reinterpret_cast<uint32_t *>(a.out + c * a.cap) : nullptr, live ? (uint32_t)(a.cap / 4) : 0u, oring_col);
  if (continues && live)
  {
    q.acc = ((uint64_t)state[1 * a.C] << 32) | state[0];
    q.cnt = state[2 * a.C];
    last = state[3 * a.C];
    last64 = ((uint64_t)state[4 * a.C] << 32) | last;
    lane_err = (int32_t)state[5 * a.C];
    wr.err = (int32_t)state[16 * a.C];
    wr.F = ((uint64_t)state[7 * a.C] << 32) | state[6 * a.C];
    wr.fcnt = state[8 * a.C];
    wr.prev = state[9 * a.C];
    wr.pos = state[10 * a.C];
    wr.drained = wr.pos > 0u ? wr.pos - 1u : 0u; // everything but the held-back word is in the slab
  }
  uint32_t rrd = 0; // raw entries absorbed

  // Input rows travel HBM -> LDS directly (LDS-DMA, `global_load_lds_dword`: one 256-byte row segment per wave
  // instruction, no VGPR destination) and are read from LDS by the next fill; the next batch is requested right after a
  // fill and has the coder's next two or three steps to arrive.
  const uint32_t *const rows_col = rows_wave + lane;
  size_t t = 0; // rows consumed by fills, wave uniform; after the wait LDS holds rows [t, t + ROWS)
  const size_t t_last = a.T > 0 ? a.T - 1 : 0;
  const int32_t *const last_row = a.x + t_last * a.ld;
  const uint32_t col_idx = live ? (uint32_t)c : 0u; // C <= 2^32 columns

  // When the wave's 64 channels all exist and rows are 16-byte aligned, one LDS-DMA instruction fetches FOUR rows:
  // lanes 16r .. 16r+15 read row r's 256 bytes as 16-byte pieces, which land as row r of the [row][64] LDS image.
  const bool rows_x4 = !ROWS64 && (ROWS % 4 == 0) && c_wave0 + 64 <= a.C && (a.ld % 4 == 0) && (((size_t)a.x) % 16 == 0);
  auto issue_rows = [&](size_t t0) // rows [t0, t0 + ROWS), clamped to the last row
  {
    if constexpr (ROWS64)
    {
      const int64_t *const x64 = reinterpret_cast<const int64_t *>(a.x);
#pragma unroll
      for (uint32_t i = 0; i < ROWS; i++)
      {
        const size_t row = t0 + i < a.T ? t0 + i : t_last;
        const int32_t *const p = reinterpret_cast<const int32_t *>(x64 + row * a.ld) + 2u * (size_t)col_idx;
        if (live)
        {
          dma_row_to_lds(p, rows_wave + (2u * i) * 64u, lane);          // low dword
          dma_row_to_lds(p + 1, rows_wave + (2u * i + 1u) * 64u, lane); // high dword
        }
      }
      return;
    }
    if (rows_x4)
    {
      const uint32_t r = lane >> 4, q4 = lane & 15u;
#pragma unroll
      for (uint32_t j = 0; j < ROWS / 4; j++)
      {
        const size_t row = t0 + 4 * j + r < a.T ? t0 + 4 * j + r : t_last;
        dma_x4_to_lds(a.x + row * a.ld + c_wave0 + q4 * 4u, rows_wave + j * 256u, lane);
      }
      return;
    }
    const int32_t *rowp = a.x + t0 * a.ld; // wave uniform; the lane adds its 32-bit column index
#pragma unroll
    for (uint32_t i = 0; i < ROWS; i++)
    {
      const int32_t *const r = t0 + i < a.T ? rowp : last_row;
      if (live)
        dma_row_to_lds(r + col_idx, rows_wave + i * 64u, lane);
      rowp += a.ld;
    }
  };

  // ---- one batch of ROWS rows (or what is left of them) into the lanes' bit queues -------------------------------------------
  auto fill_batch = [&]() {
    const size_t left = a.T - t;
    if constexpr (W64)
    {
      const uint64_t vmask64 = a.valuesize >= 64u ? ~0ull : (1ull << a.valuesize) - 1ull;
      if (live)
      {
#pragma unroll
        for (uint32_t i = 0; i < ROWS; i++)
          if (i < left)
          {
            uint64_t u;
            if constexpr (F32IN)
            {
              if (!normalize_value64(__uint_as_float(rows_col[i * 64u]), a.factor, u, a.lo, a.hi, vmask64) && lane_err == OK)
                lane_err = ERR_INVALID_VALUE; // normalize.c:21-22
            }
            else
              u = (((uint64_t)rows_col[(2u * i + 1u) * 64u] << 32) | rows_col[(2u * i) * 64u]) & vmask64;
            const SegWord64 sw = diff_seg64(u, last64, a.valuesize);
            if (!sw.ok && lane_err == OK)
              lane_err = ERR_INVALID_VALUE;
            q.put_codeword64<RING>(sw, ring_col);
          }
      }
    }
    else
    {
      uint32_t xr[ROWS];
#pragma unroll
      for (uint32_t i = 0; i < ROWS; i++)
        xr[i] = rows_col[i * 64u];
#pragma unroll
      for (uint32_t i = 0; i < ROWS; i++)
        DG_MATERIALISE(xr[i]); // one LDS wait here, none between the ring writes below
      if constexpr (F32IN)
      {
        // Normalize (normalize.c:16-24) on the way in; rows past the end of the channel repeat the last row, whose
        // verdict is the same, so the whole batch can be checked
        bool all_in_range = true;
#pragma unroll
        for (uint32_t i = 0; i < ROWS; i++)
        {
          int32_t n;
          all_in_range = normalize_value(__uint_as_float(xr[i]), a.factor, n, a.lo, a.hi, vmask) && all_in_range;
          xr[i] = (uint32_t)n;
        }
        if (!all_in_range && lane_err == OK)
          lane_err = ERR_INVALID_VALUE;
      }
      else
      {
        if (a.big_endian) // wave uniform
        {
#pragma unroll
          for (uint32_t i = 0; i < ROWS; i++)
            xr[i] = bswap32(xr[i]);
        }
        if (NARROW)
        {
#pragma unroll
          for (uint32_t i = 0; i < ROWS; i++)
            xr[i] &= vmask;
        }
      }
      // Pass 1, no side effects: the codeword values of the whole batch, assuming the steady state -- every sample
      // below 2^31 (then every difference fits, diff.c:17-18) and every codeword short (|delta| < 2^15): one OR over
      // the batch answers both
      uint32_t w[ROWS];
      uint32_t last_try = last, seen = last, wseen = 0;
#pragma unroll
      for (uint32_t i = 0; i < ROWS; i++)
      {
        w[i] = diff_seg_steady(xr[i], last_try);
        seen |= xr[i];
        wseen |= w[i];
      }
      const bool plain = NARROW ? false : ((seen >> 31) | (wseen >> 16)) == 0u; // narrow values: range check per sample
      if (left >= ROWS && !wave_any(!plain && live))
      {
        // the steady state: a full batch of short codewords, straight-line appends
        if (live)
        {
          last = last_try;
#pragma unroll
          for (uint32_t i = 0; i < ROWS; i++)
            q.put_short<RING>(w[i], ring_col);
        }
      }
      else if (live)
      {
        // first samples of a channel, jumps, narrow value sizes, the last partial batch: the general three-piece
        // writer, row by row
#pragma unroll
        for (uint32_t i = 0; i < ROWS; i++)
        {
          if (i < left)
          {
            const SegWord sw = diff_seg<NARROW>(xr[i], last, vhalf);
            if (!sw.ok && lane_err == OK)
              lane_err = ERR_INVALID_VALUE;
            q.put_codeword<RING>(sw, ring_col);
          }
        }
      }
    }
    t += left < ROWS ? left : ROWS;
  };

  wave_priority<(WRITES ? DG_ENC_WRITE_PRIO : DG_ENC_FILL_PRIO)>();
  if (FILLS && a.T > 0)
    issue_rows(0);
  bool tail_placed = !FILLS; // the final, partial word is in the ring (or stays in the state) and "done" is published
  uint32_t pub_flags = 0;
  for (;;)
  {
    const uint32_t cp = peer_load(pub_peer);
    bool worked = false;
    // ---- fill: the same ROWS rows for every lane, as soon as every lane's ring has room for what they may add -----------
    if (FILLS && t < a.T)
    {
      const bool room = ((q.wr - cp) & 0xFFFFu) + FILL_WORDS <= RING;
      if (wave_all(room))
      {
        wait_vector_memory();
        fill_batch();
        if (t < a.T)
          issue_rows(t); // in flight while the coder works through this batch
        worked = true;
      }
    }
    // The last, partial word of the seg stream goes into the next ring slot, left aligned -- once that slot is free: a
    // batch of worst-case codewords can leave the ring full to the last slot (RING words queued), and the next slot is
    // then the oldest word the coder has not taken yet.  When more rows follow in a later launch the bits stay in the
    // queue (saved with the state) and the coder gets no partial word.
    if (FILLS && t >= a.T && !tail_placed && !wave_any(((q.wr - cp) & 0xFFFFu) >= RING))
    {
      uint32_t tail_bits = 0;
      if (!more)
      {
        ring_col[(q.wr % RING) * 64u] = q.cnt != 0u ? (uint32_t)(q.acc << (32u - q.cnt)) : 0u;
        tail_bits = q.cnt;
      }
      pub_flags = ENC_PUB_DONE | (tail_bits << 27) | (lane_err != OK ? ENC_PUB_BAD : 0u);
      tail_placed = true;
      worked = true;
    }
    // ---- write: the coder's raw entries, oldest first, all lanes in step ---------------------------------------------------
    uint32_t avail = WRITES ? ((cp >> 16) - rrd) & 0xFFu : 0u;
    if (WRITES && wave_any(avail != 0u))
    {
      uint32_t budget = GROUP; // entries per pass: the staging ring takes a group's worth on top of what waits to be stored
      // The steady state: every lane has the four entries of a fast word step, a word held back, and room in F for all
      // four -- one address, four reads in flight, straight-line code for all 64 lanes, one hand-over.
      if (wave_all(avail >= 4u && wr.pos != 0u))
      {
        const uint32_t *const slot = raw_col + (rrd % RAW) * 64u;
        uint32_t hi[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
          hi[k] = slot[k * 64u];
        if (wave_all(wr.fits4(hi)))
        {
          const bool rippled = wr.absorb4_in_step(hi);
          if (wave_any(rippled))
          {
            if (rippled)
              wr.ripple_carry_from(wr.pos - 2u);
          }
          rrd += 4u;
          avail -= 4u;
          budget -= 4u;
        }
      }
      // the four entries of a fast word step together (the entry count stays a multiple of 4 until the stream ends:
      // BacCoder::end_bits_word): one address, four reads in flight, one hand-over
      while (budget >= 4u && wave_any(avail >= 4u))
      {
        const bool four = avail >= 4u;
        const uint32_t *const slot = raw_col + (rrd % RAW) * 64u;
        uint32_t hi[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
          hi[k] = four ? slot[(k % RAW) * 64u] : 2u;
        if (!wave_any(four && !wr.fits4(hi)))
        {
          if (four)
            wr.absorb4(hi);
        }
        else if (four)
        {
#pragma unroll 1
          for (uint32_t k = 0; k < 4; k++)
            wr.absorb(hi[k]);
        }
        rrd += four ? 4u : 0u;
        avail -= four ? 4u : 0u;
        budget -= 4u;
      }
      while (budget != 0u && wave_any(avail != 0u && avail < 4u)) // the last entries of a stream
      {
        if (avail != 0u && avail < 4u)
        {
          wr.absorb(raw_col[(rrd % RAW) * 64u]);
          rrd++;
          avail--;
        }
        budget--;
      }
      // whole groups of staged words -> the slab: consecutive 16-byte stores, so that a lane's stores fill whole 64-byte
      // segments of its slab (16-byte stores scattered in time left 128-byte lines half written: 1.6 x the stream bytes
      // went to HBM)
      while (wave_any(wr.staged >= GROUP))
      {
        if (wr.staged >= GROUP)
        {
          uint32_t g[GROUP];
#pragma unroll
          for (uint32_t k = 0; k < GROUP; k++)
            g[k] = wr.oring[((wr.drained + k) % ORING) * 64u];
#if defined(DEGA_DIAG) && (DEGA_DIAG & 1)
          if (g[0] == 0x12345u) // diagnostic build: no output stores
#endif
          {
#pragma unroll
            for (uint32_t k = 0; k < GROUP; k += 4)
              wr.put_group(wr.drained + k, g[k], g[k + 1], g[k + 2], g[k + 3]);
          }
          wr.drained += GROUP;
          wr.staged -= GROUP;
        }
      }
      worked = true;
    }
    peer_store(pub_mine, (q.wr & 0xFFFFu) | ((rrd & 0xFFu) << 16) | pub_flags);
    // through: the tail is placed (filling), the coder has written its last entry and every entry is absorbed (writing)
    if (tail_placed && (!WRITES || wave_all((cp & ENC_PUB_DONE) != 0u && (((cp >> 16) - rrd) & 0xFFu) == 0u)))
      break;
    if (!worked)
      wave_sleep<(WRITES ? DG_ENC_WRITE_SLEEP : DG_ENC_FILL_SLEEP)>(); // (a poll costs a dozen instructions of a SIMD that has none to spare)
  }
  if (FILLS)
    wait_vector_memory(); // no DMA may still be writing to LDS when the workgroup's allocation is released
  if (!live)
    return;
  if (more)
  {
    // the state machines' registers to the state (the coder saves its own); everything but the held-back word to the slab
    if (FILLS)
    {
      state[0] = (uint32_t)q.acc;
      state[1 * a.C] = (uint32_t)(q.acc >> 32);
      state[2 * a.C] = q.cnt;
      state[3 * a.C] = W64 ? (uint32_t)last64 : last;
      state[4 * a.C] = (uint32_t)(last64 >> 32);
      state[5 * a.C] = (uint32_t)lane_err;
    }
    if (WRITES)
    {
      wr.drain_lane();
      state[6 * a.C] = (uint32_t)wr.F;
      state[7 * a.C] = (uint32_t)(wr.F >> 32);
      state[8 * a.C] = wr.fcnt;
      state[9 * a.C] = wr.prev;
      state[10 * a.C] = wr.pos;
      state[16 * a.C] = (uint32_t)wr.err;
    }
  }
  else
  {
    // the last launch of a channel: the writer reports its length and the verdict -- a value out of range (the filler's
    // finding: diff.c:17-18, normalize.c:21; a filler in a wave of its own publishes it with "done" and the coder hands
    // it on with its own) before a slab too small
    if (WRITES)
    {
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
      (void)wr.finish();
#else
      a.out_bits[c] = wr.finish();
#endif
      const bool bad_value = FILLS ? lane_err != OK : (peer_load(pub_peer) & ENC_PUB_BAD) != 0u;
      a.err[c] = bad_value ? ERR_INVALID_VALUE : wr.err;
    }
  }
}

// ---- the coding wave -------------------------------------------------------------------------------------------------
template <bool ADAPTIVE, uint32_t RING, uint32_t RAW>
DG_DEV void encode_coding_wave(const EncodeArgs &a, const uint32_t *tab, const uint32_t *ring_col, uint32_t *raw_col, uint32_t *pub_mine, const uint32_t *pub_peer,
                               const uint32_t *pub_writer, uint32_t lane, size_t c, bool live)
{
  BacCoder<ADAPTIVE, RAW> enc;
  enc.init(raw_col, pub_writer);
  const bool continues = a.seg_state != nullptr && (a.seg_flags & ENC_SEG_CONTINUES) != 0u;
  const bool more = a.seg_state != nullptr && (a.seg_flags & ENC_SEG_MORE) != 0u;
  uint32_t *const state = a.seg_state != nullptr && live ? a.seg_state + c : nullptr;
  if (continues && live)
  {
    enc.L = ((uint64_t)2 << 32) | state[11 * a.C];
    enc.B = state[12 * a.C];
    enc.c1 = state[13 * a.C];
    enc.tot = state[14 * a.C];
    enc.mps = state[15 * a.C];
  }
  uint32_t rd = 0; // ring words coded so far
  // The helper's word as read one step ago: a ring word may only be read after a count that covers it, and waiting for
  // the count before asking for the word would put two LDS round trips at the head of every step.  One step late costs
  // nothing (the counts only grow, the ring holds a dozen words).
  uint32_t peer = peer_load(pub_peer), peerw = peer_load(pub_writer);
  enc.classify(); // the class of the first word
  wave_priority<DG_ENC_CODE_PRIO>();
  bool waiting = false; // (wave uniform) at low priority, nothing to do
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
  // diagnostic build: what the coding wave's passes were (steady / steady masked / other word paths / nothing to do: no word, no room), in cycles too
  uint64_t dg_n[4] = {0, 0, 0, 0}, dg_c[4] = {0, 0, 0, 0}, dg_t = __builtin_amdgcn_s_memtime();
#define DG_COUNT(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); dg_n[k]++; dg_c[k] += now_ - dg_t; dg_t = now_; } while (0)
#else
#define DG_COUNT(k)
#endif

  for (;;)
  {
    // ---- one queued word (32 symbols) for every lane that has one ------------------------------------------------------
    // What a code step reads from LDS first -- the helper's word, the queued word and the first quarter of its division
    // magics -- is asked for here, ahead of the ballots that choose the word path; nothing LDS is carried around the loop
    // (a wait at the back edge would be a wait for the whole queue).
    const bool has = ((peer - rd) & 0xFFFFu) != 0u;
    const bool all_done = (peer & ENC_PUB_DONE) != 0u;
    const bool room4 = enc.raw_room(enc.rwr, peerw, 4u), room8 = enc.raw_room(enc.rwr, peerw, 8u);
    const uint32_t word = ring_col[(rd % RING) * 64u];
    uint32_t Mg[32];
    enc.fetch_magics_first(tab, Mg);
    peer = peer_load(pub_peer);
    peerw = peer_load(pub_writer);
    // The steady state: every lane of the wave has a word, room for its four entries, and a fast class (every lane knows the
    // class of its NEXT word and for how many words it holds: BacCoder::classify, looked at again when the count has run out,
    // at the end of the step) -- one ballot, then straight-line code for all 64 lanes, no exec masks, no merges.
    const bool ready = has && room4;
    if (wave_all(ready && enc.cls == CLS_FAST8))
    {
      if (waiting)
      {
        wave_priority<DG_ENC_CODE_PRIO>();
        waiting = false;
      }
      enc.template encode_word<false, 8>(word, tab, Mg);
      rd++;
      enc.safe--;
      if (wave_any(enc.safe == 0u))
      {
        if (enc.safe == 0u)
          enc.classify();
      }
      peer_store(pub_mine, (rd & 0xFFFFu) | ((enc.rwr & 0xFFu) << 16));
      DG_COUNT(0);
      continue;
    }
    // The same with some lanes at a halving of the counts or a change of the division shift (CLS_SPLIT: the word in two
    // parts, see BacCoder): the masked fast path for all 64 lanes -- whole words for the others
    if constexpr (ADAPTIVE)
    {
      if (wave_all(ready && enc.cls <= CLS_SPLIT))
      {
        if (waiting)
        {
          wave_priority<DG_ENC_CODE_PRIO>();
          waiting = false;
        }
        enc.template encode_word<false, 8, true>(word, tab, Mg);
        const bool whole = enc.cls != CLS_SPLIT || enc.after_part(word);
        rd += whole ? 1u : 0u;
        enc.safe -= enc.safe != 0u ? 1u : 0u;
        const bool again = whole && enc.safe == 0u;
        if (wave_any(again))
        {
          if (again)
            enc.classify();
        }
        peer_store(pub_mine, (rd & 0xFFFFu) | ((enc.rwr & 0xFFu) << 16));
        DG_COUNT(1);
        continue;
      }
    }
    // (the slower word paths write eight entries; the bit path waits for room by itself).  A lane with a word but no room
    // for its entries holds the whole wave up: the writer never waits for this wave, so room comes, and the lanes stay in
    // step -- a wave whose lanes take turns needs more steps, each with its exec masks and merges
    const bool can = has && room8;
    if (!wave_any(can) || wave_any(has && !room8))
    {
      if (wave_all(all_done && !has))
        break; // all rows consumed and every queue drained ("done" comes in one word with the final count)
      // Nothing to do until a helper has moved -- and the helpers need the SIMD's issue slots to move: a waiting wave that
      // kept its priority would poll them out of the way (two coding waves on a SIMD, both waiting at priority 3, starved
      // their helpers of every slot: a batch of 128 Ki channels took seconds)
      if (!waiting)
      {
        wave_priority<0>();
        waiting = true;
      }
      wave_sleep<DG_ENC_CODE_SLEEP>();
      DG_COUNT(3);
      continue;
    }
    if (waiting)
    {
      wave_priority<DG_ENC_CODE_PRIO>();
      waiting = false;
    }
    // Which word path?  The wave takes the most expensive class among its lanes.
    uint32_t cls = can ? enc.cls : CLS_FAST8;
    bool act = can; // lanes that code (a part of) a word in this step
    if (!wave_any(cls != CLS_FAST8))
    {
      if (can)
        enc.template encode_word<false, 8>(word, tab, Mg);
    }
    else
    {
      // A lane in the middle of a halving word (CLS_SPLIT, second part) needs the masked fast path: lanes that need a
      // slower one sit this step out.  A halving word not yet begun goes whole through the general path when the
      // wave takes that one anyway.
      if (wave_any(can && enc.part_lo != 0u))
        act = can && cls <= CLS_SPLIT;
      else if (wave_any(cls > CLS_SPLIT))
      {
        if (cls == CLS_SPLIT)
        {
          cls = CLS_GENERAL;
          enc.whole_word();
        }
      }
      if (!wave_any(act && cls > CLS_SPLIT))
      {
        if constexpr (ADAPTIVE)
        {
          if (act)
            enc.template encode_word<false, 8, true>(word, tab, Mg);
        }
      }
      else
      {
        const bool any_bits = wave_any(cls == CLS_BITS), any_general = wave_any(cls == CLS_GENERAL);
        if (can)
        {
          if (any_bits)
          {
#pragma unroll 1
            for (uint32_t i = 0; i < 32; i++)
              enc.encode_bit((word >> (31u - i)) & 1u, tab);
            enc.end_bits_word();
          }
          else if (any_general)
          {
            if constexpr (ADAPTIVE)
              enc.template encode_word<true, 4>(word, tab, Mg);
          }
          else
            enc.template encode_word<false, 4>(word, tab, Mg);
        }
      }
    }
    if (act)
    {
      const bool whole = enc.cls != CLS_SPLIT || cls != CLS_SPLIT || enc.after_part(word);
      rd += whole ? 1u : 0u;
      enc.safe -= enc.safe != 0u ? 1u : 0u;
      if (whole && enc.safe == 0u)
        enc.classify(); // the class of the next word
    }
    peer_store(pub_mine, (rd & 0xFFFFu) | ((enc.rwr & 0xFFu) << 16));
    DG_COUNT(2);
  }

  if (live)
  {
    // the last, partial word of the seg stream (left aligned in the helper's next ring slot), then EOF + flush
    // (bac.c:163-164) -- or, when more rows follow in a later launch, nothing but a last dump: the state keeps A only
    const uint32_t tail = peer >> 27;
    const uint32_t tword = ring_col[(rd % RING) * 64u];
    for (uint32_t i = 0; i < tail; i++)
      enc.encode_bit((tword >> (31u - i)) & 1u, tab);
    if (more)
    {
      enc.dump_when_room();
      state[11 * a.C] = (uint32_t)enc.L;
      state[12 * a.C] = enc.B;
      state[13 * a.C] = enc.c1;
      state[14 * a.C] = enc.tot;
      state[15 * a.C] = enc.mps;
    }
    else
      enc.finish(tab);
  }
  peer_store(pub_mine, (rd & 0xFFFFu) | ((enc.rwr & 0xFFu) << 16) | ENC_PUB_DONE | (peer & ENC_PUB_BAD));
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
  if (live && lane < 8) // (the writer of a diagnostic build leaves out_bits alone)
    a.out_bits[c] = lane < 4 ? dg_n[lane] : dg_c[lane - 4];
#endif
}

// One workgroup = GROUPS groups of three waves (filler, coder, writer) = GROUPS * 64 channels sharing the table of division
// magics: 4 groups = one coding wave per SIMD.  Batches of more than 64 Ki channels simply have more workgroups than the
// chip holds at once (a workgroup takes a CU's LDS): measured, that beats packing two coding waves onto a SIMD with
// smaller rings and one helper each (119 against 81 Gsamples/s on 128 Ki channels x 8 640).
// TABW: division magics kept in LDS.  Channels of at most ENC_SHORT_T samples never count beyond 8 192 (cum[0] starts at 3
// and grows by one per symbol, at most 65 symbols per sample): half the table and the small rings <.., 4, 16, 8, 16, ..>
// leave room for TWO workgroups per CU -- two coding waves per SIMD, which cover for each other and get more instructions
// per cycle out of it (BASELINE config 3: 1 Mi channels x 96 samples).
constexpr uint32_t ENC_SHORT_TABLE = 8192;
constexpr size_t ENC_SHORT_T = (ENC_SHORT_TABLE - 70) / 65; // 124
template <bool ADAPTIVE, bool NARROW = false, uint32_t ROWS = ENC_ROWS, uint32_t RING = ENC_RING, uint32_t RAW = ENC_RAW, uint32_t ORING = ENC_ORING, bool W64 = false,
          bool F32IN = false, uint32_t GROUPS = ENC_PAIRS, uint32_t TABW = DIV_TABLE_SIZE>
__global__ void __launch_bounds__(GROUPS * 192) dega_encode_kernel(const EncodeArgs a)
{
  constexpr uint32_t LDS_ROWS = (W64 && !F32IN) ? 2 * ROWS : ROWS;
  // One LDS array for everything (with the LDS-DMA destination in an object of its own hipcc guards every other LDS
  // access with a vmcnt(0) wait):  division magics (64 KiB) | per group: seg-bit ring, raw ring, staging ring, input rows,
  // the three published rows
  constexpr uint32_t TAB_WORDS = ADAPTIVE ? TABW : 4;
  constexpr uint32_t PER_GROUP = (RING + RAW + ORING + LDS_ROWS + 3) * 64;
  __shared__ __attribute__((aligned(16))) uint32_t lds[TAB_WORDS + GROUPS * PER_GROUP];
  static_assert(sizeof(lds) <= (TABW < DIV_TABLE_SIZE ? 80 : 160) * 1024, "LDS budget of a CU (half of it for the short-channel shape)");
  uint32_t *const tab = lds;

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = wave_uniform(threadIdx.x >> 6);
  const uint32_t group = wave % GROUPS;
  // 0: fills, 1: codes, 2: writes -- waves g, g + GROUPS, g + 2 GROUPS of the workgroup: the CU deals a workgroup's waves out
  // to its four SIMDs in turn, so the three share a SIMD.  (Which of them is the oldest there decides who issues first
  // when priorities tie; the priorities are what matters: coder 3 > writer 1 > filler 0.)
  const uint32_t role = wave / GROUPS;
  uint32_t *const group_lds = lds + TAB_WORDS + group * PER_GROUP;
  uint32_t *const ring_col = group_lds + lane;                              // seg bits waiting to be coded
  uint32_t *const raw_col = group_lds + RING * 64 + lane;                   // the coder's dumps waiting to be absorbed
  uint32_t *const oring_col = group_lds + (RING + RAW) * 64 + lane;         // coded words waiting to be stored
  uint32_t *const rows_wave = group_lds + (RING + RAW + ORING) * 64;        // the next input rows (wave uniform)
  uint32_t *const pub_filler = group_lds + (RING + RAW + ORING + LDS_ROWS) * 64 + lane;
  uint32_t *const pub_coder = pub_filler + 64;
  uint32_t *const pub_writer = pub_filler + 128;
  if (role == 0) // nothing written, nothing consumed, nothing absorbed
  {
    *pub_filler = 0;
    *pub_coder = 0;
    *pub_writer = 0;
  }
  // (a launch that goes on from saved state may be anywhere in its channels' streams: the whole table)
  load_div_table<ADAPTIVE>(tab, a.div_magic, a.seg_state != nullptr ? ~0ull : (uint64_t)a.T * (W64 ? 127u : 65u) + 2u); // ends with the workgroup's only barrier

  const size_t c_wave0 = (size_t)blockIdx.x * (GROUPS * 64u) + group * 64u;
  const size_t c = c_wave0 + lane;
  const bool live = c < a.C;
  if (!wave_any(live))
    return; // a group past the last channel
  if (role == 1)
    encode_coding_wave<ADAPTIVE, RING, RAW>(a, tab, ring_col, raw_col, pub_coder, pub_filler, pub_writer, lane, c, live);
  else if (role == 0)
    encode_helping_wave<NARROW, ROWS, RING, RAW, ORING, W64, F32IN, true, false>(a, ring_col, raw_col, oring_col, rows_wave, pub_filler, pub_coder, lane, c, live, c_wave0);
  else
    encode_helping_wave<NARROW, ROWS, RING, RAW, ORING, W64, F32IN, false, true>(a, ring_col, raw_col, oring_col, rows_wave, pub_writer, pub_coder, lane, c, live, c_wave0);
}

// =====================================================================================================================
// Decode (bac -> seg -> prefix sum fused), by PAIRS of waves.
//
// A channel's decoder is serial and instruction bound, and a batch of 64 Ki channels is only one wave per SIMD: whatever
// that wave waits for (LDS, the row stores, its own ballots and branches) leaves the SIMD idle.  So the work of 64
// channels is split between two waves that the CU places on the same SIMD (wave w and wave w + 4 of the workgroup):
//   the CODING wave   owns the compressed stream: LDS-DMA of the slab words into the lane's ring, the arithmetic
//                     decoder -- 32 symbols per step, word lockstep, as in the encoder -- and hands every decoded 32-bit
//                     word of seg bits to its partner through a small LDS ring;
//   the PARSING wave  takes the words, cuts them into exp-Golomb codewords, adds the differences up (diff.c:32-35), parks
//                     the samples in the lane's column of an LDS sample ring and stores every row that all 64 lanes have
//                     produced as one coalesced 256-byte segment.
// The two run concurrently, each filling the issue slots the other leaves empty.  They talk through one published
// word per lane and direction (counters modulo 2^16, see peer_store / peer_load); a wave with nothing to do sleeps.
// =====================================================================================================================
constexpr uint32_t DEC_IRING = 16;                 // staged stream words per lane
constexpr uint32_t DEC_BRING = 8;                  // decoded words in flight between the two waves, per lane
constexpr uint32_t DEC_SRING = 16;                 // decoded samples per lane (the 64-bit variants)
constexpr uint32_t DEC_PAIRS = 4;                  // pairs of waves per workgroup
constexpr uint32_t DEC_BLOCK = DEC_PAIRS * 192;    // threads per workgroup: a coding, a parsing and a loading wave per 64 channels
constexpr uint32_t DEC_CHANNELS = DEC_PAIRS * 64;  // channels per workgroup

// what the coding wave publishes: words handed over so far (mod 2^16) | valid bits of the LAST word if it is a partial
// one << 16 | no more words will come << 24 | the stream was found invalid << 25
constexpr uint32_t DEC_PUB_DONE = 1u << 24, DEC_PUB_BAD = 1u << 25;
// what the parsing wave publishes: words taken so far (mod 2^16) | the lane needs no more words << 16
constexpr uint32_t DEC_PUB_FINAL = 1u << 16;

struct DecodeArgs
{
  const uint8_t *in; // [C][cap]
  size_t cap;
  const uint64_t *in_bits;
  size_t C, T, ld;
  int32_t *x; // [T][ld]
  int32_t *err;
  const uint32_t *div_magic;
  uint64_t *out_count; // NULL: every channel must hold exactly T samples.  Else: up to T samples, count reported here
  uint32_t valuesize;  // 1..32: samples come out as the low valuesize bits, zero extended (diff.c:34)
  uint32_t big_endian; // 32-bit samples are stored byte swapped (what `decode diff` writes: big-endian words)
  float factor;        // F32OUT variants: Denormalize (normalize.c:29-41) fused into the row write; x is float [T][ld]
  // NULL, or one word per wave of 64 channels in HOST memory: the rows the wave has stored, reported whenever it passes a
  // multiple of band_rows (all ones when it is done) -- the host pipeline downloads a batch of few, long channels in bands
  // of rows while the kernel is still decoding (dega_pipeline.hpp)
  uint32_t *rows_done;
  uint32_t band_rows;
};

struct alignas(16) DecodeQuad
{
  uint32_t w[4];
};

// ---- the coding wave -------------------------------------------------------------------------------------------------
// pair_lds: [DEC_IRING rows: the lane's stream words][4 rows: DMA landing area][DEC_BRING rows: decoded words]
//           [1 row: published by this wave][1 row: published by the partner] ...
// SPLIT: the stream words are staged by a loading wave of their own (decode_loading_wave; `pub_loader` = its published
// word: words staged so far, mod 2^16) and this wave has its steady paths; else it stages them itself, as the pairs of
// the wide workgroups do.
template <bool ADAPTIVE, bool SPLIT>
DG_DEV void decode_coding_wave(const DecodeArgs &a, const uint32_t *tab, uint32_t *pair_lds, const uint32_t *pub_loader, uint32_t lane, size_t c, bool live)
{
  uint32_t *const iring = pair_lds + lane;
  uint32_t *const stage_wave = pair_lds + DEC_IRING * 64;
  uint32_t *const bring = pair_lds + (DEC_IRING + 4) * 64 + lane;
  uint32_t *const pub_mine = pair_lds + (DEC_IRING + 4 + DEC_BRING) * 64 + lane;
  const uint32_t *const pub_peer = pub_mine + 64;

  const uint32_t cap_words = (uint32_t)(a.cap / 4);
  const uint32_t *const src = reinterpret_cast<const uint32_t *>(a.in + (live ? c : 0) * a.cap);
  const uint64_t nbits = live ? a.in_bits[c] : 0;
  StreamTail tail;
  tail.init(nbits, cap_words);
  const uint32_t max_seg_bits = (uint32_t)a.T * 65u; // no valid stream of T samples decodes to more bits (T <= 2^25)
  // 16 bytes per lane and DMA instruction when the slabs allow it (a lane's words are 16-byte aligned, and a group of
  // four never leaves the slab); else four single words
  const bool quads = (a.cap % 16u) == 0u && (((size_t)a.in) % 16u) == 0u;

  StreamWindow<DEC_IRING> in;
  in.ring_col = iring;
  BacDecoder<ADAPTIVE> dec;
  dec.init();

  uint32_t in_loaded = 0;   // stream words staged so far (a multiple of 4; beyond the stream: zeros) -- SPLIT: as far as this wave has seen, see `ahead`
  bool requested = false;   // a group of 4 is on its way (or due as zeros)
  bool slow_word = false;   // (SPLIT) the steady path gave this lane's word up: it goes bit by bit
  bool started = false;     // StartDecoding done
  bool bac_done = !live;    // EOF symbol seen, error, or the partner wants no more
  uint32_t seg_bits = 0;    // seg bits decoded so far
  uint32_t wr = 0;          // words handed over
  uint32_t acc = 0, nacc = 0; // a word in the making, bit by bit (see below)
  uint32_t pub_flags = 0;
  if (!wave_any(live))
    return; // a wave past the last channel
  wave_priority<DG_DEC_CODE_PRIO>();

  // SPLIT: the loader's count is known modulo 2^16; it never trails the words this wave has passed, and leads them by at
  // most the ring: the difference is exact.  `in_loaded` is rebuilt from it at the top of every step.
  auto loaded_from = [&](uint32_t loader_word) { return (uint32_t)(dec.bp >> 5) + ((loader_word - (uint32_t)(dec.bp >> 5)) & 0xFFFFu); };
  [[maybe_unused]] auto request_refill = [&]() // 4 more words for every lane that has ring room for them
  {
    const uint32_t k0 = (uint32_t)(dec.bp >> 5);
    const bool want = live && !requested && !bac_done && in_loaded + 4u - k0 <= DEC_IRING;
    if (want && in_loaded < tail.words)
    {
      if (quads)
        dma_x4_to_lds(reinterpret_cast<const int32_t *>(src + in_loaded), stage_wave, lane);
      else
      {
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
          if (in_loaded + j < cap_words)
            dma_row_to_lds(reinterpret_cast<const int32_t *>(src + in_loaded + j), stage_wave + j * 64u, lane);
      }
    }
    requested = requested || want;
  };
  if constexpr (!SPLIT)
    request_refill();
  uint32_t loader_seen = SPLIT ? peer_load(pub_loader) : 0u;
  auto publish = [&]() {
    // (SPLIT: with the word this wave has reached, mod 64, for the loader: it stages at most a ring's worth ahead)
    peer_store(pub_mine, (wr & 0xFFFFu) | pub_flags | (bac_done ? DEC_PUB_DONE : 0u) | (SPLIT ? ((uint32_t)(dec.bp >> 5) & 63u) << 26 : 0u));
  };
  dec.classify();
#if defined(DEGA_DIAG) && (DEGA_DIAG & 128) && !defined(DEGA_SIM)
  // diagnostic build: what the coding wave's passes were (steady / steady masked / general / nothing to do), in cycles too
  uint64_t dd_n[4] = {0, 0, 0, 0}, dd_c[4] = {0, 0, 0, 0}, dd_t = __builtin_amdgcn_s_memtime();
#define DD_COUNT(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); dd_n[k]++; dd_c[k] += now_ - dd_t; dd_t = now_; } while (0)
#else
#define DD_COUNT(k)
#endif

  for (;;)
  {
    // ---- what the step reads from LDS first: four stream words, the first quarter of the division magics, and how far
    //      the partner(s) have come; nothing LDS is carried around the loop
    const uint32_t k0 = (uint32_t)(dec.bp >> 5);
    uint32_t pre[4], Mnext[32];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++)
      pre[j] = in.word(k0 + j);
    dec.fetch_magics_first(tab, Mnext);
    const uint32_t peer = peer_load(pub_peer);
    if constexpr (SPLIT)
    {
      in_loaded = loaded_from(loader_seen); // (the count as read one step ago: the words it covers may be read now)
      loader_seen = peer_load(pub_loader);
    }
    const bool input_ok = in_loaded >= k0 + 4u;
    if (live && !started && input_ok)
    {
      dec.start(in);
      started = true;
    }
    if ((peer & DEC_PUB_FINAL) != 0u)
      bac_done = true; // the parser has all it wants from this channel (or has given it up)
    const bool room = ((wr - peer) & 0xFFFFu) < DEC_BRING;
    const bool can = live && started && !bac_done && room;
    const bool can_word = can && input_ok && nacc == 0u;
    if constexpr (SPLIT)
    {
      // The steady state: every lane of the wave can decode a whole word, and its class (BacDecoder::classify, looked at a
      // step ahead) is a fast one -- one ballot, then straight-line code for all 64 lanes, no exec masks, no merges.  A
      // word that does not come through (the EOF symbol is in it, or its symbols take more stream bits than the word path
      // looks ahead) is given up: back to where the step began, and bit by bit in the general path below.
      const bool ready = can_word && !slow_word;
      const bool plain = wave_all(ready && dec.cls == DCLS_FAST);
      if (plain || (ADAPTIVE && wave_all(ready && dec.cls <= DCLS_SPLIT)))
      {
        const BacDecoder<ADAPTIVE> checkpoint = dec;
        uint32_t bits = 0;
        bool done, whole = true;
        if (plain)
          done = dec.template decode_word<false>(tab, Mnext, pre, bits);
        else
        {
          dec.begin_word();
          done = dec.template decode_word<false, ADAPTIVE>(tab, Mnext, pre, bits);
          if (done)
            whole = dec.after_part(bits);
        }
        if (wave_any(!done))
        {
          if (!done)
          {
            dec = checkpoint;
            nacc = dec.part_lo; // what a first part has decoded is kept as the beginning of the bit-by-bit word
            acc = nacc != 0u ? dec.part_bits >> (32u - nacc) : 0u;
            dec.whole_word();
            slow_word = true;
            whole = false;
          }
        }
        bring[(wr % DEC_BRING) * 64u] = bits; // (harmless when the word is not whole: the slot is rewritten)
        wr += whole ? 1u : 0u;
        seg_bits += whole ? 32u : 0u;
        dec.safe -= dec.safe != 0u ? 1u : 0u;
        const bool again = done && whole && dec.safe == 0u;
        if (wave_any(again))
        {
          if (again)
            dec.classify();
        }
        publish();
        DD_COUNT(plain ? 0 : 1);
        continue;
      }
    }
    if (wave_any(can))
    {
      bool done = false;
      if (wave_any(can_word))
      {
        // Which word path?  The plain fast one when no lane's counts can do anything but count in this word; the masked
        // fast one when what some lanes have is a halving (they decode the word in two parts, BacDecoder::split_ok);
        // else the general one -- unless a lane is in the middle of such a word: it goes first, the others wait a step.
        bool act = can_word;
        const bool plain = wave_all(!can_word || dec.fast_ok());
        bool masked = false;
        if constexpr (ADAPTIVE)
        {
          if (!plain)
          {
            masked = wave_all(!can_word || dec.fast_ok() || dec.split_ok());
            if (!masked && wave_any(can_word && dec.part_lo != 0u))
            {
              masked = true;
              act = can_word && (dec.fast_ok() || dec.split_ok());
            }
          }
        }
        if (act)
        {
          const BacDecoder<ADAPTIVE> checkpoint = dec;
          uint32_t bits = 0;
          bool whole = true;
          if (plain)
            done = dec.template decode_word<false>(tab, Mnext, pre, bits);
          else if constexpr (ADAPTIVE)
          {
            if (masked)
            {
              dec.begin_word();
              done = dec.template decode_word<false, true>(tab, Mnext, pre, bits);
              if (done)
                whole = dec.after_part(bits);
            }
            else
              done = dec.template decode_word<true>(tab, Mnext, pre, bits); // MPS/LPS swap, shift change, halving with little room
          }
          if (done)
          {
            if (whole)
            {
              bring[(wr % DEC_BRING) * 64u] = bits;
              wr++;
              seg_bits += 32;
            }
          }
          else
          {
            // back to where the step began; what a first part has decoded is kept as the beginning of the bit-by-bit word
            dec = checkpoint;
            nacc = dec.part_lo;
            acc = nacc != 0u ? dec.part_bits >> (32u - nacc) : 0u;
            dec.whole_word();
          }
        }
        done = done || (can_word && !act); // a lane that waits does not take the bit-by-bit path either
      }
      // Bit by bit -- the EOF symbol is in this word, or its symbols take more stream bits than the word path looks
      // ahead.  A word made this way may take several steps: a symbol is decoded only while the two stream words it may
      // touch are staged (32 rare symbols in a row can take 300 bits and more).
      const bool slow = can && !done && (nacc != 0u || can_word);
      if (wave_any(slow))
      {
        if (slow)
        {
#pragma unroll 1
          while (nacc < 32u && !bac_done && in_loaded >= (uint32_t)(dec.bp >> 5) + 2u)
          {
            const uint32_t r = dec.decode_bit(in, tab);
            if (r == 2)
            {
              bac_done = true;
              if (dec.bp > nbits + 14u)
                pub_flags |= DEC_PUB_BAD; // more than 14 phantom bits (bac.c:171-186)
            }
            else
            {
              acc = (acc << 1) | r;
              nacc++;
              seg_bits++;
            }
          }
          if (!bac_done && seg_bits > max_seg_bits)
          {
            bac_done = true; // runaway stream: cannot be T samples
            pub_flags |= DEC_PUB_BAD;
          }
          if (nacc == 32u || (bac_done && nacc != 0u))
          {
            bring[(wr % DEC_BRING) * 64u] = acc << ((32u - nacc) & 31u);
            pub_flags = (pub_flags & ~(31u << 16)) | ((nacc & 31u) << 16);
            wr++;
            acc = 0;
            nacc = 0;
          }
        }
      }
      if constexpr (SPLIT)
      {
        if (can) // the classes of the steady paths: whatever happened here, look again
        {
          slow_word = false;
          dec.classify();
        }
      }
      publish();
      DD_COUNT(2);
    }
    else
    {
      publish();
      if (wave_all(bac_done))
        break;
      wave_sleep<DG_DEC_CODE_SLEEP>(); // waiting for the partner (ring full) or for stream words
      DD_COUNT(3);
    }
    if constexpr (SPLIT)
      continue;
    // ---- the DMA issued at the end of the previous step has landed: cook the words into the lane's own ring slots ----
    if (wave_any(requested))
    {
      wait_vector_memory();
      if (requested)
      {
        uint32_t s[4];
        if (quads)
        {
          const DecodeQuad q = *reinterpret_cast<const DecodeQuad *>(stage_wave + lane * 4u);
#pragma unroll
          for (uint32_t j = 0; j < 4; j++)
            s[j] = q.w[j];
        }
        else
        {
#pragma unroll
          for (uint32_t j = 0; j < 4; j++)
            s[j] = stage_wave[j * 64u + lane];
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
          iring[((in_loaded + j) % DEC_IRING) * 64u] = tail.cook(s[j], in_loaded + j);
        in_loaded += 4;
        requested = false;
      }
    }
    // ---- refill: when some lane is down to two steps' worth, every lane with room asks for its next four words ----------
    {
      const uint32_t k1 = (uint32_t)(dec.bp >> 5);
      const bool low = live && !bac_done && in_loaded < k1 + 8u;
      if (wave_any(low))
        request_refill();
    }
  }
  if constexpr (!SPLIT)
    wait_vector_memory(); // no DMA may still be writing to LDS when the workgroup's allocation is released
#if defined(DEGA_DIAG) && (DEGA_DIAG & 128) && !defined(DEGA_SIM)
  if (live && lane < 8) // (the stream lengths have been read long ago)
    const_cast<uint64_t *>(a.in_bits)[c] = lane < 4 ? dd_n[lane] : dd_c[lane - 4];
#endif
}

// ---- the loading wave (workgroups of three waves per 64 channels) ---------------------------------------------------------
// Owns the compressed stream: four words per lane come by LDS-DMA (16 bytes per lane and instruction when the slabs are
// 16-byte aligned) into a landing area and are *cooked* into the lane's ring when they have landed -- host byte order, the
// bits beyond the exact length cleared, zeros after the last word -- whenever the coder has left room for them; it knows
// the word the coder has reached modulo 64 (bits 26..31 of the coder's published word).
DG_DEV void decode_loading_wave(const DecodeArgs &a, uint32_t *pair_lds, uint32_t *pub_mine, uint32_t lane, size_t c, bool live)
{
  uint32_t *const iring = pair_lds + lane;
  uint32_t *const stage_wave = pair_lds + DEC_IRING * 64;
  const uint32_t *const pub_coder = pair_lds + (DEC_IRING + 4 + DEC_BRING) * 64 + lane;
  const uint32_t cap_words = (uint32_t)(a.cap / 4);
  const uint32_t *const src = reinterpret_cast<const uint32_t *>(a.in + (live ? c : 0) * a.cap);
  StreamTail tail;
  tail.init(live ? a.in_bits[c] : 0, cap_words);
  const bool quads = (a.cap % 16u) == 0u && (((size_t)a.in) % 16u) == 0u;
  uint32_t in_loaded = 0; // words staged so far: a multiple of 4
  if (!wave_any(live))
    return;
  wave_priority<DG_DEC_LOAD_PRIO>();
  for (;;)
  {
    const uint32_t cp = peer_load(pub_coder);
    const bool coder_done = !live || (cp & DEC_PUB_DONE) != 0u;
    const uint32_t staged_ahead = (in_loaded - (cp >> 26)) & 63u; // words staged that the coder has not passed
    const bool want = !coder_done && staged_ahead + 4u <= DEC_IRING;
    if (wave_any(want))
    {
      if (want && in_loaded < tail.words)
      {
        if (quads)
          dma_x4_to_lds(reinterpret_cast<const int32_t *>(src + in_loaded), stage_wave, lane);
        else
        {
#pragma unroll
          for (uint32_t j = 0; j < 4; j++)
            if (in_loaded + j < cap_words)
              dma_row_to_lds(reinterpret_cast<const int32_t *>(src + in_loaded + j), stage_wave + j * 64u, lane);
        }
      }
      wait_vector_memory();
      if (want)
      {
        uint32_t w4[4];
        if (quads)
        {
          const DecodeQuad q = *reinterpret_cast<const DecodeQuad *>(stage_wave + lane * 4u);
#pragma unroll
          for (uint32_t j = 0; j < 4; j++)
            w4[j] = q.w[j];
        }
        else
        {
#pragma unroll
          for (uint32_t j = 0; j < 4; j++)
            w4[j] = stage_wave[j * 64u + lane];
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++)
          iring[((in_loaded + j) % DEC_IRING) * 64u] = tail.cook(w4[j], in_loaded + j);
        in_loaded += 4;
      }
      peer_store(pub_mine, in_loaded & 0xFFFFu);
    }
    else
    {
      if (wave_all(coder_done))
        break;
      wave_sleep<DG_DEC_LOAD_SLEEP>();
    }
  }
  wait_vector_memory(); // no DMA may still be writing to LDS when the workgroup's allocation is released
}

// ---- the parsing wave ------------------------------------------------------------------------------------------------
template <bool NARROW, bool W64, bool F32OUT, uint32_t SRING>
DG_DEV void decode_parsing_wave(const DecodeArgs &a, uint32_t *pair_lds, uint32_t lane, size_t c, bool live, size_t c_wave0)
{
  const uint32_t *const bring = pair_lds + (DEC_IRING + 4) * 64 + lane;
  const uint32_t *const pub_peer = pair_lds + (DEC_IRING + 4 + DEC_BRING) * 64 + lane;
  uint32_t *const pub_mine = pair_lds + (DEC_IRING + 4 + DEC_BRING + 1) * 64 + lane;
  uint32_t *const sring = pair_lds + (DEC_IRING + 4 + DEC_BRING + 2) * 64 + lane; // decoded samples (+ a spare slot)
  uint32_t *const sring_hi = sring + (SRING + 1) * 64;                            // their high dwords (W64 only)

  typename std::conditional<W64, SegParser64, SegParser>::type sp;
  sp.init(NARROW || W64 ? a.valuesize : 32u);
  uint32_t rd = 0;          // words taken from the partner
  bool final_in = !live;    // the partner is done and every word of it has been taken
  bool lane_final = !live;  // nothing more will come out of this lane
  uint32_t t_lane = 0;      // samples produced (T <= 2^25)
  uint32_t rows_stored = 0; // wave uniform
  const uint32_t T32 = (uint32_t)a.T;
  int32_t lane_err = OK;
  bool carry_over = true;   // the previous pass changed something a further pass could build on
  bool force_general = false; // (wave uniform) a lane's window holds something the steady pass cannot take
  const bool full_wave = c_wave0 + 64u <= a.C;
  if (!wave_any(live))
    return; // a wave past the last channel
  uint32_t peer = peer_load(pub_peer);
  wave_priority<DG_DEC_PARSE_PRIO>();
#if defined(DEGA_DIAG) && (DEGA_DIAG & 128) && !defined(DEGA_SIM)
  // diagnostic build: the parsing wave's passes (steady / general / cheap polls / sleeps in the general pass), in cycles too
  uint32_t dp_n[4] = {0, 0, 0, 0};
  uint64_t dp_c[4] = {0, 0, 0, 0}, dp_t = __builtin_amdgcn_s_memtime();
#define DP_COUNT(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); dp_n[k]++; dp_c[k] += now_ - dp_t; dp_t = now_; } while (0)
#else
#define DP_COUNT(k)
#endif
  uint32_t *const report = a.rows_done != nullptr ? a.rows_done + c_wave0 / 64u : nullptr; // (wave uniform)
  uint32_t next_report = a.band_rows;

  // one decoded value -> memory, in the form the variant writes (rows past a failed channel's last sample: zeros).
  // `row` is the same in every lane: the row's address is a scalar, the lane adds its column.
  const bool swap_bytes = a.big_endian != 0;
  auto store_value = [&](uint32_t row, uint32_t lo, uint32_t hi, bool valid) {
    const size_t at = (size_t)row * a.ld + c_wave0; // (wave uniform)
    if constexpr (F32OUT)
    {
      float v;
      if constexpr (W64)
      {
        const uint32_t sh64 = 64u - a.valuesize; // 0..31
        v = denormalize_value((float)((int64_t)((((uint64_t)hi << 32) | lo) << sh64) >> sh64), a.factor);
      }
      else
        v = denormalize_value((float)(NARROW ? (int32_t)(lo << sp.vshift) >> sp.vshift : (int32_t)lo), a.factor);
      (reinterpret_cast<float *>(a.x) + at)[lane] = valid ? v : 0.0f;
    }
    else if constexpr (W64)
      (reinterpret_cast<int64_t *>(a.x) + at)[lane] = valid ? (int64_t)(((uint64_t)hi << 32) | lo) : 0;
    else
      (a.x + at)[lane] = valid ? (int32_t)(swap_bytes ? bswap32(lo) : lo) : 0;
  };

  for (;;)
  {
    // ---- the next decoded word, if the partner has one and the window has room for it ---------------------------------
    // (the partner's count as read one pass ago: a word may only be read after a count that covers it, and this way the
    // two reads go out together; the count only grows)
    const uint32_t avail = (peer - rd) & 0xFFFFu;
    const bool peer_done = (peer & DEC_PUB_DONE) != 0u;
    const bool got = !lane_final && avail != 0u && sp.has_room();
    const uint32_t peer_seen = peer;
    if constexpr (!W64)
    {
      // The steady state: all 64 channels exist, no lane is at its stream's end, inside a long codeword or in error, and
      // the channels' ends are more than a sample ring away -- two ballots, then straight-line code for all 64 lanes: the
      // word in (for the lanes that have one and room for it), four codewords off the top (two more while some lane's
      // window could not take its next word), and every eight rows that all lanes have, out.  A lane whose window holds
      // 32 bits and no short codeword (a long one: first samples, jumps) sends the next pass through the general code.
      if (full_wave && !force_general && rows_stored + SRING <= T32 && wave_all(!lane_final && !peer_done && !sp.pending() && lane_err == OK) && wave_any(got))
      {
        const uint32_t word = bring[(rd % DEC_BRING) * 64u];
        peer = peer_load(pub_peer);
        sp.push_word(got ? word : 0u, got ? 32u : 0u);
        rd += got ? 1u : 0u;
        peer_store(pub_mine, rd & 0xFFFFu);
        const uint32_t t_limit = rows_stored + SRING;
        bool more = false;
        auto take = [&]() {
          uint32_t sample;
          const uint32_t okm = sp.template take_short_lean<NARROW>(t_lane < t_limit, sample);
          sring[select32(okm, t_lane % SRING, SRING) * 64u] = sample; // (a lane that took nothing writes to the spare slot)
          t_lane -= okm;
          return okm != 0u;
        };
#pragma unroll
        for (uint32_t k = 0; k < DG_DEC_TAKES; k++)
          more = take();
        while (wave_any(more && !sp.has_room()))
        {
          more = take();
          more = take();
        }
        force_general = wave_any(!more && t_lane < t_limit && sp.cnt >= 32u);
        while (wave_all(t_lane >= rows_stored + 8u))
        {
          const uint32_t r0 = wave_uniform(rows_stored);
          uint32_t cand[8];
#pragma unroll
          for (uint32_t k = 0; k < 8; k++)
            cand[k] = sring[((r0 + k) % SRING) * 64u];
#pragma unroll
          for (uint32_t k = 0; k < 8; k++)
            store_value(r0 + k, cand[k], 0u, true);
          rows_stored = r0 + 8u;
        }
        if (report != nullptr && rows_stored >= next_report)
        {
          if (lane == 0)
            store_read_by_host(report, rows_stored);
          next_report = (rows_stored / a.band_rows + 1u) * a.band_rows;
        }
        carry_over = force_general; // (else what is left in the windows waits for the next word: nothing to gain from a pass without one)
        DP_COUNT(0);
        continue;
      }
      force_general = false;
      // nothing new and nothing pending: a cheap poll (the general pass below costs a few hundred instructions)
      if (!carry_over && !wave_any((avail != 0u && !lane_final) || (live && peer_done && !lane_final)))
      {
        peer = peer_load(pub_peer);
        wave_sleep<DG_DEC_PARSE_SLEEP>();
        DP_COUNT(2);
        continue;
      }
    }
    {
      const uint32_t word = bring[(rd % DEC_BRING) * 64u];
      peer = peer_load(pub_peer);
      const uint32_t part = (peer_seen >> 16) & 31u;
      const uint32_t n = (peer_done && avail == 1u && part != 0u) ? part : 32u; // only the last word can be a partial one
      sp.push_word(got ? word : 0u, got ? n : 0u);
      rd += got ? 1u : 0u;
    }
    final_in = !live || (peer_done && ((peer_seen - rd) & 0xFFFFu) == 0u);
    peer_store(pub_mine, (rd & 0xFFFFu) | (lane_final ? DEC_PUB_FINAL : 0u));
    if (!wave_any(got || (final_in && !lane_final)) && !carry_over)
    {
      wave_sleep<DG_DEC_PARSE_SLEEP>();
      DP_COUNT(3);
      continue;
    }
    if (final_in && (peer_seen & DEC_PUB_BAD) != 0u && lane_err == OK)
      lane_err = ERR_INVALID_FORMAT; // found by the arithmetic decoder: phantom bits, runaway stream
    const uint32_t t_before = t_lane, rows_before = rows_stored;
    const bool final_before = lane_final;

    // ---- parse what is there -------------------------------------------------------------------------------------------
    bool more = false;
    // (1) the steady state, branch free: short codewords off the top of the window; a lane that cannot take one writes to
    //     a spare slot of its sample column instead.  A 32-bit word holds 3-4 codewords of this data: four takes (measured: 3 .. 6 are within 1 %), then
    //     two more for as long as some lane's window would not have room for its next word
    if constexpr (!W64)
    {
      const uint32_t t_limit = rows_stored + SRING < T32 ? rows_stored + SRING : T32; // room in the sample ring, samples asked for
      auto take = [&]() {
        uint32_t sample;
        const bool allowed = !lane_final && t_lane < t_limit;
        const bool took = sp.template take_short<NARROW>(allowed, sample);
        sring[(took ? (uint32_t)(t_lane % SRING) : SRING) * 64u] = sample;
        t_lane += took ? 1u : 0u;
        return took;
      };
#pragma unroll
      for (uint32_t k = 0; k < DG_DEC_TAKES; k++)
        more = take();
      while (wave_any(more && (!sp.has_room() || final_in)))
      {
        more = take();
        more = take();
      }
    }
    // (2) everything else -- codewords of 33+ bits, the end of the stream, too many samples -- one codeword per pass;
    //     entered only by lanes that cannot simply wait for more bits
    bool stalled = lane_final || more || t_lane - rows_stored >= SRING || !(final_in || sp.cnt >= 32u || sp.pending());
    while (wave_any(!stalled))
    {
      if (!stalled)
      {
        if (t_lane - rows_stored >= SRING)
          stalled = true; // sample ring full until rows are written
        else
        {
          typename std::conditional<W64, uint64_t, uint32_t>::type sample = 0;
          int32_t r;
          if constexpr (W64)
            r = sp.next(final_in, sample);
          else
            r = sp.template next<NARROW>(final_in, sample);
          if (r == 1)
          {
            if (t_lane >= T32)
            {
              if (lane_err == OK)
                lane_err = a.out_count != nullptr ? ERR_MEMORY : ERR_INVALID_FORMAT; // more samples than room / than asked for
              lane_final = true;
              stalled = true;
            }
            else
            {
              sring[(t_lane % SRING) * 64u] = (uint32_t)sample;
              if constexpr (W64)
                sring_hi[(t_lane % SRING) * 64u] = (uint32_t)((uint64_t)sample >> 32);
              t_lane++;
            }
          }
          else if (r == 0)
            stalled = true; // needs more decoded bits
          else
          {
            if (r < 0 && lane_err == OK)
              lane_err = r;
            if (r == 2 && t_lane != T32 && a.out_count == nullptr && lane_err == OK)
              lane_err = ERR_INVALID_FORMAT; // fewer samples than the caller asked for
            lane_final = true;
            stalled = true;
          }
        }
      }
    }
    if (lane_err != OK)
      lane_final = true; // a failed channel stops here; its remaining rows are written as zeros

    // ---- rows every lane has -----------------------------------------------------------------------------------------------
    // The steady state: all 64 channels exist, none has ended, and the next eight rows are there for every lane -- one
    // ballot, eight samples out of the ring, eight 256-byte stores.
    while (full_wave && rows_stored + 8u <= T32 && wave_all(t_lane >= rows_stored + 8u))
    {
      const uint32_t r0 = wave_uniform(rows_stored);
      uint32_t cand[8], cand_hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (uint32_t k = 0; k < 8; k++)
      {
        cand[k] = sring[((r0 + k) % SRING) * 64u];
        if constexpr (W64)
          cand_hi[k] = sring_hi[((r0 + k) % SRING) * 64u];
      }
#pragma unroll
      for (uint32_t k = 0; k < 8; k++)
        store_value(r0 + k, cand[k], cand_hi[k], true);
      rows_stored = r0 + 8u;
    }
    // The ends -- a partial wave, channels that have finished or failed, the last rows, a reported count (then rows past
    // the longest channel of the wave are not written at all): up to 4 rows per pass, each checked on its own.
    if (!full_wave || wave_any(lane_final) || rows_stored + 8u > T32)
    {
      for (;;)
      {
        const bool four = rows_stored + 4u <= T32 && wave_all(lane_final || t_lane >= rows_stored + 4u);
        if (!four && !wave_all(lane_final))
          break; // lanes are still producing: the rows wait until four are whole (a lane may run SRING samples ahead)
        uint32_t cand[4], cand_hi[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
        {
          cand[k] = sring[((rows_stored + k) % SRING) * 64u];
          if constexpr (W64)
            cand_hi[k] = sring_hi[((rows_stored + k) % SRING) * 64u];
        }
        uint32_t wrote = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
        {
          const uint32_t row = wave_uniform(rows_stored) + k;
          if (wrote == k && row < T32 && (four || wave_all(lane_final || t_lane > row)) && (a.out_count == nullptr || wave_any(t_lane > row)))
          {
            if (live)
              store_value(row, cand[k], cand_hi[k], t_lane > row);
            wrote = k + 1;
          }
        }
        rows_stored += wrote;
        if (wrote < 4)
          break;
      }
    }
    if (report != nullptr && rows_stored >= next_report)
    {
      if (lane == 0)
        store_read_by_host(report, rows_stored);
      next_report = (rows_stored / a.band_rows + 1u) * a.band_rows;
    }
    if (wave_all(lane_final) && (rows_stored >= T32 || (a.out_count != nullptr && !wave_any(t_lane > rows_stored))))
      break;
    carry_over = rows_stored != rows_before || wave_any(t_lane != t_before || lane_final != final_before);
    DP_COUNT(1);
  }
  peer_store(pub_mine, (rd & 0xFFFFu) | DEC_PUB_FINAL);
  if (report != nullptr && lane == 0)
    store_read_by_host(report, 0xFFFFFFFFu);
  if (live)
  {
    a.err[c] = lane_err;
    if (a.out_count != nullptr)
      a.out_count[c] = t_lane;
#if defined(DEGA_DIAG) && (DEGA_DIAG & 128) && !defined(DEGA_SIM)
    if (lane < 8)
      a.err[c] = lane < 4 ? (int32_t)dp_n[lane] : (int32_t)(dp_c[lane - 4] >> 10); // (cycles in units of 1024)
#endif
  }
}

// One workgroup = PAIRS groups of waves = PAIRS * 64 channels: wave p codes, wave PAIRS + p parses for it and -- SPLIT --
// wave 2 PAIRS + p stages its stream words (the CU deals a workgroup's waves out to its four SIMDs in turn, so the waves
// of a group land on the same SIMD; nothing but speed depends on it).  LDS: division magics (64 KiB) | per group: stream
// ring, DMA rows, decoded-word ring, two published rows, sample ring, the loader's published row -- 160 KiB for the
// 32-bit variants.
// NARROW: valuesize < 32.  W64: valuesize 33..64 -- a.x is int64 [T][ld]; the parser is SegParser64 (no short-codeword
// passes), samples take two LDS slots.  F32OUT: the decoded value, read back as valuesize bits sign extended
// (normalize.c:36-37), leaves as (float)n / factor (:38, IEEE division) -- float32 rows [T][ld] also for W64, no
// integer intermediate in HBM.
// PAIRS / SPLIT: 4 groups of three waves (one coding wave per SIMD, its two helpers beside it); or 8 pairs, with the short
// sample ring and the coder staging its own words -- two coding waves per SIMD.
template <bool ADAPTIVE, bool NARROW = false, bool W64 = false, bool F32OUT = false, uint32_t PAIRS = DEC_PAIRS, bool SPLIT = true>
__global__ void __launch_bounds__(PAIRS * (SPLIT ? 192 : 128)) dega_decode_kernel(const DecodeArgs a)
{
  constexpr uint32_t TAB_WORDS = ADAPTIVE ? DIV_TABLE_SIZE : 4;
  // decoded samples a lane may run ahead of the slowest lane of its wave before it has to wait for the row writer
  constexpr uint32_t SRING = (W64 || PAIRS > 4) ? DEC_SRING : 64;
  constexpr uint32_t LOADER_ROW = DEC_IRING + 4 + DEC_BRING + 2 + (W64 ? 2 : 1) * (SRING + 1); // + a spare sample slot
  constexpr uint32_t PER_PAIR = (LOADER_ROW + (SPLIT ? 1 : 0)) * 64;
  __shared__ __attribute__((aligned(16))) uint32_t lds[TAB_WORDS + PAIRS * PER_PAIR];
  static_assert(sizeof(lds) <= 160 * 1024, "LDS budget of a CU");
  uint32_t *const tab = lds;

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = wave_uniform(threadIdx.x >> 6);
  const uint32_t pair = wave % PAIRS;
  const uint32_t role = wave / PAIRS; // 0: codes, 1: parses, 2: loads
  uint32_t *const pair_lds = lds + TAB_WORDS + pair * PER_PAIR;
  uint32_t *const pub_loader = pair_lds + LOADER_ROW * 64 + lane;
  if (role == 0) // nothing handed over, nothing taken, nothing staged
  {
    pair_lds[(DEC_IRING + 4 + DEC_BRING) * 64 + lane] = 0;
    pair_lds[(DEC_IRING + 4 + DEC_BRING + 1) * 64 + lane] = 0;
    if (SPLIT)
      *pub_loader = 0;
  }
  load_div_table<ADAPTIVE>(tab, a.div_magic, (uint64_t)a.T * (W64 ? 127u : 65u) + 2u); // ends with the workgroup's only barrier

  const size_t c_wave0 = (size_t)blockIdx.x * (PAIRS * 64u) + pair * 64u;
  const size_t c = c_wave0 + lane;
  const bool live = c < a.C;
  if (role == 1)
    decode_parsing_wave<NARROW, W64, F32OUT, SRING>(a, pair_lds, lane, c, live, c_wave0);
  else if (role == 0)
    decode_coding_wave<ADAPTIVE, SPLIT>(a, tab, pair_lds, pub_loader, lane, c, live);
  else
    decode_loading_wave(a, pair_lds, pub_loader, lane, c, live);
}

// ---------------------------------------------------------------------------------------------------------------------
// normalize / denormalize (DCLib/src/normalize.c), elementwise, HBM bound.  Float parity rules: the
// multiply and the +-0.5 are two separately rounded operations (no FMA), truncating convert, IEEE division.
// ---------------------------------------------------------------------------------------------------------------------
struct NormalizeArgs
{
  const float *v;
  int32_t *x;
  size_t C, T, ld;
  float factor;
  int32_t *err; // [C], pre-zeroed
  float lo, hi; // range of normalize.c:21 for the value size
  uint32_t mask;
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
    bad |= !normalize_value(a.v[t * a.ld + c], a.factor, n, a.lo, a.hi, a.mask);
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
  uint32_t vshift; // 32 - valuesize: the value is the low valuesize bits, sign extended (normalize.c:36-37)
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
    const int32_t n = (int32_t)((uint32_t)a.x[t * a.ld + c] << a.vshift) >> a.vshift;
    a.v[t * a.ld + c] = denormalize_value((float)n, a.factor);
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

// One wave copies n bytes, any alignment on either side: bytes up to the destination's next 16-byte boundary, then 16
// bytes per lane and round -- the source read as the five dwords that hold them at its own alignment, shifted into place
// (v_alignbyte), one 16-byte store -- then the last few bytes.  Reads stay inside [src rounded down to 4, src + n).
DG_DEV void wave_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t n, uint32_t lane)
{
  uint64_t head = (16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u;
  head = head < n ? head : n;
  if (lane < head)
    dst[lane] = src[lane];
  dst += head;
  src += head;
  n -= head;
  const uint32_t sh = (uint32_t)((uintptr_t)src & 3u);
  const uint32_t *const s4 = reinterpret_cast<const uint32_t *>(src - sh);
  const uint64_t rounds = n >= 20u ? (n - 4u) / 16u : 0u;
  for (uint64_t i = lane; i < rounds; i += 64)
  {
    const uint32_t *const p = s4 + 4u * i;
    const uint32_t w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
#if defined(DEGA_SIM)
    const uint64_t q0 = ((uint64_t)w1 << 32) | w0, q1 = ((uint64_t)w2 << 32) | w1, q2 = ((uint64_t)w3 << 32) | w2, q3 = ((uint64_t)w4 << 32) | w3;
    const uint32_t o0 = (uint32_t)(q0 >> (8u * sh)), o1 = (uint32_t)(q1 >> (8u * sh)), o2 = (uint32_t)(q2 >> (8u * sh)), o3 = (uint32_t)(q3 >> (8u * sh));
#else
    const uint32_t o0 = __builtin_amdgcn_alignbyte(w1, w0, sh), o1 = __builtin_amdgcn_alignbyte(w2, w1, sh), o2 = __builtin_amdgcn_alignbyte(w3, w2, sh),
                   o3 = __builtin_amdgcn_alignbyte(w4, w3, sh);
#endif
    uint32_t *const d4 = reinterpret_cast<uint32_t *>(dst + 16u * i);
    d4[0] = o0;
    d4[1] = o1;
    d4[2] = o2;
    d4[3] = o3;
  }
  const uint64_t done = 16u * rounds;
  if (done + lane < n) // fewer than 20 bytes are left
    dst[done + lane] = src[done + lane];
}

// one wave per channel (offsets are arbitrary byte positions)
__global__ void __launch_bounds__(256) dega_gather_kernel(const GatherArgs a)
{
  const size_t c = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (c >= a.C)
    return;
  const uint64_t o0 = a.offsets[c], o1 = a.offsets[c + 1];
  wave_copy_bytes(a.packed + o0, a.slabs + c * a.cap, o1 - o0, threadIdx.x & 63u);
}

// the inverse: packed streams -> slabs (the decoders mask what lies beyond a stream's bit length, so nothing is padded)
__global__ void __launch_bounds__(256) dega_scatter_kernel(const GatherArgs a)
{
  const size_t c = (size_t)blockIdx.x * WAVES + (threadIdx.x >> 6);
  if (c >= a.C)
    return;
  const uint64_t o0 = a.offsets[c], o1 = a.offsets[c + 1];
  wave_copy_bytes(const_cast<uint8_t *>(a.slabs) + c * a.cap, a.packed + o0, o1 - o0 < a.cap ? o1 - o0 : a.cap, threadIdx.x & 63u);
}

} // namespace dg
