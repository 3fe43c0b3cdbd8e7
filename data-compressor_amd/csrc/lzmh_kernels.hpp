// lzmh_kernels.hpp -- LZMH (BASELINE config 4) for gfx950: the reference's second codec, DCLib/src/lzmh.c:130-574.
// Encode kernel first (this comment), decode kernel and the ASCII rendering of the workload further down.
//
// Mapping: one lane = one channel (an independent byte string), one wave = 64 channels, one workgroup = 4 PAIRS of waves:
// of each pair one wave searches (decode: reads the code), its partner on the same SIMD codes (decode: writes) -- see the
// kernels.  The codec is serial per channel (every code depends on the match window, the recent-offset cache and the
// frequency list left by the previous one), so -- as for DEGA -- all parallelism is across channels.
//
// What the reference does per step, restated without its 403-byte ring (lzmh.c:143-146, 174-191, 343-363): with P
// bytes consumed so far, look back at most min(P, 128) bytes and ahead at most min(274, min(n, max(P-128, 0) + 403) - P)
// bytes; take the longest match, the smallest offset among equals (ascending scan with `length > bestlength`, :196-214),
// code it if it is 3 bytes or longer, else code one literal through the frequency-sorted symbol list.  The ring only
// shows in those two bounds and in one quirk, kept here: an input of exactly 403 bytes produces no output (:168-174).
//
// How a lane searches (the 128-offset scan is ~90% of the reference's time):
//   phase 1  the lane's last 132 window bytes + the next 24 are read from LDS into registers (38 dwords, one latency),
//            and a byte-parallel compare marks every position whose first three bytes equal the next three input bytes:
//            per dword 2 v_alignbyte, xor, two v_bitop3, a zero-byte test (add, v_bitop3) and a multiply that gathers the
//            four flags -- 9
//            (round 2: 11) instructions for 4 positions instead of a loop iteration with two dependent LDS reads per offset.
//   phase 2  candidates are popped nearest first (= ascending offset), six per pass of the wave's
//            loop, and measured against the input bytes held in registers -- the first 8 bytes of all six, then the
//            next 8 of the few that match those (all LDS reads of a round in flight together); `len > best` in that
//            order keeps the nearest of equals.  A lane with more candidates keeps its step and takes further passes
//            while the other lanes go on to their next steps.
// Window: 432 bytes per lane in LDS ([dword][lane], conflict free), reloaded from HBM for every lane with work when one
// runs out of look-ahead (every ~270 consumed bytes; L2 absorbs the overlap).  Frequency list (48 x {symbol, count}),
// four staged output words and the token ring per lane are in LDS as well: 158 KiB per workgroup, one workgroup per CU.
//
// Compiled by hipcc (dega_hip.hip) and, for offline debugging only, by g++ under tests/sim/.
#pragma once

#include "dega_kernels.hpp"

namespace dg
{

constexpr uint32_t LZ_HISTORY = 128;    // LZ_MAX_OFFSET, lzmh.c:57
constexpr uint32_t LZ_MAX_LENGTH = 274; // lzmh.c:61
constexpr uint32_t LZ_RING = 403;       // INTERN_BUFFER_LENGTH, lzmh.c:64
constexpr uint32_t LZ_LIST = 48;        // HUFFLIST_LENGTH, lzmh.c:67
constexpr uint32_t LZ_TREE = 19;        // HUFFTREE_LENGTH, lzmh.c:70

constexpr uint32_t LZ_BLOCK = 256;
constexpr uint32_t LZ_WIN_DW = 108;              // window dwords per lane (432 bytes)
constexpr uint32_t LZ_WIN_BYTES = 4 * LZ_WIN_DW;
// a freshly loaded window (history, up to 15 bytes of alignment) must hold the longest match, or a step could never end
static_assert(LZ_WIN_BYTES >= LZ_HISTORY + 15 + LZ_MAX_LENGTH && LZ_WIN_DW % 4 == 0, "window too small");
#ifndef DG_LZ_POP
#define DG_LZ_POP 6 // (on the probe batch: 4 and 5 cost 1.6 % more, 8 costs 8 % more)
#endif
constexpr uint32_t LZ_POP = DG_LZ_POP;           // candidates of one mask register handled per pass (tools/tunebench.py)
constexpr uint32_t LZ_AHEAD = 48;                // look-ahead a step needs in the window: 24 bytes in registers + slack
constexpr uint32_t LZ_SYM_DW = LZ_LIST / 4, LZ_CNT_DW = LZ_LIST / 2, LZ_STAGE_DW = 4;
constexpr uint32_t LZ_OFF_WIN = 0, LZ_OFF_SYM = LZ_OFF_WIN + LZ_WIN_DW * LZ_BLOCK, LZ_OFF_CNT = LZ_OFF_SYM + LZ_SYM_DW * LZ_BLOCK,
                   LZ_OFF_STAGE = LZ_OFF_CNT + LZ_CNT_DW * LZ_BLOCK, LZ_LDS_DW = LZ_OFF_STAGE + LZ_STAGE_DW * LZ_BLOCK;
static_assert(LZ_LDS_DW * 4 <= 160 * 1024, "LDS budget of one CU");

typedef uint32_t lz_u32x4 __attribute__((vector_size(16)));

struct LzmhEncodeArgs
{
  const uint8_t *in;       // [C][stride] bytes, stride a multiple of 16, base 16-byte aligned
  size_t stride;
  const uint64_t *in_len;  // [C] bytes per channel, <= stride
  size_t C;
  uint8_t *out;            // [C][cap], cap a multiple of 16
  size_t cap;
  uint64_t *out_bits;      // [C] exact bit length
  int32_t *err;            // [C]
};

DG_DEV uint32_t lz_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) // bytes s..s+3 of the 8 bytes {lo, hi}
{
#if defined(DEGA_SIM)
  return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (s & 3u)));
#else
  return __builtin_amdgcn_alignbyte(hi, lo, s);
#endif
}

DG_DEV uint32_t lz_shift_in_nibble(uint32_t acc, uint32_t m) // (acc >> 4) | (m << 28)
{
#if defined(DEGA_SIM)
  return (acc >> 4) | (m << 28);
#else
  return __builtin_amdgcn_alignbit(m, acc, 4);
#endif
}

// bit 7 of every byte of the result is set where the byte of y is zero; a byte directly above a zero byte may be
// flagged too when it is 0x01 (borrow) -- callers verify candidates
DG_DEV uint32_t lz_zero_bytes_approx(uint32_t y)
{
  return andn_and(y, y - 0x01010101u, 0x80808080u);
}

DG_DEV uint32_t lz_zero_bytes_exact(uint32_t y)
{
  return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// the four flags (bits 7, 15, 23, 31) gathered into bits 0..3 of the result; bits above are junk
DG_DEV uint32_t lz_gather_flags(uint32_t f)
{
  return mulhi32(f, 0x02040810u);
}

DG_DEV uint32_t lz_mask_from(int32_t x) // bits x..31 set (x <= 0: all, x >= 32: none)
{
  const uint32_t c = (uint32_t)(x < 0 ? 0 : (x > 32 ? 32 : x));
  return (uint32_t)(0xFFFFFFFFull << c);
}

// number of equal leading bytes (memory order) of two 16-byte strings held as 4 little-endian dwords, 0..16
DG_DEV uint32_t lz_common16(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
  const uint32_t x0 = a0 ^ b0, x1 = a1 ^ b1, x2 = a2 ^ b2, x3 = a3 ^ b3;
  uint32_t r = 16u;
  r = x3 != 0 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : r;
  r = x2 != 0 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : r;
  r = x1 != 0 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : r;
  r = x0 != 0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : r;
  return r;
}

// the same for 8-byte strings (2 dwords), 0..8
DG_DEV uint32_t lz_common8(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1)
{
  // the lowest differing bit of the 64: v_ffbl_b32 gives all ones for a zero word, so no selects are needed
  const uint32_t x0 = a0 ^ b0, x1 = a1 ^ b1;
  const uint32_t f0 = lowest_bit_or_ones(x0), f1 = lowest_bit_or_ones(x1) | 32u;
  const uint32_t f = f0 < f1 ? f0 : f1;
  return (f < 64u ? f : 64u) >> 3;
}

// the static prefix code of list position p < 19 (lzmh.c:86-106): returns the code, its length in len
DG_DEV uint32_t lz_list_code(uint32_t p, uint32_t &len)
{
  uint32_t code = 0x0Fu - p;
  len = 4;
  if (p >= 4)
  {
    code = 0x17u - (p - 4u);
    len = 5;
  }
  if (p >= 9)
  {
    code = 0x25u - (p - 9u);
    len = 6;
  }
  if (p >= 13)
  {
    code = 0x43u - (p - 13u);
    len = 7;
  }
  if (p >= 15)
  {
    code = 0x83u - (p - 15u);
    len = 8;
  }
  return code;
}

// the inverse: list position from the top 8 bits t >= 0x80 of the code register.  The 19 codes are a complete prefix
// code (4 x 4 bits 11xx, 5 x 5 bits 10011..10111, 4 x 6 bits 100010..100101, 2 x 7 bits, 4 x 8 bits), so the
// reference's scan over the table (lzmh.c:421-446) finds exactly this entry -- or none, when fewer bits are left than the
// code is long, which the caller checks
DG_DEV uint32_t lz_list_position(uint32_t t, uint32_t &len)
{
  uint32_t p = 15u + (0x83u - t);
  len = 8;
  if (t >= 0x84u)
  {
    p = 13u + (0x43u - (t >> 1));
    len = 7;
  }
  if (t >= 0x88u)
  {
    p = 9u + (0x25u - (t >> 2));
    len = 6;
  }
  if (t >= 0x98u)
  {
    p = 4u + (0x17u - (t >> 3));
    len = 5;
  }
  if (t >= 0xC0u)
  {
    p = 0x0Fu - (t >> 4);
    len = 4;
  }
  return p;
}

// The kernel runs as PAIRS of waves, like the DEGA kernels: of 64 channels' two waves (wave p and wave p + 4 of the
// workgroup, which the CU places on the same SIMD) the first SEARCHES -- window, candidates, the longest match -- and
// hands one token per step (match length | offset | the byte at P) to the second through a small LDS ring; the second
// CODES: recent-offset cache, frequency list, bits, output.  The parse does not depend on the coder's state (greedy
// longest match, nearest of equals), so the searcher never waits for anything but ring room, and each wave fills the
// issue slots the other leaves empty while it waits for LDS.
//   searcher publishes: tokens written (mod 2^16) | no more will come << 24
//   coder publishes:    tokens taken (mod 2^16)   | the lane wants no more (output full) << 16
#ifndef LZ_TOK_RING_OVERRIDE
constexpr uint32_t LZ_TOK_RING = 8;
#else
constexpr uint32_t LZ_TOK_RING = LZ_TOK_RING_OVERRIDE; // (stress builds of the emulator)
#endif
constexpr uint32_t LZ_PUB_DONE = 1u << 24, LZ_PUB_FINAL = 1u << 16;
#ifndef DG_LZ_SEARCH_PRIO // issue priority of the searching / reading waves (tools/tunebench.py)
#define DG_LZ_SEARCH_PRIO 0
#endif
#ifndef DG_LZ_CODE_PRIO // ... of the encoder's coding wave and the decoder's writing wave
#define DG_LZ_CODE_PRIO 0
#endif
#ifndef DG_LZ_WRITE_PRIO
#define DG_LZ_WRITE_PRIO 0 // (the reading wave is the chain now: with the writing wave above it 10.6 ms on the probe batch, level 8.85)
#endif
#ifndef DG_LZ_SEARCH_SLEEP // how long a wave with nothing to do sleeps (units of 64 cycles)
#define DG_LZ_SEARCH_SLEEP 1
#endif
#ifndef DG_LZ_CODE_SLEEP
#define DG_LZ_CODE_SLEEP 2
#endif
#ifndef DG_LZ_READ_SLEEP
#define DG_LZ_READ_SLEEP 1
#endif
#ifndef DG_LZ_WRITE_SLEEP
#define DG_LZ_WRITE_SLEEP 1
#endif
#ifndef DG_LZ_READ_PRIO
#define DG_LZ_READ_PRIO 0
#endif
constexpr uint32_t LZ_OFF_TOK = LZ_LDS_DW, LZ_OFF_PUB = LZ_OFF_TOK + LZ_TOK_RING * LZ_BLOCK, LZ_ENC_LDS_DW = LZ_OFF_PUB + 2 * LZ_BLOCK;
static_assert(LZ_ENC_LDS_DW * 4 <= 160 * 1024, "LDS budget of one CU");
constexpr uint32_t LZ_ENC_THREADS = 2 * LZ_BLOCK; // LZ_BLOCK channels per workgroup, two waves per 64 of them

#define LZ_WIN8(b) win8[((b) >> 2) * (4u * LZ_BLOCK) + ((b) & 3u)]
#define LZ_SYM8(i) sym8[((i) >> 2) * (4u * LZ_BLOCK) + ((i) & 3u)]
#define LZ_CNT(i) cnt16[((i) >> 1) * (2u * LZ_BLOCK) + ((i) & 1u)]

// ---- the searching wave ----------------------------------------------------------------------------------------------
DG_DEV void lzmh_searching_wave(const LzmhEncodeArgs &a, uint32_t *lds, uint32_t slot, size_t c, bool live)
{
  uint32_t *const win = lds + LZ_OFF_WIN + slot;                                  // dword d of this lane: win[d * LZ_BLOCK]
  uint8_t *const win8 = reinterpret_cast<uint8_t *>(lds + LZ_OFF_WIN + slot);     // byte b: win8[(b >> 2) * 4 * LZ_BLOCK + (b & 3)]
  uint32_t *const tok = lds + LZ_OFF_TOK + slot;
  uint32_t *const pub_mine = lds + LZ_OFF_PUB + slot;
  const uint32_t *const pub_peer = pub_mine + LZ_BLOCK;

  const uint64_t n64 = live ? a.in_len[c] : 0;
  const uint32_t n = (uint32_t)n64;
  const uint8_t *const src = a.in + (live ? c : 0) * a.stride;
  const bool bad = n64 > a.stride || n64 > 0x7FFFFFF0ull;
  // lzmh.c:161-174: exactly one ring of input wraps the write index onto the read index and the main loop never runs
  const uint32_t n_eff = (n == LZ_RING || bad) ? 0u : n;

  uint32_t P = 0;          // bytes consumed
  int32_t base = -128;     // absolute position of window byte 0 (multiple of 16)
  uint32_t wr = 0;         // tokens handed over
  bool reload = true;
  bool stop = false;       // the coder wants no more of this channel
  // A step in the making: the candidates of a step are measured six per pass of the wave's loop, and a lane whose step
  // has more of them (a line start of the ASCII workload has 16, the average step 2) simply takes more passes -- the
  // other lanes go on to their next steps meanwhile.  (In one lockstep step per lane the wave measured 25 candidates
  // per step for an average lane that has 2.)
  bool verifying = false;
  wave_priority<DG_LZ_SEARCH_PRIO>();
  uint32_t cm[5] = {0, 0, 0, 0, 0}; // candidates left: bit i of the 132-bit mask = window position 4 * wd0 + i
  uint32_t best = 2, besto = 0;
  uint32_t T0 = 0, T1 = 0, T2 = 0, T3 = 0; // the next 16 input bytes
  uint32_t rel = LZ_HISTORY, wd0 = 0, lim = 0, maxlen = 0;
  bool search = false;
  DG_STAMP_DECL;

  uint32_t peer = peer_load(pub_peer); // one pass old when it is used (the coder's count only grows: the room test errs on the safe side)
  for (;;)
  {
    stop = stop || (peer & LZ_PUB_FINAL) != 0u;
    const bool todo = P < n_eff && !stop;
    if (!wave_any(todo))
      break;
    const bool start = todo && !verifying && ((wr - peer) & 0xFFFFu) < LZ_TOK_RING; // begins a step in this pass
    if (!wave_any(start || (todo && verifying)))
    {
      wave_sleep<DG_LZ_SEARCH_SLEEP>();
      peer = peer_load(pub_peer);
      continue;
    }
    DG_STAMP(7);
    peer = peer_load(pub_peer); // for the next pass: comes back together with this pass's window reads

    // ---- window: when a lane that begins a step needs it, every lane with work reloads [P-128 (rounded down to 16), +432).
    //      A lane in the middle of a step takes part as well -- its candidates all lie in the last 132 bytes before P,
    //      which the new window holds too; only their window-relative indices move with the base (a multiple of 16) ----
    if (wave_any(start && (reload || (int32_t)(P + LZ_AHEAD) - base > (int32_t)LZ_WIN_BYTES)))
    {
      DG_STAMP(0);
      if (start || (todo && verifying))
      {
        base = (int32_t)((P - LZ_HISTORY) & ~15u);
#pragma unroll
        for (uint32_t k = 0; k < LZ_WIN_DW / 4; k++)
        {
          const int32_t at = base + (int32_t)(16u * k);
          // rows before the start of the channel (at < 0) and past the end of its row read row 0 / the last 16 bytes
          // instead and are zeroed: no branch around the load, so the 27 loads are in flight together
          const bool inside = at >= 0 && (size_t)at + 16u <= a.stride;
          const size_t from = inside ? (size_t)at : 0u;
          const lz_u32x4 v = *reinterpret_cast<const lz_u32x4 *>(src + from);
          const uint32_t keep = inside ? 0xFFFFFFFFu : 0u;
          const uint32_t v0 = v[0] & keep, v1 = v[1] & keep, v2 = v[2] & keep, v3 = v[3] & keep;
          win[(4u * k + 0u) * LZ_BLOCK] = v0;
          win[(4u * k + 1u) * LZ_BLOCK] = v1;
          win[(4u * k + 2u) * LZ_BLOCK] = v2;
          win[(4u * k + 3u) * LZ_BLOCK] = v3;
        }
        if (verifying)
        {
          rel = (uint32_t)((int32_t)P - base);
          wd0 = (rel - LZ_HISTORY) >> 2; // (lim stays: lengths measured so far were cut at it)
        }
        reload = false; // (a lane that waits for ring room keeps its request)
      }
      DG_STAMP(6);
    }

    DG_STAMP(0);
    // ---- a new step: its bounds (see the header) and phase 1, the 3-byte candidates among the last 132 positions.
    //      Worked out by every lane, taken over by those that begin a step ----
    {
      const uint32_t maxoff = P < LZ_HISTORY ? P : LZ_HISTORY;
      const uint32_t wabs_raw = (P > LZ_HISTORY ? P - LZ_HISTORY : 0u) + LZ_RING;
      const uint32_t wabs = wabs_raw < n ? wabs_raw : n;
      uint32_t maxlen_n = wabs - P;
      maxlen_n = maxlen_n > LZ_MAX_LENGTH ? LZ_MAX_LENGTH : maxlen_n;
      const uint32_t in_window = (uint32_t)(base + (int32_t)LZ_WIN_BYTES - (int32_t)P); // bytes of look-ahead the window holds
      const uint32_t lim_n = maxlen_n < in_window ? maxlen_n : in_window;
      const bool search_n = start && maxoff > 0 && maxlen_n > 2;

      const uint32_t rel_n = start ? (uint32_t)((int32_t)P - base) : LZ_HISTORY; // window byte index of position P (128..400)
      const uint32_t wd0_n = (rel_n - LZ_HISTORY) >> 2;                          // first history dword
      const uint32_t s = rel_n & 3u;
      uint32_t d[38];
#pragma unroll
      for (uint32_t j = 0; j < 38; j++)
        d[j] = win[(wd0_n + j) * LZ_BLOCK];
#pragma unroll
      for (uint32_t j = 0; j < 38; j++)
        DG_MATERIALISE(d[j]); // all 19 reads in flight together: left alone, hipcc issues them one at a time between the compares
      const uint32_t U0 = lz_alignbyte(d[33], d[32], s), U1 = lz_alignbyte(d[34], d[33], s), U2 = lz_alignbyte(d[35], d[34], s),
                     U3 = lz_alignbyte(d[36], d[35], s);
      const uint32_t A0 = (U0 & 0xFFu) * 0x01010101u, A1 = ((U0 >> 8) & 0xFFu) * 0x01010101u, A2 = ((U0 >> 16) & 0xFFu) * 0x01010101u;
      uint32_t cn[5] = {0, 0, 0, 0, 0}; // bit i of the 132-bit mask: window position 4*wd0 + i starts a 3-byte match
#pragma unroll
      for (uint32_t j = 0; j < 33; j++)
      {
        const uint32_t y = xor_then_or(lz_alignbyte(d[j + 1], d[j], 2), A2, xor_then_or(lz_alignbyte(d[j + 1], d[j], 1), A1, d[j] ^ A0));
        cn[j >> 3] = lz_shift_in_nibble(cn[j >> 3], lz_gather_flags(lz_zero_bytes_approx(y)));
      }
      cn[4] >>= 28;
      {
        // position i is offset 128 + s - i: valid offsets are 1..maxoff
        const int32_t lo = (int32_t)(LZ_HISTORY + s - maxoff);
        const uint32_t keep = search_n ? 0xFFFFFFFFu : 0u;
        cn[0] &= lz_mask_from(lo) & keep;
        cn[1] &= lz_mask_from(lo - 32) & keep;
        cn[2] &= lz_mask_from(lo - 64) & keep;
        cn[3] &= lz_mask_from(lo - 96) & keep;
        cn[4] &= ((1u << s) - 1u) & keep;
      }
      if (start)
      {
#pragma unroll
        for (uint32_t r = 0; r < 5; r++)
          cm[r] = cn[r];
        T0 = U0;
        T1 = U1;
        T2 = U2;
        T3 = U3;
        rel = rel_n;
        wd0 = wd0_n;
        lim = lim_n;
        maxlen = maxlen_n;
        search = search_n;
        best = 2;
        besto = 0;
        verifying = true;
      }
    }

    DG_STAMP(1);
    // ---- phase 2, one pass: up to LZ_POP candidates of the nearest mask register that still has some, nearest first, every
    // one measured against the input bytes held in registers (all reads of the pass in flight together: one LDS
    // latency); `len > best` in that order keeps the nearest of equals, like the reference's ascending scan (:196-214) ----
    if (wave_any(verifying && best < lim && (cm[0] | cm[1] | cm[2] | cm[3] | cm[4]) != 0u))
    {
      const bool go = verifying && best < lim;
      // the nearest mask register that still has candidates, and the next one: a lane whose nearest register holds fewer
      // than LZ_POP goes on in the next (still nearest first), so that a step's few candidates rarely need a second pass
      // (one upward sweep with selects: the nested conditions of the obvious form came out as masked branches)
      uint32_t r32 = 0u, s32 = 0u, m = cm[0], m2 = 0u; // r32, s32: 32 * the register
#pragma unroll
      for (uint32_t k = 1; k < 5; k++)
      {
        const bool nz = cm[k] != 0u;
        m2 = nz ? m : m2;
        s32 = nz ? r32 : s32;
        m = nz ? cm[k] : m;
        r32 = nz ? 32u * k : r32;
      }
      m = go ? m : 0u;
      m2 = go ? m2 : 0u;
      uint32_t qb[LZ_POP], e[LZ_POP][3];
      uint32_t has = 0;
#pragma unroll
      for (uint32_t k = 0; k < LZ_POP; k++)
      {
        const bool first = m != 0u;
        const uint32_t mm = first ? m : m2;
        has |= mm != 0 ? 1u << k : 0u;
        const uint32_t bit = 31u - clz32(mm | 1u);
        const uint32_t left = mm & ~(1u << bit);
        m = first ? left : m;
        m2 = first ? m2 : left;
        qb[k] = 4u * wd0 + (first ? r32 : s32) + bit; // window byte index of the candidate (a valid address also when there is none)
        const uint32_t qd = qb[k] >> 2;
#pragma unroll
        for (uint32_t j = 0; j < 3; j++)
          e[k][j] = win[(qd + j) * LZ_BLOCK];
      }
      if (go)
      {
        cm[4] = r32 == 128u ? m : cm[4];
        cm[3] = r32 == 96u ? m : (s32 == 96u && r32 > 96u) ? m2 : cm[3];
        cm[2] = r32 == 64u ? m : (s32 == 64u && r32 > 64u) ? m2 : cm[2];
        cm[1] = r32 == 32u ? m : (s32 == 32u && r32 > 32u) ? m2 : cm[1];
        cm[0] = r32 == 0u ? m : (s32 == 0u && r32 > 0u) ? m2 : cm[0];
      }
      // first the leading 8 bytes of every candidate (3 window dwords each): all that 97 % of them have to show
      uint32_t len[LZ_POP];
      uint32_t longer = 0; // candidates that match all 8 and may go on
#pragma unroll
      for (uint32_t k = 0; k < LZ_POP; k++)
      {
        const uint32_t qs = qb[k] & 3u;
        len[k] = lz_common8(lz_alignbyte(e[k][1], e[k][0], qs), lz_alignbyte(e[k][2], e[k][1], qs), T0, T1);
        longer |= (len[k] == 8u && ((has >> k) & 1u) != 0 && lim > 8u) ? 1u << k : 0u;
      }
      // then, one candidate per lane and round, the next 8 bytes (and beyond them byte by byte)
      while (wave_any(longer != 0u))
      {
        if (longer != 0u)
        {
          const uint32_t k = (uint32_t)__builtin_ctz(longer);
          longer &= longer - 1u;
          uint32_t q = qb[0];
#pragma unroll
          for (uint32_t i = 1; i < LZ_POP; i++)
            q = k == i ? qb[i] : q;
          const uint32_t qd = q >> 2, qs = q & 3u;
          const uint32_t f2 = win[(qd + 2u) * LZ_BLOCK], f3 = win[(qd + 3u) * LZ_BLOCK], f4 = win[(qd + 4u) * LZ_BLOCK];
          uint32_t l = 8u + lz_common8(lz_alignbyte(f3, f2, qs), lz_alignbyte(f4, f3, qs), T2, T3);
          if (l == 16u)
            while (l < lim && LZ_WIN8(q + l) == LZ_WIN8(rel + l))
              l++;
#pragma unroll
          for (uint32_t i = 0; i < LZ_POP; i++)
            len[i] = k == i ? l : len[i];
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < LZ_POP; k++)
      {
        const uint32_t l = len[k] < lim ? len[k] : lim;
        if (((has >> k) & 1u) != 0 && l > best)
        {
          best = l;
          besto = rel - qb[k];
        }
      }
    }
    DG_STAMP(2);
    // ---- the step is complete when no candidate is left, or the match cannot get longer ----
    const bool complete = verifying && (best >= lim || (cm[0] | cm[1] | cm[2] | cm[3] | cm[4]) == 0u);
    // a match that ran into the end of the window before the reference's own limit: reload around P and do the step again
    const bool again = complete && search && best >= lim && lim < maxlen;
    if (again)
      reload = true;
    if (complete && !again)
    {
      tok[(wr % LZ_TOK_RING) * LZ_BLOCK] = (best << 16) | (besto << 8) | (T0 & 0xFFu);
      wr++;
      P += best >= 3u ? best : 1u;
    }
    verifying = verifying && !complete;
    peer_store(pub_mine, (wr & 0xFFFFu) | ((P >= n_eff || stop) ? LZ_PUB_DONE : 0u));
    DG_STAMP(3);
  }
  peer_store(pub_mine, (wr & 0xFFFFu) | LZ_PUB_DONE);
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
  if (live)
  {
    if ((slot & 63u) < 8)
      a.out_bits[c] = stamp_sum[slot & 63u];
    else if ((slot & 63u) < 16)
      a.out_bits[c] = stamp_cnt[(slot & 63u) - 8];
  }
#endif
}

// ---- the coding wave -------------------------------------------------------------------------------------------------
DG_DEV void lzmh_coding_wave(const LzmhEncodeArgs &a, uint32_t *lds, uint32_t slot, size_t c, bool live)
{
  uint32_t *const symd = lds + LZ_OFF_SYM + slot;
  uint8_t *const sym8 = reinterpret_cast<uint8_t *>(lds + LZ_OFF_SYM + slot);
  uint16_t *const cnt16 = reinterpret_cast<uint16_t *>(lds + LZ_OFF_CNT + slot);  // count i: cnt16[(i >> 1) * 2 * LZ_BLOCK + (i & 1)]
  uint32_t *const stage = lds + LZ_OFF_STAGE + slot;
  const uint32_t *const tok = lds + LZ_OFF_TOK + slot;
  const uint32_t *const pub_peer = lds + LZ_OFF_PUB + slot;
  uint32_t *const pub_mine = lds + LZ_OFF_PUB + LZ_BLOCK + slot;

  const uint64_t n64 = live ? a.in_len[c] : 0;
  uint8_t *const dst = a.out + (live ? c : 0) * a.cap;
  int32_t err = (n64 > a.stride || n64 > 0x7FFFFFF0ull) ? ERR_INVALID_VALUE : OK;

  uint32_t mru = 0;        // the four most recent offsets, most recent in the low byte (lzmh.c:135)
  uint32_t nlist = 0;      // used entries of the frequency list
  uint64_t acc = 0;        // output bits, MSB first
  uint32_t nacc = 0, staged = 0, pos = 0; // bits in acc, words in stage[], words stored
  uint32_t rd = 0;         // tokens taken
  uint32_t peer = peer_load(pub_peer); // (one pass old when it is used: see the DEGA coding waves)
  wave_priority<DG_LZ_CODE_PRIO>();

  for (;;)
  {
    const bool emit = ((peer - rd) & 0xFFFFu) != 0u && err == OK;
    const bool peer_done = (peer & LZ_PUB_DONE) != 0u;
    const uint32_t token = tok[(rd % LZ_TOK_RING) * LZ_BLOCK];
    peer = peer_load(pub_peer);
    if (!wave_any(emit))
    {
      if (wave_all(peer_done || err != OK))
        break;
      wave_sleep<DG_LZ_CODE_SLEEP>();
      continue;
    }
    const uint32_t best = token >> 16, besto = (token >> 8) & 0xFFu, T0 = token & 0xFFu;
    rd += emit ? 1u : 0u;

    // ---- code the step (lzmh.c:216-340) ----
    uint32_t code = 0, codelen = 0;
    const bool lit = emit && best < 3;
    if (emit && best >= 3)
    {
      const uint32_t o0 = mru & 0xFFu, o1 = (mru >> 8) & 0xFFu, o2 = (mru >> 16) & 0xFFu, o3 = mru >> 24;
      if (o0 == besto)
      {
        code = 0x06u;
        codelen = 4;
      }
      else if (o1 == besto)
      {
        code = 0x0Eu;
        codelen = 5;
        mru = (mru & 0xFFFF0000u) | (o0 << 8) | besto;
      }
      else if (o2 == besto)
      {
        code = 0x1Eu;
        codelen = 6;
        mru = (mru & 0xFF000000u) | ((mru & 0xFFFFu) << 8) | besto;
      }
      else if (o3 == besto)
      {
        code = 0x1Fu;
        codelen = 6;
        mru = (mru << 8) | besto;
      }
      else
      {
        code = 0x100u | (besto - 1u);
        codelen = 10;
        mru = (mru << 8) | besto;
      }
      if (best < 11)
      {
        code = (code << 4) | (best - 3u);
        codelen += 4;
      }
      else if (best < 19)
      {
        code = (code << 5) | 0x10u | (best - 11u);
        codelen += 5;
      }
      else
      {
        code = (code << 10) | 0x300u | (best - 19u);
        codelen += 10;
      }
    }
    if (wave_any(lit))
    {
      // literal: position of the symbol in the frequency list (lzmh.c:285-333)
      const uint32_t sym = T0 & 0xFFu, splat = sym * 0x01010101u;
      uint32_t found = 0xFFFFu;
#pragma unroll
      for (uint32_t g = 0; g < LZ_SYM_DW / 4; g++)
      {
        if (!wave_any(lit && found == 0xFFFFu && 16u * g < nlist))
          break;
#pragma unroll
        for (uint32_t k = 4 * g + 3; k + 1 > 4 * g; k--)
        {
          const uint32_t z = lz_zero_bytes_exact(symd[k * LZ_BLOCK] ^ splat);
          if (z != 0 && 16u * g < nlist)
          {
            const uint32_t p = 4u * k + ((uint32_t)__builtin_ctz(z) >> 3);
            found = (p < nlist && p < found) ? p : found; // entries at and above nlist hold stale bytes
          }
        }
      }
      if (lit)
      {
        if (found != 0xFFFFu)
        {
          // the count, and the three entries in front of it, in one round of LDS reads: bubbling past more than three
          // entries (all of them have the same count) is rare and continues one read at a time
          const uint32_t p1 = found > 0 ? found - 1u : 0u, p2 = found > 1 ? found - 2u : 0u, p3 = found > 2 ? found - 3u : 0u;
          const uint32_t c0 = LZ_CNT(found), c1 = LZ_CNT(p1), c2 = LZ_CNT(p2), c3 = LZ_CNT(p3);
          const uint32_t s1 = LZ_SYM8(p1), s2 = LZ_SYM8(p2), s3 = LZ_SYM8(p3);
          if (c0 < 65535u)
          {
            uint32_t i = found; // towards the front past entries with a smaller count: only symbols move (:306-309)
            if (i > 0 && c0 + 1u > c1)
            {
              LZ_SYM8(i) = (uint8_t)s1;
              i--;
              if (i > 0 && c0 + 1u > c2)
              {
                LZ_SYM8(i) = (uint8_t)s2;
                i--;
                if (i > 0 && c0 + 1u > c3)
                {
                  LZ_SYM8(i) = (uint8_t)s3;
                  i--;
                  while (i > 0 && c0 + 1u > LZ_CNT(i - 1u))
                  {
                    LZ_SYM8(i) = LZ_SYM8(i - 1u);
                    i--;
                  }
                }
              }
            }
            LZ_CNT(i) = (uint16_t)(c0 + 1u);
            LZ_SYM8(i) = (uint8_t)sym;
          }
        }
        else if (nlist < LZ_LIST)
        {
          LZ_SYM8(nlist) = (uint8_t)sym;
          LZ_CNT(nlist) = 1;
          nlist++;
        }
        if (found < LZ_TREE)
          code = lz_list_code(found, codelen);
        else
        {
          code = sym; // 00 + byte
          codelen = 10;
        }
      }
    }


    // ---- output: bits -> 64-bit accumulator -> staged words in LDS -> 16-byte stores ----
    if (emit)
    {
      acc |= (uint64_t)code << (64u - nacc - codelen);
      nacc += codelen;
    }
    if (nacc >= 32u)
    {
      stage[staged * LZ_BLOCK] = (uint32_t)(acc >> 32);
      staged++;
      acc <<= 32;
      nacc -= 32u;
    }
    if (wave_any(staged == LZ_STAGE_DW))
    {
      if (staged == LZ_STAGE_DW)
      {
        if ((size_t)pos * 4u + 32u > a.cap) // keep room for the last words of finish
          err = ERR_MEMORY;
        else
        {
          const lz_u32x4 v = {bswap32(stage[0 * LZ_BLOCK]), bswap32(stage[1 * LZ_BLOCK]), bswap32(stage[2 * LZ_BLOCK]),
                              bswap32(stage[3 * LZ_BLOCK])};
          *reinterpret_cast<lz_u32x4 *>(dst + (size_t)pos * 4u) = v;
          pos += LZ_STAGE_DW;
        }
        staged = 0;
      }
    }
    peer_store(pub_mine, (rd & 0xFFFFu) | (err != OK ? LZ_PUB_FINAL : 0u));
  }
  peer_store(pub_mine, (rd & 0xFFFFu) | LZ_PUB_FINAL);

  // ---- finish: the staged words and the partial word, zero padded ----
  if (live)
  {
    uint64_t bits = 0;
    if (err == OK)
    {
      if ((size_t)(pos + staged + 1u) * 4u > a.cap)
        err = ERR_MEMORY;
      else
      {
        for (uint32_t k = 0; k < staged; k++)
          *reinterpret_cast<uint32_t *>(dst + (size_t)(pos + k) * 4u) = bswap32(stage[k * LZ_BLOCK]);
        if (nacc > 0)
          *reinterpret_cast<uint32_t *>(dst + (size_t)(pos + staged) * 4u) = bswap32((uint32_t)(acc >> 32));
        bits = 32ull * (pos + staged) + nacc;
      }
    }
    a.err[c] = err;
#if defined(DEGA_DIAG) && (DEGA_DIAG & 32) && !defined(DEGA_SIM)
    (void)bits; // diagnostic build: the searching wave dumps its stamps over out_bits
#else
    a.out_bits[c] = bits;
#endif
  }
}

__global__ void __launch_bounds__(LZ_ENC_THREADS) lzmh_encode_kernel(const LzmhEncodeArgs a)
{
  __shared__ uint32_t lds[LZ_ENC_LDS_DW];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = wave_uniform(threadIdx.x >> 6);
  const uint32_t slot = (wave % 4u) * 64u + lane; // the channel's column in every LDS array
  const bool codes = wave >= 4u;
  const size_t c = (size_t)blockIdx.x * LZ_BLOCK + slot;
  const bool live = c < a.C;
  if (!codes) // nothing handed over, nothing taken
  {
    lds[LZ_OFF_PUB + slot] = 0;
    lds[LZ_OFF_PUB + LZ_BLOCK + slot] = 0;
  }
  __syncthreads();
  if (!wave_any(live))
    return;
  if (codes)
    lzmh_coding_wave(a, lds, slot, c, live);
  else
    lzmh_searching_wave(a, lds, slot, c, live);
}
#undef LZ_WIN8
#undef LZ_SYM8
#undef LZ_CNT

// ---------------------------------------------------------------------------------------------------------------------
// LZMH decode (DCLib/src/lzmh.c:383-574), one lane per channel.  The decoder is a 32-bit code register that is topped up
// to 25 bits and consumed from the top (:410-415), a 128-byte history ring (:396) and the same frequency list; it is
// restated literally, including what it does at the end of a stream: it keeps decoding while the register holds a set
// bit (:571) and stops silently on an unknown list code (:447-449).  List entries that were never written read as symbol
// 0 here (the reference reads uninitialised stack there).
// The history ring is kept TWICE the reference's 128 bytes (:396): the writing wave appends up to 8 bytes a pass as whole
// dwords, zeros behind the last byte included, and those zeros land on ring bytes 245..256 back -- never read.
constexpr uint32_t LZD_RING_BYTES = 2 * LZ_HISTORY;
constexpr uint32_t LZD_HIST_DW = LZD_RING_BYTES / 4;
constexpr uint32_t LZD_OFF_HIST = 0, LZD_OFF_SYM = LZD_OFF_HIST + LZD_HIST_DW * LZ_BLOCK, LZD_OFF_CNT = LZD_OFF_SYM + LZ_SYM_DW * LZ_BLOCK,
                   LZD_LDS_DW = LZD_OFF_CNT + LZ_CNT_DW * LZ_BLOCK;

struct LzmhDecodeArgs
{
  const uint8_t *in;       // [C][cap] streams (32-bit big-endian words), cap a multiple of 4
  size_t cap;
  const uint64_t *in_bits; // [C]
  size_t C;
  uint8_t *out;            // [C][stride] decoded bytes, stride a multiple of 8
  size_t stride;
  uint64_t *out_len;       // [C]
  int32_t *err;            // [C]
};

// The decoder runs as pairs of waves too (see the encoder): of 64 channels' two waves on one SIMD the first READS --
// code register, recent offsets, frequency list: one token (a byte, or length << 8 | offset) per pass -- and the second
// WRITES: history ring, match copies, 8-byte output stores.  Each wave has every lane filled (the single-wave kernel ran
// batches of up to 64 Ki channels with half-filled waves to get two of them onto a SIMD, at twice the instructions).
//   reader publishes: tokens written (mod 2^16) | no more will come << 24
//   writer publishes: tokens taken (mod 2^16)   | the lane wants no more (output full) << 16
constexpr uint32_t LZD_TOK_RING = 8;
constexpr uint32_t LZD_OFF_TOK = LZD_LDS_DW, LZD_OFF_PUB = LZD_OFF_TOK + LZD_TOK_RING * LZ_BLOCK, LZD_OFF_PEND = LZD_OFF_PUB + 2 * LZ_BLOCK,
                   LZD_PAIR_LDS_DW = LZD_OFF_PEND + LZ_BLOCK;
constexpr uint32_t LZD_THREADS = 2 * LZ_BLOCK; // LZ_BLOCK channels per workgroup, two waves per 64 of them

#define LZ_SYM8(i) sym8[((i) >> 2) * (4u * LZ_BLOCK) + ((i) & 3u)]
#define LZ_CNT(i) cnt16[((i) >> 1) * (2u * LZ_BLOCK) + ((i) & 1u)]

// ---- the reading wave ------------------------------------------------------------------------------------------------
DG_DEV void lzmh_reading_wave(const LzmhDecodeArgs &a, uint32_t *lds, uint32_t slot, uint32_t slot0, size_t c, bool live) // slot0: the wave's first
{
  uint8_t *const sym8 = reinterpret_cast<uint8_t *>(lds + LZD_OFF_SYM + slot);
  uint16_t *const cnt16 = reinterpret_cast<uint16_t *>(lds + LZD_OFF_CNT + slot);
  const uint32_t *const symd = lds + LZD_OFF_SYM + slot;
  uint32_t *const tok = lds + LZD_OFF_TOK + slot;
  uint32_t *const pub_mine = lds + LZD_OFF_PUB + slot;
  const uint32_t *const pub_peer = pub_mine + LZ_BLOCK;

  const uint64_t nbits = live ? a.in_bits[c] : 0;
  const uint32_t *const src = reinterpret_cast<const uint32_t *>(a.in + (live ? c : 0) * a.cap);
  const uint32_t last_word = a.cap >= 4 ? (uint32_t)(a.cap / 4 - 1) : 0;
  const bool bad = nbits > 8ull * a.cap || a.cap < 4;

  uint64_t ip = 0;            // next input bit
  uint32_t wi = 0;            // index of w0
  // stream words wi, wi+1, wi+2 in registers; word wi+3 is on its way into the lane's dword of an LDS row (LDS-DMA: no
  // register waits for it) and is picked up at the lane's NEXT word step, a few passes on.  (Loaded into a register at the
  // step that needs it, every pass in which any lane stepped -- nearly every pass -- ended with a wait for device memory.)
  uint32_t *const pend_row = lds + LZD_OFF_PEND + slot0; // (wave uniform; lane l's dword is pend_row[l])
  const uint32_t *const pend = lds + LZD_OFF_PEND + slot;
  uint32_t w0 = 0, w1 = 0, w2 = 0;
  if (live && !bad)
  {
    w0 = bswap32(src[0]);
    w1 = bswap32(src[1 < last_word ? 1 : last_word]);
    w2 = bswap32(src[2 < last_word ? 2 : last_word]);
    dma_row_to_lds(reinterpret_cast<const int32_t *>(src + (3 < last_word ? 3 : last_word)), pend_row, slot - slot0);
  }
  uint32_t code_sym = 0;
  int32_t code_length = 0;
  uint32_t mru = 0;
  uint32_t nvalid = 0; // entries 0 .. nvalid-1 of the list are in use (count > 0), entry nvalid is the first free one
  uint32_t wr = 0;     // tokens handed over
  bool finished = !live || bad;
  bool stop = false;   // the writer wants no more of this channel
  wave_priority<DG_LZ_READ_PRIO>();
  DG_STAMP_DECL;

  uint32_t mail = 0;       // the lane's dword of the row as read at the top of this pass
  bool mail_fresh = false; // ... and not used yet
  // one pass of the reference's loop body up to the point where it writes (lzmh.c:410-560): the next token, if any
  auto next_token = [&](uint32_t &token, bool &emitted) {
    // ---- top the register up to 25 bits (lzmh.c:410-415); bits shifted in while code_length <= 0 fall off ----
    while (ip < nbits && code_length <= 24)
    {
      const uint64_t avail = nbits - ip;
      uint32_t k;
      bool lost = false;
      if (code_length < 0)
      {
        k = (uint32_t)(-code_length);
        lost = true;
      }
      else
        k = 25u - (uint32_t)code_length;
      k = avail < k ? (uint32_t)avail : k;
      const uint32_t sh = (uint32_t)(ip & 31u);
      const uint64_t two = ((uint64_t)w0 << 32) | w1;
      const uint32_t chunk = (uint32_t)((two << sh) >> (64u - k)); // k in 1..25
      if (!lost)
        code_sym |= chunk << (32u - (uint32_t)code_length - k);
      code_length += (int32_t)k;
      ip += k;
      const uint32_t nwi = (uint32_t)(ip >> 5);
      if (nwi != wi)
      {
        wi = nwi;
        w0 = w1;
        w1 = w2;
        uint32_t raw = mail; // (read at the top of the pass, with the partner's word: no LDS round trip of its own)
        if (!mail_fresh)     // a second word step in one pass (bits that fall off a damaged stream's register): rare
        {
          wait_vector_memory();
          raw = peer_load(pend);
          wait_lds(); // it is out of the row before the next one can land
        }
        mail_fresh = false;
        w2 = bswap32(raw);
        const uint32_t idx = wi + 3u;
        dma_row_to_lds(reinterpret_cast<const int32_t *>(src + (idx < last_word ? idx : last_word)), pend_row, slot - slot0);
      }
    }
    DG_STAMP(1);
    // The three kinds of code -- 1... a list position, 00 + byte, 01 a match -- without a branch per kind where that can be
    // helped: the lanes of a wave are spread over all of them.  The list entry's LDS reads go out first (every lane, a
    // harmless address for those with another kind), the match fields are worked out while they are on their way.
    const bool is_list = (code_sym & 0x80000000u) != 0;
    const bool is_match = !is_list && (code_sym & 0x40000000u) != 0;
    uint32_t llen;
    uint32_t li = lz_list_position(code_sym >> 24, llen);
    // (a code cut off by the end of the stream matches no table entry, lzmh.c:423: the reference returns NO_ERROR there)
    const bool unknown = is_list && code_length < (int32_t)llen;
    li = is_list ? li : 0u;
    // the entry, its count and the two entries in front of it in one round of LDS reads (the bubble-up rarely goes further)
    const uint32_t lp1 = li > 0 ? li - 1u : 0u, lp2 = li > 1 ? li - 2u : 0u;
    const uint32_t lsym = LZ_SYM8(li), c0 = LZ_CNT(li), c1 = LZ_CNT(lp1), c2 = LZ_CNT(lp2), s1 = LZ_SYM8(lp1), s2 = LZ_SYM8(lp2);

    // ---- a match (lzmh.c:487-553) ----
    // the offset field after the two bits "01": 0 + 7 bits = a new offset | 10 | 110 | 1110 | 1111 = the most recent
    // offset, the one before, ... (:489-531) -- the run of ones tells which
    const uint32_t x = code_sym << 2;
    const uint32_t ones = clz32(~x | 0x08000000u);                    // 0..4
    const uint32_t used_o = ones == 0u ? 8u : (ones == 4u ? 4u : ones + 1u);
    const uint32_t at = 8u * (ones == 0u ? 3u : ones - 1u);            // bit position of the entry that leaves its place
    const uint32_t offset = ones == 0u ? ((x >> 24) & 0x7Fu) + 1u : (mru >> at) & 0xFFu;
    // move to front: the entries in front of it move up one place, those behind stay (a new offset: the oldest drops out)
    const uint32_t mru_m = (mru & (0xFFFFFF00u << at)) | ((mru & ((1u << at) - 1u)) << 8) | offset;
    // the length field: 0 + 3 bits = 3..10 | 10 + 3 bits = 11..18 | 11 + 8 bits = 19..274 (:533-553)
    const uint32_t y = x << used_o;
    const uint32_t top = y >> 30;
    const uint32_t length = top < 2u ? ((y >> 28) & 7u) + 3u : (top == 2u ? ((y >> 27) & 7u) + 11u : ((y >> 22) & 0xFFu) + 19u);
    const uint32_t used_m = 2u + used_o + (top < 2u ? 4u : (top == 2u ? 5u : 10u));
    mru = is_match ? mru_m : mru;

    // ---- every kind: the bits it used, its token ----
    const uint32_t used = is_list ? (unknown ? 0u : llen) : (is_match ? used_m : 10u);
    const uint32_t raw_sym = (code_sym >> 22) & 0xFFu;
    token = is_list ? lsym : (is_match ? (length << 8) | offset : raw_sym);
    emitted = !unknown;
    finished = finished || unknown;
    code_length -= (int32_t)used;
    code_sym <<= used;

    if (is_list && !unknown) // the list entry moves towards the front past entries with a smaller count: only symbols move
    {
      uint32_t i = li;
      if (c0 < 65535u)
      {
        if (i > 0 && c0 + 1u > c1)
        {
          LZ_SYM8(i) = (uint8_t)s1;
          i--;
          if (i > 0 && c0 + 1u > c2)
          {
            LZ_SYM8(i) = (uint8_t)s2;
            i--;
            while (i > 0 && c0 + 1u > LZ_CNT(i - 1u))
            {
              LZ_SYM8(i) = LZ_SYM8(i - 1u);
              i--;
            }
          }
        }
        LZ_CNT(i) = (uint16_t)(c0 + 1u);
        LZ_SYM8(i) = (uint8_t)lsym;
        if (i == nvalid) // only a damaged stream names an entry that is not in use yet
          for (nvalid++; nvalid < LZ_LIST && LZ_CNT(nvalid) > 0; nvalid++)
            ;
      }
    }
    if (!is_list && !is_match) // 00 + byte
    {
      const uint32_t sym = raw_sym;
      // the reference walks the list until the symbol or the first unused entry (:463-465); here the symbol dwords are
      // searched byte-parallel in one round of LDS reads and the first unused entry is known (nvalid)
      uint32_t i = nvalid;
      {
        const uint32_t splat = sym * 0x01010101u;
#pragma unroll
        for (uint32_t k = LZ_SYM_DW; k-- > 0;)
        {
          const uint32_t z = lz_zero_bytes_exact(symd[k * LZ_BLOCK] ^ splat);
          const uint32_t p = 4u * k + ((uint32_t)__builtin_ctz(z | 0x80000000u) >> 3);
          i = (z != 0 && p < i) ? p : i; // a hit at or above nvalid is a stale byte
        }
      }
      if (i < LZ_LIST)
      {
        const uint32_t c0 = LZ_CNT(i);
        if (c0 < 65535u)
        {
          while (i > 0 && c0 + 1u > LZ_CNT(i - 1u)) // whole entries move here (:470-474)
          {
            LZ_SYM8(i) = LZ_SYM8(i - 1u);
            LZ_CNT(i) = LZ_CNT(i - 1u);
            i--;
          }
          LZ_CNT(i) = (uint16_t)(c0 + 1u);
          LZ_SYM8(i) = (uint8_t)sym;
          if (i == nvalid)
            for (nvalid++; nvalid < LZ_LIST && LZ_CNT(nvalid) > 0; nvalid++)
              ;
        }
      }
    }
  };

  for (;;)
  {
    wait_vector_memory(); // (the words asked for in the last pass: long there)
    const uint32_t peer = peer_load(pub_peer);
    mail = peer_load(pend);
    mail_fresh = true;
    stop = stop || (peer & LZ_PUB_FINAL) != 0u;
    const bool todo = !finished && !stop;
    if (!wave_any(todo))
      break;
    const bool active = todo && ((wr - peer) & 0xFFFFu) < LZD_TOK_RING;
    if (!wave_any(active))
    {
      wave_sleep<DG_LZ_READ_SLEEP>();
      DG_STAMP(7);
      continue;
    }
    DG_STAMP(0);
    if (active)
    {
      uint32_t token = 0;
      bool emitted = false;
      next_token(token, emitted);
      DG_STAMP(2);
      if (emitted)
      {
        tok[(wr % LZD_TOK_RING) * LZ_BLOCK] = token;
        wr++;
      }
      finished = finished || !(ip < nbits || code_sym > 0); // the loop condition of lzmh.c:571 (its error half is the writer's)
    }
    peer_store(pub_mine, (wr & 0xFFFFu) | ((finished || stop) ? LZ_PUB_DONE : 0u));
    DG_STAMP(3);
  }
  peer_store(pub_mine, (wr & 0xFFFFu) | LZ_PUB_DONE);
  wait_vector_memory(); // no DMA may still be writing to LDS when the workgroup's allocation is released
#if defined(DEGA_DIAG) && (DEGA_DIAG & 256) && !defined(DEGA_SIM)
  if (live) // diagnostic build: the stamps of the first 16 lanes of every wave instead of the decoded lengths
  {
    if ((slot & 63u) < 8)
      a.out_len[c] = stamp_sum[slot & 63u];
    else if ((slot & 63u) < 16)
      a.out_len[c] = stamp_cnt[(slot & 63u) - 8];
  }
#endif
}

// ---- the writing wave ------------------------------------------------------------------------------------------------
// One pass appends up to 8 bytes per lane -- a literal, or the next 8 of a match (a longer match keeps its lane for
// further passes) -- in one round of LDS reads: the 8 bytes `offset` back as three ring dwords, a match that overlaps its
// own output (offset < 8) as its period repeated; then the ring and the 8-byte output accumulator each take them as one
// shifted value.  (Byte by byte the copy was a chain of dependent LDS round trips, 15 instructions a byte.)
DG_DEV void lzmh_writing_wave(const LzmhDecodeArgs &a, uint32_t *lds, uint32_t slot, size_t c, bool live)
{
  uint32_t *const histd = lds + LZD_OFF_HIST + slot; // dword d of the ring: histd[d * LZ_BLOCK]
  const uint32_t *const tok = lds + LZD_OFF_TOK + slot;
  const uint32_t *const pub_peer = lds + LZD_OFF_PUB + slot;
  uint32_t *const pub_mine = lds + LZD_OFF_PUB + LZ_BLOCK + slot;

  const uint64_t nbits = live ? a.in_bits[c] : 0;
  uint8_t *const dst = a.out + (live ? c : 0) * a.stride;
  int32_t err = (nbits > 8ull * a.cap || a.cap < 4) ? ERR_INVALID_VALUE : OK;
  uint32_t hp = 0;      // ring position of the next byte
  uint32_t hcur = 0;    // the ring dword hp lies in: its bytes below hp, zeros above
  uint64_t obuf = 0;    // output bytes not stored yet, first byte lowest
  uint32_t nob = 0;
  uint64_t olen = 0;
  uint32_t rd = 0;
  uint32_t rem = 0, offset = 1; // the match being copied: bytes left, distance
  uint32_t peer = peer_load(pub_peer); // (one pass old when it is used: see the DEGA coding waves)
  wave_priority<DG_LZ_WRITE_PRIO>();

  for (;;)
  {
    const bool copying_on = rem > 0u;
    const bool has = live && !copying_on && ((peer - rd) & 0xFFFFu) != 0u && err == OK;
    const bool peer_done = (peer & LZ_PUB_DONE) != 0u;
    const uint32_t token = tok[(rd % LZD_TOK_RING) * LZ_BLOCK];
    peer = peer_load(pub_peer);
    if (!wave_any(has || copying_on))
    {
      if (wave_all(!live || peer_done || err != OK))
        break;
      wave_sleep<DG_LZ_WRITE_SLEEP>();
      continue;
    }
    const bool lit = has && token < 0x100u;
    if (has)
    {
      rd++;
      if (!lit)
      {
        // (offset 0 -- only a damaged stream has it, from a recent-offset entry never set -- reads the reference's
        // 128-byte ring at the write position itself: the byte 128 back)
        offset = (token & 0xFFu) != 0u ? (token & 0xFFu) : LZ_HISTORY;
        rem = token >> 8;
      }
    }
    const bool copying = rem > 0u;
    uint32_t vlo = token & 0xFFu, vhi = 0, n = 1;
    if (wave_any(copying))
    {
      const uint32_t from = (hp - offset) & (LZD_RING_BYTES - 1u);
      const uint32_t fd = from >> 2, fs = from & 3u;
      const uint32_t w0 = histd[fd * LZ_BLOCK], w1 = histd[((fd + 1u) & (LZD_HIST_DW - 1u)) * LZ_BLOCK],
                     w2 = histd[((fd + 2u) & (LZD_HIST_DW - 1u)) * LZ_BLOCK];
      uint64_t m = ((uint64_t)lz_alignbyte(w2, w1, fs) << 32) | lz_alignbyte(w1, w0, fs); // the 8 bytes `offset` back
      if (wave_any(copying && offset < 8u))
      {
        // the bytes from hp on are not written yet: a period of `offset` bytes, repeated (1 -> 2 -> 4 -> 8 periods)
        const uint32_t pb = 8u * (offset < 8u ? offset : 8u);
        uint64_t r = pb < 64u ? m & ((1ull << pb) - 1ull) : m;
        r |= pb < 64u ? r << pb : 0ull;
        r |= 2u * pb < 64u ? r << (2u * pb) : 0ull;
        r |= 4u * pb < 64u ? r << (4u * pb) : 0ull;
        m = r;
      }
      const uint32_t nm = rem < 8u ? rem : 8u;
      m = nm < 8u ? m & ((1ull << (8u * nm)) - 1ull) : m;
      vlo = copying ? (uint32_t)m : vlo;
      vhi = copying ? (uint32_t)(m >> 32) : vhi;
      n = copying ? nm : n;
    }
    if (lit || copying)
    {
      const uint64_t v = ((uint64_t)vhi << 32) | vlo;
      rem -= copying ? n : 0u;
      // ---- the ring: the 96 bits hcur | v << (8 * (hp & 3)) as three dwords ----
      {
        const uint32_t sh = 8u * (hp & 3u);
        const uint64_t lo64 = v << sh;
        const uint32_t x0 = hcur | (uint32_t)lo64, x1 = (uint32_t)(lo64 >> 32), x2 = (uint32_t)((v >> 32) >> (32u - sh));
        const uint32_t d0 = hp >> 2;
        histd[d0 * LZ_BLOCK] = x0;
        histd[((d0 + 1u) & (LZD_HIST_DW - 1u)) * LZ_BLOCK] = x1;
        histd[((d0 + 2u) & (LZD_HIST_DW - 1u)) * LZ_BLOCK] = x2;
        const uint32_t at = ((hp & 3u) + n) >> 2; // the dword of the three that the new hp lies in (0..2)
        hcur = at == 0u ? x0 : at == 1u ? x1 : x2;
        hp = (hp + n) & (LZD_RING_BYTES - 1u);
      }
      // ---- the output: the 128 bits obuf | v << (8 * nob); 8 bytes stored when there are that many ----
      {
        const uint32_t osh = 8u * nob;
        const uint64_t ylo = obuf | (v << osh);
        const uint64_t yhi = osh != 0u ? v >> (64u - osh) : 0ull;
        const uint32_t tot = nob + n;
        if (tot >= 8u)
        {
          if (olen + 8u > a.stride)
          {
            err = ERR_MEMORY;
            rem = 0;
          }
          else
            *reinterpret_cast<uint64_t *>(dst + olen) = ylo;
          olen += 8;
          obuf = yhi;
          nob = tot - 8u;
        }
        else
        {
          obuf = ylo;
          nob = tot;
        }
      }
    }
    peer_store(pub_mine, (rd & 0xFFFFu) | (err != OK ? LZ_PUB_FINAL : 0u));
  }
  peer_store(pub_mine, (rd & 0xFFFFu) | LZ_PUB_FINAL);

  if (live)
  {
    if (err == OK && nob > 0)
    {
      if (olen + 8u > a.stride)
        err = ERR_MEMORY;
      else
        *reinterpret_cast<uint64_t *>(dst + olen) = obuf;
      olen += nob;
    }
#if defined(DEGA_DIAG) && (DEGA_DIAG & 256) && !defined(DEGA_SIM)
    if ((slot & 63u) >= 16) // (the reading wave dumps its stamps over the first 16)
#endif
      a.out_len[c] = err == OK ? olen : 0;
    a.err[c] = err;
  }
}

// (Tried: 32 channels per wave and twice the waves, two reading and two writing waves per SIMD, for up to 64 Ki
// channels -- 15.4 ms against 11.1 on the probe batch: the reading wave's chain does not leave the gaps a second one
// could use, the four waves only get in each other's way.)
__global__ void __launch_bounds__(LZD_THREADS) lzmh_decode_kernel(const LzmhDecodeArgs a)
{
  __shared__ uint32_t lds[LZD_PAIR_LDS_DW];
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = wave_uniform(threadIdx.x >> 6);
  const uint32_t slot = (wave % 4u) * 64u + lane; // the channel's column in every LDS array
  const bool writes = wave >= 4u;
  const size_t c = (size_t)blockIdx.x * LZ_BLOCK + slot;
  const bool live = c < a.C;
  if (!writes) // history, list (entries never written read as symbol 0), token ring, published words: all zero
    for (uint32_t k = 0; k < LZD_PAIR_LDS_DW / LZ_BLOCK; k++)
      lds[k * LZ_BLOCK + slot] = 0;
  __syncthreads();
  if (!wave_any(live))
    return;
  if (writes)
    lzmh_writing_wave(a, lds, slot, c, live);
  else
    lzmh_reading_wave(a, lds, slot, slot - lane, c, live);
}
#undef LZ_SYM8
#undef LZ_CNT

// ASCII rendering of the synthetic meter channels for the LZMH workload (cfg 4): channel c's samples
// x[t][c] (centi-units) as "%d.%02d\n" lines -- the "ASCII float in" domain of the reference's LZMH (DCLib/doc/readme.md:30).
struct RenderArgs
{
  const int32_t *x; // [T][ld]
  size_t C, T, ld;
  uint8_t *out;     // [C][stride]
  size_t stride;
  uint64_t *out_len; // [C]
  int32_t *err;      // [C]: ERR_MEMORY when a channel's text does not fit
};

__global__ void __launch_bounds__(256) lzmh_render_kernel(const RenderArgs a)
{
  const size_t c = (size_t)blockIdx.x * 256u + threadIdx.x;
  if (c >= a.C)
    return;
  uint8_t *const dst = a.out + c * a.stride;
  size_t len = 0;
  int32_t err = OK;
  uint64_t acc = 0; // up to 8 pending bytes, first byte lowest
  uint32_t nacc = 0;
  for (size_t t = 0; t < a.T; t++)
  {
    const int32_t v = a.x[t * a.ld + c];
    uint32_t u = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    uint8_t txt[16];
    uint32_t k = 0;
    txt[k++] = '\n';
    txt[k++] = (uint8_t)('0' + u % 10u);
    u /= 10u;
    txt[k++] = (uint8_t)('0' + u % 10u);
    u /= 10u;
    txt[k++] = '.';
    do
    {
      txt[k++] = (uint8_t)('0' + u % 10u);
      u /= 10u;
    } while (u != 0);
    if (v < 0)
      txt[k++] = '-';
    if (len + nacc + k + 8u > a.stride)
    {
      err = ERR_MEMORY;
      break;
    }
    while (k > 0)
    {
      acc |= (uint64_t)txt[--k] << (8u * nacc);
      if (++nacc == 8u)
      {
        *reinterpret_cast<uint64_t *>(dst + len) = acc;
        len += 8;
        acc = 0;
        nacc = 0;
      }
    }
  }
  if (err == OK && nacc > 0)
  {
    *reinterpret_cast<uint64_t *>(dst + len) = acc;
    len += nacc;
  }
  a.out_len[c] = err == OK ? len : 0;
  a.err[c] = err;
}

} // namespace dg
