// dega_lane.hpp -- per-lane (= per-channel) coder state machines of the DEGA path, written for one GPU lane.
//
// One lane codes one meter channel.  Everything here is straight-line integer code on a handful of registers; the
// wave-level orchestration (row-lockstep SEG producer, word-lockstep BAC consumer, LDS bit rings) is in
// dega_kernels.hpp.  The same source is compiled by hipcc for gfx950 and -- for offline debugging of the kernels
// only, never as a fallback -- by g++ under the thread-per-lane emulator in tests/sim/ (DEGA_SIM).
//
// Reference behaviour reproduced here (paths relative to the reference's DataCompressor/):
//   diff  DCLib/src/diff.c:9-37      seg  DCLib/src/seg.c:11-94      bac  DCLib/src/bac.c:39-263
//   bit order DCIOLib/src/bit_file_buffer.c:220-248,297-308 (MSB first; 32-bit words therefore big-endian)
#pragma once

#include <stdint.h>

#if defined(DEGA_SIM)
#define DG_DEV inline
#else
#define DG_DEV __device__ __forceinline__
#endif

namespace dg
{

constexpr int32_t OK = 0;
constexpr int32_t ERR_INVALID_VALUE = -1;
constexpr int32_t ERR_INVALID_FORMAT = -3;
constexpr int32_t ERR_MEMORY = -6;

constexpr uint32_t MAX_FREQUENCY = 16383; // bac.c:27
constexpr uint32_t DIV_TABLE_SIZE = 16384;

// Exact floor(n / t) for 0 <= n < 2^30, 3 <= t <= 16383 as  mulhi(n, magic) >> shift  with
// magic = ceil(2^(30+L) / t), L = ceil(log2 t), shift = L - 2  (error term < 2^-L <= 1/t, so the floor is exact).
// This replaces the two 64-bit divisions per symbol of bac.c:110-111; numerators there are R*cum <= 2^16 * 2^13.
struct DivEntry
{
  uint32_t magic;
  uint32_t shift;
};

DG_DEV uint32_t clz32(uint32_t x) // x != 0
{
  return (uint32_t)__builtin_clz(x);
}

DG_DEV uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(DEGA_SIM)
  return (uint32_t)(((uint64_t)a * b) >> 32);
#else
  return __umulhi(a, b);
#endif
}

DG_DEV uint32_t bswap32(uint32_t x)
{
  return __builtin_bswap32(x);
}

DG_DEV uint32_t div_by_total(uint32_t n, const DivEntry &e)
{
  return mulhi32(n, e.magic) >> e.shift;
}

// ---------------------------------------------------------------------------------------------------------------------
// Output side of the arithmetic coder: a 64-bit carry-propagating accumulator instead of bit-plus-follow.
//
// The reference emits a bit per E1/E2 shift and defers E3 ("underflow") shifts in a counter that is resolved by the
// next emitted bit (bac.c:93-105,127-132).  The emitted stream is exactly the binary expansion of the running sum of
// the `start` increments, each added at the current window position; an E3 shift provisionally emits 0 then 1s and
// a later carry out of the window flips them -- which is what a plain multi-word addition does.  So the lane keeps
//   W = [ cnt already-shifted-out bits | 16-bit window | zeros ]   (left aligned in 64 bits)
// adds the increment at the window's position, and lets carries ripple.  A carry out of W (needs >= 17 consecutive
// one bits in flight) is fixed up in the words already stored, which this lane wrote itself.
// ---------------------------------------------------------------------------------------------------------------------
struct BitSink
{
  uint64_t W;          // accumulator, see above
  uint32_t cnt;        // number of finished bits at the top of W (0..31 between symbols)
  uint32_t pos;        // 32-bit words already stored
  uint32_t cap_words;  // capacity of the channel's slab in words
  uint32_t *dst;       // channel's slab
  int32_t err;

  DG_DEV void init(uint32_t *dst_, uint32_t cap_words_)
  {
    W = 0;
    cnt = 0;
    pos = 0;
    cap_words = cap_words_;
    dst = dst_;
    err = OK;
  }

  DG_DEV void carry_into_stored_words()
  {
    uint32_t p = pos;
    while (p > 0)
    {
      --p;
      if (p < cap_words)
      {
        const uint32_t w = bswap32(dst[p]) + 1u;
        dst[p] = bswap32(w);
        if (w != 0)
          break;
      }
    }
  }

  // add `inc` (< 2^17) at the window: window LSB sits at bit (48 - cnt)
  DG_DEV void add_at_window(uint32_t inc)
  {
    const uint64_t add = (uint64_t)inc << (48u - cnt);
    const uint64_t nw = W + add;
    if (nw < add)
      carry_into_stored_words();
    W = nw;
  }

  DG_DEV void store_word(uint32_t word)
  {
    if (pos < cap_words)
      dst[pos] = bswap32(word); // MSB-first bit order => big-endian words
    else if (err == OK)
      err = ERR_MEMORY;
    pos++;
  }

  // the window moved n bits to the right: n more finished bits
  DG_DEV void advance(uint32_t n)
  {
    cnt += n;
    if (cnt >= 32)
    {
      store_word((uint32_t)(W >> 32));
      W <<= 32;
      cnt -= 32;
    }
  }

  // total stream length in bits after `extra` final bits have been accounted for with advance()
  DG_DEV uint64_t finish()
  {
    const uint64_t total = (uint64_t)pos * 32u + cnt;
    if (cnt > 0)
    {
      const uint32_t word = (uint32_t)(W >> 32) & ~(0xFFFFFFFFu >> cnt); // zero padding (bit_file_buffer.c:310-320)
      store_word(word);
    }
    return total;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Adaptive model, binary case of bac.c:39-81.  Index 1 holds the more frequent bit value (`mps`), index 2 the other,
// index 3 is EOF with frequency 1 forever.  cum[0] = f1+f2+1, cum[1] = f2+1, cum[2] = 1, cum[3] = 0.
// ---------------------------------------------------------------------------------------------------------------------
struct Model
{
  uint32_t f1, f2, mps;

  DG_DEV void init() // bac.c:39-52
  {
    f1 = 1;
    f2 = 1;
    mps = 0;
  }

  DG_DEV uint32_t total() const
  {
    return f1 + f2 + 1;
  }

  DG_DEV void update(bool lps) // bac.c:54-81
  {
    if (f1 + f2 + 1 == MAX_FREQUENCY) // :57-67 halve, rounding up; EOF stays 1
    {
      f1 = (f1 + 1) >> 1;
      f2 = (f2 + 1) >> 1;
    }
    if (lps)
    {
      if (f2 == f1) // :68-77 the coded symbol moves to index 1
      {
        mps ^= 1u;
        f1++;
      }
      else
        f2++;
    }
    else
      f1++;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Interval state of the coder (bac.c:83-139) in a form whose renormalisation is branch free:
//   A = start << 16,   B = (65535 - end) << 16   (low 16 bits always zero)
// so that  range - 1 = ~(A + B) >> 16,  the E1/E2 shift count is clz(~(A ^ B))  (length of the common prefix of start
// and end), and the number of E3 steps that follow is the run of ones below bit 31 of (A & B).  After an E3 step the
// reference clears the top bit of both; here it is left set in both ("spurious" bit 31): every use either shifts it
// out or cancels it (A + B mod 2^32, A ^ B), which saves two instructions per symbol.
// ---------------------------------------------------------------------------------------------------------------------
struct Interval
{
  uint32_t A, B;

  DG_DEV void init() // bac.c:86-91
  {
    A = 0;
    B = 0;
  }

  DG_DEV uint32_t range() const // 1..65536
  {
    return ((~(A + B)) >> 16) + 1u;
  }

  // returns the number of window shifts (E1/E2 + E3)
  DG_DEV uint32_t renormalise() // bac.c:112-137
  {
    const uint32_t k = clz32(~(A ^ B)); // <= 16: the low halves differ by construction
    A <<= k;
    B <<= k;
    const uint32_t v = (A & B) | 0x80000000u;
    const uint32_t j = clz32(~v) - 1u; // run of ones below bit 31; ~v != 0 because the low 16 bits of v are zero
    A <<= j;
    B <<= j;
    return k + j;
  }
};

template <bool ADAPTIVE>
struct BacEncoder
{
  Interval iv;
  Model m;
  BitSink sink;

  DG_DEV void init(uint32_t *dst, uint32_t cap_words)
  {
    iv.init();
    m.init();
    sink.init(dst, cap_words);
  }

  // One data bit: EncodeSymbol + UpdateModel (bac.c:156-161).  tab = division table indexed by cum[0].
  DG_DEV void encode_bit(uint32_t bit, const DivEntry *tab)
  {
    const DivEntry de = tab[m.total()];
    const uint32_t R = iv.range();
    const uint32_t x1 = div_by_total(R * (m.f2 + 1u), de); // range * cum[1] / cum[0]
    const bool lps = bit != m.mps;
    uint32_t inc = x1;
    if (lps)
    {
      // index 2: end = start + x1 - 1, start += range * cum[2] / cum[0] with cum[2] = 1   (bac.c:110-111)
      iv.B = 0u - (iv.A + (x1 << 16));
      inc = div_by_total(R, de);
    }
    // index 1: end unchanged (cum[0]/cum[0]), start += x1
    iv.A += inc << 16;
    sink.add_at_window(inc);
    sink.advance(iv.renormalise());
    if (ADAPTIVE)
      m.update(lps);
  }

  // EOF symbol + FinishEncoding (bac.c:163-164, 141-145); returns the exact stream length in bits
  DG_DEV uint64_t finish(const DivEntry *tab)
  {
    const DivEntry de = tab[m.total()];
    const uint32_t x2 = div_by_total(iv.range(), de); // index 3: cum[2] = 1, cum[3] = 0
    iv.B = 0u - (iv.A + (x2 << 16));                   // end = start + x2 - 1, start unchanged
    sink.advance(iv.renormalise());
    // "pending++ ; emit (start < Q ? 0 : 1) and the pending inverse bits" == round the window up to the next multiple
    // of Q and emit its top two bits (carries resolve any pending run).
    sink.add_at_window(0x4000u);
    sink.advance(2);
    return sink.finish();
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// diff + signed exp-Golomb for one sample (diff.c:15-20, seg.c:11-28).  Produces the codeword as up to three pieces of
// at most 32 bits each: the codeword of w = code_number + 1 is w written in 2p+1 bits, p = floor(log2 w).
// ---------------------------------------------------------------------------------------------------------------------
struct SegWord
{
  uint32_t w_lo; // low 32 bits of w
  uint32_t p;    // prefix length, 0..32
  bool ok;       // false: the difference does not fit 32 bits -> ERROR_INVALID_VALUE (diff.c:17-18)
};

DG_DEV SegWord diff_seg(uint32_t u, uint32_t &last)
{
  SegWord r;
  const uint32_t d = u - last;                       // low 32 bits of (int64)u - (int64)last; u is zero extended (diff.c:15)
  r.ok = (u >= last) == ((int32_t)d >= 0);           // fits int32 <=> sign of the wrapped difference is the true sign
  last = u;
  const int32_t v = (int32_t)d;
  if (v > 0)
  {
    r.w_lo = 2u * (uint32_t)v; // (2v - 1) + 1
    r.p = 31u - clz32(r.w_lo);
  }
  else
  {
    const uint32_t mag = 0u - (uint32_t)v; // |v|, 2^31 for INT32_MIN
    r.w_lo = 2u * mag + 1u;                // 2|v| + 1 ; wraps to 1 for |v| = 2^31 where w = 2^32 + 1
    r.p = (mag == 0x80000000u) ? 32u : 31u - clz32(r.w_lo);
  }
  return r;
}

// Per-lane bit queue feeding the coder: bits are appended MSB first, whole 32-bit words go to the lane's column of an
// LDS ring (slot-major: ring[slot * 64 + lane], so a wave's access is always conflict free).
struct BitQueue
{
  uint64_t acc;
  uint32_t cnt;   // bits in acc, < 32 between puts
  uint32_t wr;    // words written so far
  uint32_t rd;    // words consumed so far

  DG_DEV void init()
  {
    acc = 0;
    cnt = 0;
    wr = 0;
    rd = 0;
  }

  template <uint32_t RING>
  DG_DEV void put(uint32_t v, uint32_t n, uint32_t *ring_col) // n <= 32; ring_col = &ring[lane]
  {
    acc = (acc << n) | v;
    cnt += n;
    if (cnt >= 32)
    {
      ring_col[(wr % RING) * 64u] = (uint32_t)(acc >> (cnt - 32u));
      wr++;
      cnt -= 32;
    }
  }

  template <uint32_t RING>
  DG_DEV void put_codeword(const SegWord &s, uint32_t *ring_col)
  {
    if (s.p <= 15) // 2p+1 <= 31 bits: the whole codeword in one piece (leading zeros are the prefix)
      put<RING>(s.w_lo, 2u * s.p + 1u, ring_col);
    else
    {
      put<RING>(0u, s.p, ring_col); // prefix zeros (seg.c:18)
      if (s.p == 32)                // w = 2^32 + 1: 33 bits
      {
        put<RING>(1u, 1u, ring_col);
        put<RING>(s.w_lo, 32u, ring_col);
      }
      else
        put<RING>(s.w_lo, s.p + 1u, ring_col); // delimiting one + residual (seg.c:19)
    }
  }
};

} // namespace dg
