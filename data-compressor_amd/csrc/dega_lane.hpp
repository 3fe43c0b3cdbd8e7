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
//
// Cost model this code is written against (profiles/r01_ubench_issue_cost.txt, measured on MI355X): a wave issues one
// instruction every ~4 cycles whatever it is (VALU add, 32x32 multiply, 64-bit shift, SALU, LDS), and more waves per
// SIMD do not raise the VALU rate -- so time = instruction count, and a divergent `if` costs its exec-mask SALU
// bookkeeping on every pass.  Hence: unrolled, branch-free 32-symbol word paths with selects, every rare event
// (model halving, MPS/LPS swap, too many finished bits between two hand-overs, slab nearly full) moved to a
// precondition that is looked at once per many words, and a bit-at-a-time slow path that handles everything.
#pragma once

#include "dega_intrinsics.hpp"

namespace dg
{

constexpr int32_t OK = 0;
constexpr int32_t ERR_INVALID_VALUE = -1;
constexpr int32_t ERR_INVALID_FORMAT = -3;
constexpr int32_t ERR_MEMORY = -6;

constexpr uint32_t MAX_FREQUENCY = 16383; // bac.c:27
constexpr uint32_t DIV_TABLE_SIZE = 16384;

// Exact floor(n / t) for 0 <= n < 2^30, 3 <= t <= 16383 as  mulhi(n, magic[t]) >> shift(t)  with
// magic = ceil(2^(30+L) / t), L = ceil(log2 t), shift = L - 2: the error term n*(magic*t - 2^(30+L)) / (t * 2^(30+L)) is
// below 2^-L <= 1/t, so the floor cannot move.  This replaces the two 64-bit divisions per symbol of bac.c:110-111
// (numerators there are range * cum <= 2^16 * 2^13).  The table holds the magics (64 KiB); the shift comes from t.
DG_DEV uint32_t div_shift(uint32_t t)
{
  return 30u - clz32(t - 1u);
}

// ---------------------------------------------------------------------------------------------------------------------
// The encoder -- in two halves that run in two different waves of a pair (dega_kernels.hpp):
//   BacCoder   the serial arithmetic: interval, model, renormalisation.  Nothing else: the coding wave's instruction
//              stream is what bounds the kernel (one wave issues one instruction per 4.1 - 4.6 cycles whatever its
//              partner does: profiles/r03_ubench2_issue_cost.txt), so everything that is not on the chain is gone from it.
//   BacWriter  the output side: finished bits -> 32-bit words -> the channel's slab, in the partner wave.
// Between them: a ring of RAW entries per lane, each the high dword of the coder's register L (below) at a dump.
//
// Interval state (bac.c:83-139) in a form whose renormalisation is branch free:
//   A = start << 16,   B = (65535 - end) << 16   (low 16 bits always zero)
// so that range - 1 = ~(A + B) >> 16 and the number of renormalisation shifts is one count-leading-zeros
// (renorm_shifts).  After an E3 step the reference clears the top bit of both; here it is left set in both ("spurious"
// bit 31): every use either shifts it out or cancels it (A + B mod 2^32, A ^ B).
//
// Output side: carry propagation instead of bit-plus-follow.  The reference emits a bit per E1/E2 shift and defers E3
// shifts in a counter resolved by the next emitted bit (bac.c:93-105,127-132).  The stream it produces is exactly the
// binary expansion of the running sum of the `start` increments, each added at the current position: an E3 shift
// provisionally emits 0 then 1s and a later carry flips them, which is what a multi-word addition does.  So the lane
// keeps ONE 64-bit register
//   L = [ 1 | c | finished bits ] : [ A ]        high dword : low dword
// A is the interval's start as above; every `start` increment is a 64-bit add into L (the carry out of A lands in the
// finished bits by itself) and every renormalisation a 64-bit shift (the bits leaving A become finished bits by
// themselves).  The high dword starts as binary 10: the leading one is a sentinel whose position tells how many bits are
// finished (no counter to keep per symbol), the zero below it takes a carry that runs past ALL finished bits (it can
// take one: a pending run starts with a provisional 0, so a second such carry needs a new run, which stays inside).
// Every few symbols the high dword -- at most 30 finished bits fit -- is DUMPED: written as it is into the lane's next
// raw-ring slot and reset to binary 10.  That is all the coding wave does about output.
//
// The writer absorbs the entries in order:
//   F    = finished bits not yet in whole words, right aligned, fcnt of them (< 32 between entries, + an entry's <= 30,
//          + 1 for the carry: never more than 62, so F cannot overflow)
//   prev = the last completed 32-bit word, held back from memory so that it can still absorb a carry out of F
// and hands whole words on: prev (+ that carry) to the lane's staging column, the new word to prev.  A carry running
// even past prev (33+ pending bits) ripples into the words already stored, which this lane wrote itself.
//
// How many symbols may go between two dumps?  n symbols shift out fewer than 2 + sum(-log2 p_i) bits (the range starts
// and ends the group in (Q, 4Q]), and no symbol is less probable than f2 / tot, so with tot <= 11 * f2 eight symbols
// stay below 2 + 8 * 3.46 < 30 bits, and with tot <= 128 * f2 four do.  These are per-word preconditions (classify());
// a lane outside both codes bit by bit and dumps whenever 15 or more bits are finished (a symbol adds at most 16).
//
// Model (bac.c:39-81, binary case): index 1 = more frequent bit value (`mps`), index 2 the other, index 3 = EOF with
// frequency 1 forever; cum[0] = tot = f1 + f2 + 1, cum[1] = c1 = f2 + 1, cum[2] = 1, cum[3] = 0.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef DG_ENC_CODE_PRIO
#define DG_ENC_CODE_PRIO 3 // issue priority of the coding waves (0 .. 3; their helpers: 0 and 1)
#endif
constexpr uint32_t ENC_RAW = 16;   // raw-ring entries per lane (narrow batches); a word path needs up to 8 free
constexpr uint32_t ENC_ORING = 32; // staged output words per lane on the writer's side (drained in groups of 16)

// classes of the next 32 symbols of a lane, cheapest first (the wave takes the most expensive one any of its lanes needs)
constexpr uint32_t CLS_FAST8 = 0;   // no model event possible, a dump every 8 symbols is enough
constexpr uint32_t CLS_SPLIT = 1;   // as FAST8 but for ONE halving of the counts inside the word: coded in two parts (see below)
constexpr uint32_t CLS_FAST4 = 2;   // no model event possible, skewed counts: a dump every 4 symbols
constexpr uint32_t CLS_GENERAL = 3; // halving / MPS-LPS swap / division-shift change handled by selects, dump every 4
constexpr uint32_t CLS_BITS = 4;    // bit at a time: extreme counts

// The number of renormalisation shifts (E1/E2 + E3, bac.c:112-137) of the interval (a, b) in the representation above.
// U = ~(a ^ b) is zero over the common prefix of start and end (k bits: the E1/E2 shifts), one at the first difference --
// start 0, end 1 -- and one over the E3 run below it (start 1, end 0, i.e. a = b = 1), which is exactly where V = a & b has
// its leading run, one bit lower.  So U ^ (V << 1) is zero over prefix and run and one right after: its leading zeros
// are k + e.  (Whatever V holds further down only reaches bits below that one; the low 16 bits of U are ones, so the
// count is at most 16; spurious top bits: U's bit 31 is one then, k = 0, and V's bit 31 is shifted out.)
DG_DEV uint32_t renorm_shifts(uint32_t a, uint32_t b)
{
  const uint32_t v = a & b;
  return clz32(~(a ^ b) ^ (v + v));
}

template <bool ADAPTIVE, uint32_t RAW = ENC_RAW>
struct BacCoder
{
  static_assert((RAW & (RAW - 1)) == 0 && RAW >= 8 && RAW <= 128, "raw ring: a power of two, room for a word path's 8 entries, counted modulo 256");
  uint64_t L; // low dword: A; high dword: sentinel, carry slot, finished bits (see above)
  uint32_t B;
  uint32_t c1, tot, mps;
  uint32_t *raw;        // lane's column of the raw ring: entry e at raw[(e % RAW) * 64]
  uint32_t rwr;         // entries written so far
  const uint32_t *peer; // the writer's published word of this lane (entries absorbed, modulo 256, in bits 16..23)
  uint32_t safe;        // words that may still be coded in class `cls` before the preconditions have to be looked at again
  uint32_t cls;
  // The symbols of the next word that the next word path codes: all 32 -- or, around a halving of the counts (CLS_SPLIT),
  // first 0 .. h (h = the symbol whose update halves, bac.c:57), then, a step later, h+1 .. 31.  The other symbols of
  // such a step are made no-ops -- a more probable symbol whose division magic is 0 leaves interval, counts and output
  // untouched -- so the word path stays the straight-line fast one.  (The counts halve every 8191 symbols, and the 64
  // lanes of a wave reach that point in different steps: handled by the general path, the wave spent a fifth of its
  // steps there.)
  uint32_t part_lo, part_hi;
  bool part_halves; // the two parts meet at a halving of the counts (else: at a change of the division shift)

  DG_DEV void init(uint32_t *raw_, const uint32_t *peer_)
  {
    L = (uint64_t)2 << 32; // bac.c:86-91: start = 0; no finished bits
    B = 0;
    c1 = 2; // bac.c:39-52
    tot = 3;
    mps = 0;
    raw = raw_;
    rwr = 0;
    peer = peer_;
    safe = 0;
    cls = CLS_BITS;
    part_lo = 0;
    part_hi = 31;
    part_halves = false;
  }

  DG_DEV uint32_t A() const
  {
    return (uint32_t)L;
  }

  // ---- the raw ring ------------------------------------------------------------------------------------------------------

  DG_DEV static bool raw_room(uint32_t written, uint32_t peer_word, uint32_t entries) // room for `entries` more
  {
    return ((written - (peer_word >> 16)) & 0xFFu) + entries <= RAW;
  }

  DG_DEV void dump() // the caller has made sure of room
  {
    raw[(rwr % RAW) * 64u] = (uint32_t)(L >> 32);
    L = ((uint64_t)2 << 32) | (uint32_t)L;
    rwr++;
  }

  // The slow paths (bit at a time, end of stream) make sure themselves: the writer is another wave and never waits for
  // this one, so the wait ends.  (Only lanes of such a path are active here; the others wait at the reconvergence.)
  DG_DEV void dump_when_room()
  {
    if (!raw_room(rwr, peer_load(peer), 1u))
    {
      wave_priority<0>(); // the writer needs the issue slots this wave would poll away
      do
        wave_sleep<2>();
      while (!raw_room(rwr, peer_load(peer), 1u));
      wave_priority<DG_ENC_CODE_PRIO>();
    }
    dump();
  }

  // a bit-at-a-time word is over: flush, and leave the entry count a multiple of 4 (the fast word path's four entries then
  // never wrap inside the ring: one address, four offsets); an entry of binary 10 holds no bits
  DG_DEV void end_bits_word()
  {
    do
      dump_when_room();
    while ((rwr & 3u) != 0u);
  }

  // ---- one symbol, every special case handled in place -----------------------------------------------------------------

  DG_DEV void update_model(bool lps) // bac.c:54-81
  {
    if (tot == MAX_FREQUENCY) // :57-67 halve, rounding up; EOF stays 1
    {
      const uint32_t f1 = (tot - c1 + 1u) >> 1;
      const uint32_t f2 = c1 >> 1; // (f2 + 1) / 2 with f2 = c1 - 1
      c1 = f2 + 1u;
      tot = f1 + f2 + 1u;
    }
    if (lps)
    {
      if (c1 - 1u == tot - c1) // :68-77 f2 == f1: the coded symbol moves to index 1, then f1++
        mps ^= 1u;
      else
        c1++;
    }
    tot++;
  }

  DG_DEV void encode_bit(uint32_t bit, const uint32_t *magic) // EncodeSymbol + UpdateModel (bac.c:156-161)
  {
    const uint32_t M = magic[tot];
    const uint32_t sh = div_shift(tot);
    const uint32_t a = A();
    const uint32_t R = ((~(a + B)) >> 16) + 1u;
    const uint32_t x1 = mulhi32(R * c1, M) >> sh; // range * cum[1] / cum[0]
    const bool lps = bit != mps;
    uint32_t inc = x1; // index 1: end unchanged, start += x1
    if (lps)
    {
      B = 0u - (a + (x1 << 16)); // index 2: end = start + x1 - 1 ...
      inc = mulhi32(R, M) >> sh; // ... start += range * cum[2] / cum[0], cum[2] = 1   (bac.c:110-111)
    }
    L += (uint64_t)(inc << 16);
    const uint32_t n = renorm_shifts(A(), B); // <= 16
    L <<= n;
    B <<= n;
    if ((uint32_t)(L >> 32) >= 0x10000u) // 15 or more bits finished: the next symbol's 16 might not fit
      dump_when_room();
    if (ADAPTIVE)
      update_model(lps);
    safe = 0; // the model moved outside a word path: classify again
  }

  // EOF symbol + FinishEncoding (bac.c:163-164, 141-145): the last entries of the lane
  DG_DEV void finish(const uint32_t *magic)
  {
    const uint32_t a = A();
    const uint32_t R = ((~(a + B)) >> 16) + 1u;
    const uint32_t x2 = mulhi32(R, magic[tot]) >> div_shift(tot); // index 3: cum[2] = 1, cum[3] = 0
    B = 0u - (a + (x2 << 16));                                     // end = start + x2 - 1, start unchanged
    const uint32_t n = renorm_shifts(a, B);
    L <<= n;
    B <<= n;
    dump_when_room();
    // "pending++ ; emit (start < Q ? 0 : 1) and the pending inverse bits" == round the window up to the next multiple
    // of Q and emit its top two bits (the carry resolves any pending run).
    L += (uint64_t)0x40000000u;
    L <<= 2;
    dump_when_room();
  }

  // ---- the class of the next word, and for how many words it holds -------------------------------------------------------
  // Evaluated when `safe` has run out.  All conditions look ahead over whole words, so that in between one decrement
  // per word is all the bookkeeping there is:
  //   no halving          tot + 32 <= MAX_FREQUENCY at the start of each word (bac.c:57)
  //   no MPS/LPS swap     f1 - f2 >= 32 at the start of each word (bac.c:68: needs f2 == f1; a word adds at most 32 to f2)
  //   one division shift  tot - 1 and tot + 30 have the same bit length
  //   dump capacity       tot <= 11 * f2 (eight symbols) or tot <= 128 * f2 (four), up to the end of each word
  DG_DEV void classify()
  {
    if (!ADAPTIVE)
    {
      cls = CLS_FAST8; // counts 1 : 1 : 1 forever: 8 symbols shift out fewer than 2 + 8 * log2(3) < 15 bits
      safe = 0x40000000u;
      return;
    }
    const uint32_t f2 = c1 - 1u, diff = tot + 1u - 2u * c1; // f1 - f2
    const uint32_t w_halve = (MAX_FREQUENCY - tot) >> 5;
    const uint32_t w_swap = diff >> 5;
    const uint32_t pw = 0x80000000u >> (clz32(tot - 1u) - 1u); // the power of two above tot - 1
    const uint32_t w_shift = pw >= tot + 31u ? (pw - tot - 30u + 31u) >> 5 : 0u;
    uint32_t w = w_halve < w_swap ? w_halve : w_swap;
    w = w < w_shift ? w : w_shift;
    const uint32_t lim8 = 11u * f2, lim4 = 128u * f2;
    const uint32_t w8 = lim8 >= tot + 31u ? ((lim8 - tot - 31u) >> 5) + 1u : 0u;
    const uint32_t w4 = lim4 >= tot + 31u ? ((lim4 - tot - 31u) >> 5) + 1u : 0u;
    if (w != 0u && w8 != 0u)
    {
      cls = CLS_FAST8;
      safe = w < w8 ? w : w8;
    }
    else if (w != 0u && w4 != 0u)
    {
      cls = CLS_FAST4;
      safe = w < w4 ? w : w4;
    }
    else if (w_halve == 0u && tot <= MAX_FREQUENCY && diff >= 96u && lim8 >= tot + 160u)
    {
      // nothing but a halving in this word, with room to spare on both sides of it: the counts are in the top half of
      // their range (one division shift before, and after -- cum[0] restarts at 8193 or 8194, never at a power of two),
      // f1 - f2 >= 96 leaves >= 47 after the halving (no swap within 32 symbols), and 11 * f2 >= tot + 160 keeps the
      // eight-symbol dump capacity for the halved counts (f2 -> at least f2 / 2, tot -> at most tot / 2 + 2)
      cls = CLS_SPLIT;
      safe = 0;
      part_lo = 0;
      part_hi = MAX_FREQUENCY - tot; // 0..31: the update of this symbol halves
      part_halves = true;
    }
    else if (w_halve != 0u && w_swap != 0u && w_shift == 0u && w8 != 0u)
    {
      // nothing but a change of the division shift in this word -- cum[0] passes a power of two, which it does ten times
      // on its way up (short channels never get further) -- : the symbols up to cum[0] = 2^k and those after it are two
      // plain fast parts, each with its own shift; the counts just go on counting
      cls = CLS_SPLIT;
      safe = 0;
      part_lo = 0;
      part_hi = pw - tot; // 0..30: the last symbol coded with the old shift
      part_halves = false;
    }
    else
    {
      // the general word path copes with every model event; its groups of four need tot <= 128 * f2 throughout, and a
      // halving keeps the ratio within rounding (f2 -> (f2 + 1) / 2, tot -> about tot / 2): 100 leaves room for that
      cls = (tot + 32u <= 100u * f2) ? CLS_GENERAL : CLS_BITS;
      safe = 0;
    }
  }

  // A word path is about to code the whole word after all (the wave took the general path for another lane's sake)
  DG_DEV void whole_word()
  {
    part_lo = 0;
    part_hi = 31;
  }

  // After a masked word path (encode_word<false, 8, true>) of a CLS_SPLIT lane.  Returns true when the word is complete.
  // First part done: the update of symbol part_hi, which the word path applied as a plain count, is redone as
  // UpdateModel does it (bac.c:57-80): halve, then count the symbol, then cum[0]++.
  DG_DEV bool after_part(uint32_t word)
  {
    if (part_lo == 0u)
    {
      const uint32_t h = part_hi;
      if (part_halves)
      {
        const uint32_t lps = ((word >> (31u - h)) & 1u) ^ mps;
        const uint32_t c1h = c1 - lps;                     // cum[1] when symbol h was coded; cum[0] was MAX_FREQUENCY
        c1 = (c1h >> 1) + 1u + lps;                        // :60-66 (f2 + 1) / 2 + 1, then :78 (no tie: see classify)
        tot = ((MAX_FREQUENCY + 1u - c1h) >> 1) + (c1h >> 1) + 2u; // (f1 + 1) / 2 + (f2 + 1) / 2 + 1, then :80
      }
      part_lo = h + 1u;
      part_hi = 31;
      if (h != 31u)
        return false;
    }
    part_lo = 0;
    return true;
  }

  // The first quarter of the division magics of the word that a lane would code next with a fast word path
  // (cum[0] = tot .. tot+7; all a static model ever needs).  The kernel fetches them at the top of its step, together
  // with the queued word; the other quarters follow inside the word path (see encode_word).
  DG_DEV void fetch_magics_first(const uint32_t *magic, uint32_t (&Mg)[32]) const
  {
    const uint32_t *const mg = magic + (tot - part_lo); // symbol i is coded with cum[0] = tot + (i - part_lo)
#pragma unroll
    for (uint32_t i = 0; i < (ADAPTIVE ? 8u : 1u); i++)
#if defined(DEGA_DIAG) && (DEGA_DIAG & 4)
      Mg[i] = 0x40000000u + tot * 131u + i; // diagnostic build: no table reads
#else
      Mg[i] = mg[i];
#endif
  }

  // ---- word paths: 32 symbols, branch free ---------------------------------------------------------------------------------
  // GENERAL = false: no model event can occur in the word (classes FAST8 / FAST4): the magics come in registers, the
  //                  counts change only by c1 -= lps.
  // GENERAL = true : the whole model update of bac.c:54-81 by selects; magics read from the table as the counts move.
  // DUMP: symbols between two dumps (8 or 4): 32 / DUMP raw entries, for which the caller has made sure of room.
  // MASKED: only the symbols part_lo .. part_hi are coded, the others are no-ops (CLS_SPLIT; fast path only).
  template <bool GENERAL, uint32_t DUMP, bool MASKED = false>
  DG_DEV void encode_word(uint32_t word, const uint32_t *magic, uint32_t (&Mg)[32])
  {
    static_assert(!GENERAL || ADAPTIVE, "the static model never needs the general path");
    static_assert(!MASKED || (!GENERAL && ADAPTIVE), "parts of words are a matter of the adaptive fast path");
    uint32_t mm = 0u - mps;                          // all ones when the MPS is the bit value 1
    // symbol i <-> bit 31 - i; a symbol that is not coded becomes a more probable one with magic 0
    const uint32_t active = MASKED ? (0xFFFFFFFFu >> part_lo) & (0xFFFFFFFFu << (31u - part_hi)) : 0xFFFFFFFFu;
    const uint32_t lw = GENERAL ? word : ((word ^ mm) & active); // fast: bit set = less probable symbol (the MPS cannot change)
    const uint32_t sh_word = div_shift(tot);
    const uint32_t tot_word = tot - (MASKED ? part_lo : 0u);
    uint32_t Mcur = GENERAL ? magic[tot] : 0u;
    // the word's raw entries: with four of them (and the entry count a multiple of 4: end_bits_word) one address will do
    uint32_t *const slot = raw + (rwr % RAW) * 64u;
    // The 32 division magics of a fast word are scattered, bank-conflicting LDS reads: all at once they take a few hundred
    // cycles to land, and whoever needs an LDS answer meanwhile waits for the lot (LDS answers in order; the compiler
    // waits for "all outstanding").  So they come in quarters: Mg[0..7] are fetched by the caller a little ahead of the
    // word (fetch_magics_first), each further quarter here, eight symbols before its first use -- landed by then, and
    // never more than four reads in the queue.
#if defined(DEGA_DIAG) && (DEGA_DIAG & 8)
    constexpr uint32_t NSYM = 16; // diagnostic build: half the symbols per word
#else
    constexpr uint32_t NSYM = 32;
#endif
#pragma unroll
    for (uint32_t i = 0; i < NSYM; i++)
    {
      uint32_t M, sh, c1u = c1, totu = tot, Mnext = 0;
      if (GENERAL)
      {
        // the part of the model update that does not depend on this symbol: halving, next cum[0]; fetch its magic now
        const bool halve = tot == MAX_FREQUENCY;
        const uint32_t c1h = (c1 >> 1) + 1u;                // (f2 + 1) / 2 + 1
        const uint32_t toth = ((tot - c1 + 1u) >> 1) + c1h; // (f1 + 1) / 2 + (f2 + 1) / 2 + 1
        c1u = halve ? c1h : c1;                             // counts as UpdateModel sees them after :57-67
        totu = halve ? toth : tot;
        Mnext = magic[totu + 1u];
        M = Mcur;
        sh = div_shift(tot);
      }
      else
      {
        M = ADAPTIVE ? Mg[i] : Mg[0];
        if (MASKED)
          M &= (uint32_t)((int32_t)(active << i) >> 31);
        sh = sh_word;
      }
      const uint32_t a = (uint32_t)L;
      uint32_t lm = (uint32_t)((int32_t)(lw << i) >> 31); // all ones for a set bit
      if (GENERAL)
        lm ^= mm;                                         // ... for an LPS
      const uint32_t R = range_from_sum(a + B);               // 1 .. 65536
      const uint32_t x1 = mulhi32(mul24(R, c1), M) >> sh;     // range * cum[1] / cum[0]   (R * c1 < 2^31)
      const uint32_t x2 = mulhi32(R, M) >> sh;                // range * cum[2] / cum[0], cum[2] = 1
      const uint32_t inc = select32(lm, x2, x1);
      B = select32(lm, 0u - (a + (x1 << 16)), B);
      L = add_shifted16(L, inc); // the carry out of A lands in the finished bits
      // renormalise: A and the finished bits move together
      const uint32_t n = renorm_shifts((uint32_t)L, B);
      L <<= n;
      B <<= n;
      if (GENERAL)
      {
        // the symbol-dependent part of the update (:68-80): an LPS either swaps roles (f2 == f1) or counts up
        const uint32_t tie = (c1u - 1u == totu - c1u) ? 0xFFFFFFFFu : 0u;
        mm ^= lm & tie;
        c1 = c1u - (lm & ~tie);
        tot = totu + 1u;
        Mcur = Mnext;
      }
      else if (ADAPTIVE)
        c1 -= lm; // f2++ for an LPS; tot is implicit (Mg[i])
      if ((i % DUMP) == DUMP - 1)
      {
        // dump: the finished bits leave as they are
        if (DUMP == 8)
          slot[(i / DUMP) * 64u] = (uint32_t)(L >> 32);
        else
          raw[((rwr + i / DUMP) % RAW) * 64u] = (uint32_t)(L >> 32);
        L = ((uint64_t)2 << 32) | (uint32_t)L;
      }
      if (!GENERAL && ADAPTIVE && (i == 0u || i == 8u || i == 16u))
      {
        DG_COMPILER_BARRIER(); // keeps the quarters where they are (the scheduler would gather them at the top)
        const uint32_t *const mq = magic + tot_word + i + 8u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++)
#if defined(DEGA_DIAG) && (DEGA_DIAG & 4)
          Mg[i + 8u + k] = 0x40000000u + tot_word * 131u + i + k;
#else
          Mg[i + 8u + k] = mq[k];
#endif
        DG_COMPILER_BARRIER();
      }
    }
    rwr += NSYM / DUMP;
    if (GENERAL)
      mps = mm & 1u;
    else if (ADAPTIVE)
      tot += MASKED ? part_hi - part_lo + 1u : 32u;
  }
};

// The output side of a lane: raw entries in, 32-bit words out (see above).  Runs in the coder's partner wave.
template <uint32_t ORING = ENC_ORING>
struct BacWriter
{
  uint64_t F;
  uint32_t fcnt;
  uint32_t prev, pos;       // pos = words completed so far; prev is word pos-1 (held back)
  uint32_t drained, staged; // words [0, drained) are in the slab, [drained, drained + staged) in the lane's LDS column
  uint32_t cap_words;
  uint32_t *dst;   // channel's slab (global memory)
  uint32_t *oring; // lane's column of the LDS staging ring: word w at oring[(w % ORING) * 64]
  int32_t err;

  DG_DEV void init(uint32_t *dst_, uint32_t cap_words_, uint32_t *oring_)
  {
    F = 0;
    fcnt = 0;
    prev = 0;
    pos = 0;
    drained = 0;
    staged = 0;
    cap_words = cap_words_;
    dst = dst_;
    oring = oring_;
    err = OK;
  }

  DG_DEV void put_word(uint32_t index, uint32_t word)
  {
    if (index < cap_words)
      dst[index] = bswap32(word); // MSB-first bit order => big-endian words
    else if (err == OK)
      err = ERR_MEMORY;
  }

  DG_DEV void drain_lane() // LDS column -> slab, this lane only (the kernel normally drains whole waves in lockstep)
  {
    for (uint32_t s = 0; s < staged; s++)
      put_word(drained + s, oring[((drained + s) % ORING) * 64u]);
    drained += staged;
    staged = 0;
  }

  // Four staged words as one 16-byte store: scattered stores cost per instruction, not per byte.
  DG_DEV void put_group(uint32_t index, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
  {
    if (index + 4u <= cap_words)
    {
      store_x4(dst + index, bswap32(w0), bswap32(w1), bswap32(w2), bswap32(w3));
    }
    else
    {
      put_word(index, w0);
      put_word(index + 1u, w1);
      put_word(index + 2u, w2);
      put_word(index + 3u, w3);
    }
  }

  DG_DEV void push_word(uint32_t word)
  {
    if (staged == ORING)
      drain_lane();
    oring[((drained + staged) % ORING) * 64u] = word;
    staged++;
  }

  // add one to the big number formed by words [0, count): the staged ones first, then the slab (rare: a carry past `prev`)
  DG_DEV void ripple_carry_from(uint32_t count)
  {
    drain_lane();
    while (count > 0)
    {
      --count;
      if (count < cap_words)
      {
        const uint32_t w = bswap32(dst[count]) + 1u;
        dst[count] = bswap32(w);
        if (w != 0)
          break;
      }
    }
  }

  // One whole word out of F if it holds one (fcnt >= 32): prev, plus the carry that came up through F, goes to the
  // staging column; the new word is held back.
  DG_DEV void hand_off()
  {
    if (fcnt < 32u)
      return;
    const uint32_t sh = fcnt - 32u;
    const uint64_t top = F >> sh; // the word, above it the carry (0 or 1)
    const uint32_t carry = (uint32_t)(top >> 32);
    if (pos > 0u)
    {
      const uint32_t sum = prev + carry;
      if (sum < carry) // prev was all ones: the carry runs on into words already handed on
        ripple_carry_from(pos - 1u);
      push_word(sum);
    }
    pos++;
    prev = (uint32_t)top;
    F = (uint32_t)F & ((1u << sh) - 1u);
    fcnt = sh;
  }

  // One raw entry of the coder: its finished bits (and the carry above them) join F.  Exact for any entry with at
  // most 30 finished bits; an entry of binary 10 changes nothing.
  DG_DEV void absorb(uint32_t hi)
  {
    const uint32_t fb = 30u - clz32(hi);          // the sentinel sits at bit fb + 1
    const uint32_t v = hi - (2u << fb);           // finished bits, the carry (if any) at bit fb
    F = (F << fb) + v;                            // the carry adds into the bits F already holds
    fcnt += fb;
    hand_off();
  }

  // The four entries of a fast word step at once, when F can take them all -- fewer than 64 bits with what it holds, so at
  // most one word completes: the usual case by far (a word step finishes 28 bits on this data, a word at most)
  DG_DEV bool fits4(const uint32_t (&hi)[4]) const
  {
    return fcnt + (120u - clz32(hi[0]) - clz32(hi[1]) - clz32(hi[2]) - clz32(hi[3])) < 64u;
  }

  DG_DEV void absorb4(const uint32_t (&hi)[4])
  {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
    {
      const uint32_t fb = 30u - clz32(hi[k]);
      F = (F << fb) + (hi[k] - (2u << fb));
      fcnt += fb;
    }
    hand_off();
  }

  // The same for a whole wave in step, branch free: fits4(), a word held back already (pos >= 1) and room in the staging
  // column are the caller's business.  Returns true when a carry ran past the held-back word (it was all ones: once in
  // 2^32 hand-overs of random data): the caller then calls ripple_carry_from(pos - 2) -- the word just staged is word
  // pos - 2, the carry belongs in the words in front of it.
  DG_DEV bool absorb4_in_step(const uint32_t (&hi)[4])
  {
#pragma unroll
    for (uint32_t k = 0; k < 4; k++)
    {
      const uint32_t fb = 30u - clz32(hi[k]);
      F = (F << fb) + (hi[k] ^ (2u << fb)); // (the sentinel bit is set: xor takes it out)
      fcnt += fb;
    }
    const uint32_t m = (uint32_t)((int32_t)(31u - fcnt) >> 31); // all ones when a word is complete
    const uint32_t sh = (fcnt - 32u) & 31u;
    const uint64_t top = F >> sh;                               // the word, above it the carry; garbage when m == 0
    const uint32_t carry = (uint32_t)(top >> 32) & m;
    const uint32_t sum = prev + carry;
    oring[((drained + staged) % ORING) * 64u] = sum; // harmless when no word is complete: the slot is rewritten by the next one
    staged += m & 1u;
    pos += m & 1u;
    prev = select32(m, (uint32_t)top, prev);
    const uint32_t keep = (uint32_t)F & ((1u << sh) - 1u);
    F = m ? (uint64_t)keep : F;
    fcnt = select32(m, sh, fcnt);
    return sum < carry;
  }

  // What is left when the coder is through: prev, then the fcnt < 32 bits of F, zero padded (bit_file_buffer.c:310-320).
  // Returns the exact stream length in bits.
  DG_DEV uint64_t finish()
  {
    const uint32_t carry = (uint32_t)(F >> fcnt) & 1u;
    if (pos > 0)
    {
      const uint32_t sum = prev + carry;
      if (sum < carry)
        ripple_carry_from(pos - 1u);
      push_word(sum);
    }
    drain_lane();
    if (fcnt > 0)
      put_word(pos, (uint32_t)F << (32u - fcnt));
    return (uint64_t)pos * 32u + fcnt;
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

// valuesize < 32 (NARROW): u and last are below 2^valuesize, the difference is exact in 32 bits and must lie in
// [-half, half - 1] with half = 2^(valuesize-1) (diff.c:17-18); the codeword is the same function of the difference.
template <bool NARROW = false>
DG_DEV SegWord diff_seg(uint32_t u, uint32_t &last, uint32_t half = 0x80000000u)
{
  SegWord r;
  const uint32_t d = u - last;                       // low 32 bits of (int64)u - (int64)last; u is zero extended (diff.c:15)
  if (NARROW)
    r.ok = d + half < 2u * half;
  else
    r.ok = (u >= last) == ((int32_t)d >= 0);         // fits int32 <=> sign of the wrapped difference is the true sign
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

// diff + seg for the common case, branch free: returns w (the codeword's value; its length is 2*floor(log2 w) + 1 bits).
// w = zigzag(-v) + 1 with v = x - last:  v > 0 -> 2v,  v <= 0 -> 2|v| + 1  (seg.c:25-28 + the "+1" of seg.c:13).
// Only valid when the result fits 16 bits (codeword <= 31 bits); `wide` says when it does not (or v = INT32_MIN).
template <bool NARROW = false>
DG_DEV uint32_t diff_seg_short(uint32_t u, uint32_t &last, bool &ok, bool &wide, uint32_t half = 0x80000000u)
{
  const uint32_t nv = last - u;                                    // -v, low 32 bits
  if (NARROW)
    ok = half - nv < 2u * half;                                    // -half <= v <= half - 1
  else
    ok = (u >= last) == ((int32_t)nv <= 0);                        // diff.c:17-18: the true difference fits int32
  last = u;
  const uint32_t w = ((nv << 1) ^ (uint32_t)((int32_t)nv >> 31)) + 1u;
  wide = (w >> 16) != 0u || nv == 0x80000000u;
  return w;
}

// The steady state of a 32-bit channel, as few instructions as it takes: w for samples whose difference is known to fit
// (both below 2^31 -- the caller ORs the samples of a batch and looks at the top bit once) -- and is short (the caller ORs
// the w of a batch and looks above bit 15 once; v = INT32_MIN cannot occur between samples below 2^31).
DG_DEV uint32_t diff_seg_steady(uint32_t u, uint32_t &last)
{
  const uint32_t nv = last - u; // -v, low 32 bits
  last = u;
  return ((nv << 1) ^ (uint32_t)((int32_t)nv >> 31)) + 1u;
}

// ---- valuesize 33..64: samples in 64-bit containers -------------------------------------------------------------------
// Same functions of the difference, 64 bits wide, in the reference's own 64-bit arithmetic: for valuesize 64 the range
// check is skipped (diff.c:17), the difference wraps, and for |v| = 2^63 so does the code number (-2v = 0 in io_uint_t,
// seg.c:25-28): that difference is coded like 0 -- a loss the reference has and this reproduces.  w < 2^64, codeword
// 2p + 1 <= 127 bits.
struct SegWord64
{
  uint64_t w; // code_number + 1
  uint32_t p; // prefix length, 0..63
  bool ok;
};

DG_DEV SegWord64 diff_seg64(uint64_t u, uint64_t &last, uint32_t valuesize)
{
  SegWord64 r;
  const uint64_t d = u - last; // u, last < 2^valuesize, zero extended (diff.c:15)
  const uint64_t half = 1ull << (valuesize - 1u);
  r.ok = valuesize >= 64u || d + half < 2ull * half; // valuesize <= 63 where checked: 2 * half does not wrap
  last = u;
  const int64_t v = (int64_t)d;
  r.w = v > 0 ? 2ull * (uint64_t)v : 2ull * (0ull - (uint64_t)v) + 1ull; // (2v - 1) + 1  |  -2v + 1, mod 2^64
  r.p = 63u - (uint32_t)__builtin_clzll(r.w);                            // w >= 1
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

  // a codeword of value w < 2^16: 2p+1 <= 31 bits, one piece (its leading zeros are the prefix, seg.c:18-19)
  // Branch free (a divergent `if` would cost its exec-mask bookkeeping on every row): the word slot is written
  // unconditionally -- with garbage while the accumulator holds fewer than 32 bits, overwritten by the real word later.
  template <uint32_t RING>
  DG_DEV void put_short(uint32_t w, uint32_t *ring_col)
  {
    const uint32_t n = 63u - 2u * clz32(w);
    acc = (acc << n) | w;
    cnt += n; // < 64
    ring_col[(wr % RING) * 64u] = (uint32_t)(acc >> ((cnt - 32u) & 63u));
    wr += cnt >> 5; // a word is complete when cnt >= 32 ...
    cnt &= 31u;     // ... and leaves cnt - 32 bits behind
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

  template <uint32_t RING>
  DG_DEV void put_codeword64(const SegWord64 &s, uint32_t *ring_col) // p zeros, then w in p + 1 bits, in pieces of <= 32
  {
    uint32_t z = s.p;
    while (z > 0)
    {
      const uint32_t k = z < 32u ? z : 32u;
      put<RING>(0u, k, ring_col);
      z -= k;
    }
    uint32_t n = s.p + 1u; // bits of w to write, most significant first (w < 2^n)
    if (n > 32u)
    {
      put<RING>((uint32_t)(s.w >> 32), n - 32u, ring_col);
      n = 32u;
    }
    put<RING>((uint32_t)s.w, n, ring_col);
  }
};

} // namespace dg

// =====================================================================================================================
// Decoder side
// =====================================================================================================================
namespace dg
{

constexpr int32_t ERR_LIBRARY_CALL = -11;

// A lane's view of its compressed stream: 32-bit words staged in the lane's column of an LDS ring (slot-major,
// ring[(k % IRING) * 64 + lane]).  The words are "cooked" when they are staged (StreamTail::cook): host byte order, the
// bits at or beyond the stream's exact length cleared, and every word beyond the last one reads as zero -- the reference
// tolerates up to 14 such phantom bits after the end of the stream (bac.c:171-186,192).
template <uint32_t IRING>
struct StreamWindow
{
  const uint32_t *ring_col; // &ring[lane]

  DG_DEV uint32_t word(uint32_t k) const
  {
    return ring_col[(k % IRING) * 64u];
  }

  // 32 stream bits starting at bit position pos (words k = pos/32 and k+1 must be staged)
  DG_DEV uint32_t peek32(uint64_t pos) const
  {
    const uint32_t k = (uint32_t)(pos >> 5), o = (uint32_t)pos & 31u;
    const uint32_t w0 = word(k), w1 = word(k + 1u);
    return o ? (w0 << o) | (w1 >> (32u - o)) : w0;
  }
};

// Where a lane's stream ends, in the form the staging step needs: word k of the slab is part of the stream iff
// k < words; the last of them keeps only its leading bits.
struct StreamTail
{
  uint32_t words;     // min(ceil(nbits / 32), slab words)
  uint32_t last_mask; // of word words - 1

  DG_DEV void init(uint64_t nbits, uint32_t cap_words)
  {
    const uint64_t all = (nbits + 31u) / 32u;
    const uint32_t part = (uint32_t)nbits & 31u;
    words = all < cap_words ? (uint32_t)all : cap_words;
    last_mask = (all <= cap_words && part != 0u) ? ~(0xFFFFFFFFu >> part) : 0xFFFFFFFFu;
  }

  DG_DEV uint32_t cook(uint32_t raw_word, uint32_t k) const // raw_word: the slab's word k as loaded (big-endian)
  {
    const uint32_t w = bswap32(raw_word) & (k + 1u == words ? last_mask : 0xFFFFFFFFu);
    return k < words ? w : 0u;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Arithmetic decoder (bac.c:168-263).  Same interval representation as the encoder; instead of `value` the lane keeps
//   D = value - start   (0 <= D < range)
// because every renormalisation case -- E1, E2 and E3 -- then becomes the same operation: D = (D << n) | next n bits.
// The symbol is found without the division of bac.c:209: index 1 iff D >= x1, EOF iff D < x2, with
// x1 = range*cum[1]/cum[0] and x2 = range*cum[2]/cum[0] -- the very quantities the interval update needs anyway
// (integer identity: v - s >= floor(R*c/t)  <=>  (v - s + 1) * t > R * c).
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t DCLS_FAST = 0, DCLS_SPLIT = 1, DCLS_GENERAL = 2; // cheapest first, as the encoder's CLS_*

template <bool ADAPTIVE>
struct BacDecoder
{
  uint32_t A, B, D;
  uint32_t c1, tot, mps;
  uint64_t bp; // stream bits consumed so far
  // The symbols of the next word that the next masked word path decodes (as BacCoder::part_lo / part_hi): all 32, or --
  // around a halving of the counts -- first 0 .. h, then, a step later, h+1 .. 31; part_bits keeps the first part's bits.
  uint32_t part_lo, part_hi, part_bits;
  bool part_halves; // the first part ends at a halving of the counts (else: at a change of the division shift, or nowhere)
  uint32_t cls, safe; // the class of the next word (DCLS_*) and for how many words it still holds (see classify)

  DG_DEV void init()
  {
    cls = DCLS_GENERAL;
    safe = 0;
    part_halves = false;
    A = 0;
    B = 0;
    D = 0;
    c1 = 2;
    tot = 3;
    mps = 0;
    bp = 0;
    part_lo = 0;
    part_hi = 31;
    part_bits = 0;
  }

  template <uint32_t IRING>
  DG_DEV void start(const StreamWindow<IRING> &in) // StartDecoding, bac.c:188-204: the first 16 bits
  {
    D = in.peek32(0) >> 16;
    bp = 16;
  }

  DG_DEV uint32_t renormalise()
  {
    const uint32_t n = renorm_shifts(A, B);
    A <<= n;
    B <<= n;
    return n;
  }

  DG_DEV void update_model(bool lps) // as BacCoder::update_model (bac.c:54-81)
  {
    if (tot == MAX_FREQUENCY)
    {
      const uint32_t f1 = (tot - c1 + 1u) >> 1;
      const uint32_t f2 = c1 >> 1;
      c1 = f2 + 1u;
      tot = f1 + f2 + 1u;
    }
    if (lps)
    {
      if (c1 - 1u == tot - c1)
        mps ^= 1u;
      else
        c1++;
    }
    tot++;
  }

  // One symbol, everything in place (DecodeSymbol + UpdateModel, bac.c:206-242,253-261).
  // Returns 0/1 = the decoded bit, 2 = EOF symbol.
  template <uint32_t IRING>
  DG_DEV uint32_t decode_bit(const StreamWindow<IRING> &in, const uint32_t *magic)
  {
    const uint32_t M = magic[tot];
    const uint32_t sh = div_shift(tot);
    const uint32_t R = ((~(A + B)) >> 16) + 1u;
    const uint32_t x1 = mulhi32(R * c1, M) >> sh;
    const uint32_t x2 = mulhi32(R, M) >> sh;
    uint32_t result;
    bool lps = false;
    if (D >= x1) // index 1
    {
      A += x1 << 16;
      D -= x1;
      result = mps;
    }
    else
    {
      B = 0u - (A + (x1 << 16)); // end = start + x1 - 1 for index 2 ...
      if (D >= x2)
      {
        A += x2 << 16;
        D -= x2;
        result = mps ^ 1u;
        lps = true;
      }
      else
      {
        B = 0u - (A + (x2 << 16)); // ... and start + x2 - 1 for EOF (index 3), start unchanged
        result = 2;
      }
    }
    const uint32_t n = renormalise();
    if (n > 0)
    {
      D = (D << n) | (in.peek32(bp) >> (32u - n));
      bp += n;
    }
    if (ADAPTIVE && result != 2)
      update_model(lps);
    return result;
  }

  DG_DEV bool fast_ok() const
  {
    bool ok = part_lo == 0u;
    if (ADAPTIVE)
    {
      ok = ok && tot + 32u <= MAX_FREQUENCY;
      ok = ok && tot + 1u >= 2u * c1 + 32u;
      ok = ok && clz32(tot - 1u) == clz32(tot + 30u);
    }
    return ok;
  }

  // The next word holds a halving of the counts (bac.c:57) and nothing else the fast path cannot do -- or its first part
  // has been decoded already: the masked word path takes it, in two steps (see BacCoder::classify, CLS_SPLIT: the
  // counts are in the top half of their range, and f1 - f2 >= 96 rules a swap out on both sides of the halving).
  DG_DEV bool halving_ahead() const // the only event of the next word is a halving, with room to spare (see above)
  {
    return tot + 32u > MAX_FREQUENCY && tot <= MAX_FREQUENCY && tot + 1u >= 2u * c1 + 96u;
  }

  DG_DEV bool shift_change_ahead() const // the only event of the next word is cum[0] passing a power of two
  {
    return tot + 32u <= MAX_FREQUENCY && tot + 1u >= 2u * c1 + 32u && clz32(tot - 1u) != clz32(tot + 30u);
  }

  DG_DEV bool split_ok() const
  {
    return ADAPTIVE && (part_lo != 0u || halving_ahead() || shift_change_ahead());
  }

  // The class of the next word and for how many words it holds, as BacCoder::classify (no dump capacity to mind here):
  // evaluated when `safe` has run out, so that the steady state of the kernel is one decrement per word.
  DG_DEV void classify()
  {
    if (!ADAPTIVE)
    {
      cls = DCLS_FAST;
      safe = 0x40000000u;
      return;
    }
    safe = 0;
    if (part_lo != 0u)
    {
      cls = DCLS_SPLIT; // the second part of a word
      return;
    }
    const uint32_t diff = tot + 1u - 2u * c1; // f1 - f2
    const uint32_t w_halve = (MAX_FREQUENCY - tot) >> 5;
    const uint32_t w_swap = diff >> 5;
    const uint32_t pw = 0x80000000u >> (clz32(tot - 1u) - 1u); // the power of two above tot - 1
    const uint32_t w_shift = pw >= tot + 31u ? (pw - tot + 1u) >> 5 : 0u;
    uint32_t w = w_halve < w_swap ? w_halve : w_swap;
    w = w < w_shift ? w : w_shift;
    if (w != 0u)
    {
      cls = DCLS_FAST;
      safe = w;
    }
    else
      cls = (halving_ahead() || shift_change_ahead()) ? DCLS_SPLIT : DCLS_GENERAL;
  }

  DG_DEV void begin_word() // before a masked word path: which symbols it decodes
  {
    if (part_lo == 0u)
    {
      part_halves = halving_ahead();
      const uint32_t pw = 0x80000000u >> (clz32(tot - 1u) - 1u); // the power of two cum[0] reaches next
      part_hi = part_halves ? MAX_FREQUENCY - tot : (shift_change_ahead() ? pw - tot : 31u);
    }
  }

  DG_DEV void whole_word()
  {
    part_lo = 0;
    part_hi = 31;
    part_bits = 0;
  }

  // After a masked word path that came through; `bits` = its symbols (the others zero).  Returns true when the word is
  // complete (then `bits` is the whole word).  After the first part of a halving word the update of symbol part_hi, which
  // the word path applied as a plain count, is redone as UpdateModel does it: halve, count the symbol, cum[0]++
  // (BacCoder::after_part); at a change of the division shift the counts just go on.
  DG_DEV bool after_part(uint32_t &bits)
  {
    bits |= part_bits;
    if (part_lo == 0u && part_halves)
    {
      const uint32_t h = part_hi;
      const uint32_t lps = ((bits >> (31u - h)) & 1u) ^ mps;
      const uint32_t c1h = c1 - lps;
      c1 = (c1h >> 1) + 1u + lps;
      tot = ((MAX_FREQUENCY + 1u - c1h) >> 1) + (c1h >> 1) + 2u;
    }
    if (part_lo == 0u && part_hi != 31u)
    {
      part_lo = part_hi + 1u;
      part_hi = 31;
      part_bits = bits;
      return false;
    }
    part_lo = 0;
    part_hi = 31;
    part_bits = 0;
    return true;
  }

  // 32 symbols, branch free.  Needs the words bp/32 .. bp/32 + 3 staged.  Returns false -- the caller restores its
  // checkpoint and goes bit by bit -- if an EOF symbol turned up, or if one group of 4 symbols consumed more than 32
  // stream bits (the 32-bit look-ahead is rebuilt every 4 symbols), or if the word needed more than the 4 staged words.
  // GENERAL = false: no model event (halving, MPS/LPS swap, division-shift change) can occur in the word (fast_ok()).
  // GENERAL = true : the whole model update of bac.c:54-81 by selects, as in BacCoder::encode_word_general.
  // as BacCoder::fetch_magics: the kernel issues these reads a phase early
  DG_DEV void fetch_magics_first(const uint32_t *magic, uint32_t (&Mg)[32]) const // as BacCoder::fetch_magics_first
  {
    const uint32_t *const mg = magic + (tot - part_lo);
#pragma unroll
    for (uint32_t i = 0; i < (ADAPTIVE ? 8u : 1u); i++)
      Mg[i] = mg[i];
  }

  // pre[0..3]: the stream's words bp/32 .. bp/32 + 3 (StreamWindow::word) -- the kernel reads them ahead of its ballots
  // MASKED: only the symbols part_lo .. part_hi are decoded, the others are no-ops (magic 0: a more probable symbol that
  // changes nothing); bits_out holds zeros for them.
  template <bool GENERAL, bool MASKED = false>
  DG_DEV bool decode_word(const uint32_t *magic, uint32_t (&Mg)[32], const uint32_t (&pre)[4], uint32_t &bits_out)
  {
    static_assert(!MASKED || (!GENERAL && ADAPTIVE), "parts of words are a matter of the adaptive fast path");
    const uint32_t active = MASKED ? (0xFFFFFFFFu >> part_lo) & (0xFFFFFFFFu << (31u - part_hi)) : 0xFFFFFFFFu;
    const uint32_t sh_fast = div_shift(tot);
    uint32_t Mcur = GENERAL ? magic[tot] : 0u;
    const uint32_t k0 = (uint32_t)(bp >> 5);
    const uint32_t w0 = pre[0], w1 = pre[1], w2 = pre[2], w3 = pre[3];
    const uint32_t tot_word = tot - (MASKED ? part_lo : 0u);
    uint32_t off = (uint32_t)bp & 31u; // bit offset into w0:w1:w2:w3
    uint32_t off_group = off;
    uint32_t ahead = off ? (w0 << off) | (w1 >> (32u - off)) : w0; // next 32 stream bits, left aligned
    uint32_t mm = 0u - mps;
    uint32_t out = 0, eof = 0, bad = 0;
#pragma unroll
    for (uint32_t i = 0; i < 32; i++)
    {
      uint32_t M, sh, c1u = c1, totu = tot, Mnext = 0;
      if (GENERAL)
      {
        const bool halve = tot == MAX_FREQUENCY;
        const uint32_t c1h = (c1 >> 1) + 1u;
        const uint32_t toth = ((tot - c1 + 1u) >> 1) + c1h;
        c1u = halve ? c1h : c1;
        totu = halve ? toth : tot;
        Mnext = magic[totu + 1u];
        M = Mcur;
        sh = div_shift(tot);
      }
      else
      {
        M = ADAPTIVE ? Mg[i] : Mg[0];
        if (MASKED)
          M &= (uint32_t)((int32_t)(active << i) >> 31);
        sh = sh_fast;
      }
      const uint32_t R = range_from_sum_plain(A + B);     // 1 .. 65536
      const uint32_t x1 = mulhi32(mul24(R, c1), M) >> sh; // range * cum[1] / cum[0]
      const uint32_t x2 = mulhi32(R, M) >> sh;            // range * cum[2] / cum[0], cum[2] = 1
      const uint32_t lm = (uint32_t)((int32_t)(D - x1) >> 31); // all ones unless index 1 (D, x1 < 2^17)
      const uint32_t inc = select32(lm, x2, x1);
      B = select32(lm, 0u - (A + (x1 << 16)), B);
      A += inc << 16;
      D -= inc;
      eof |= D; // D < x2 (the EOF symbol, index 3) leaves D negative: the sign bit sticks, the word is redone bit by bit
      out = GENERAL ? (out << 1) | ((lm ^ mm) & 1u) : shift_in_msb(out, lm); // fast word: lm now, mm once at the end
      if (GENERAL)
      {
        const uint32_t tie = (c1u - 1u == totu - c1u) ? 0xFFFFFFFFu : 0u;
        mm ^= lm & tie;
        c1 = c1u - (lm & ~tie);
        tot = totu + 1u;
        Mcur = Mnext;
      }
      else if (ADAPTIVE)
        c1 -= lm;
      const uint32_t n = renorm_shifts(A, B);
      A <<= n;
      B <<= n;
      const uint64_t da = (((uint64_t)D << 32) | ahead) << n; // D takes the next n bits
      D = (uint32_t)(da >> 32);
      ahead = (uint32_t)da;
      off += n;
      if (!GENERAL && ADAPTIVE && (i == 0u || i == 8u || i == 16u)) // the next quarter of the magics (see BacCoder::encode_word)
      {
        DG_COMPILER_BARRIER();
        const uint32_t *const mq = magic + tot_word + i + 8u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++)
          Mg[i + 8u + k] = mq[k];
        DG_COMPILER_BARRIER();
      }
      if ((i & 7u) == 7u) // rebuild the look-ahead from the staged words (8 symbols rarely take more than 32 bits: else redo)
      {
        DG_MATERIALISE(out); // (left alone hipcc gathers the 32 symbols' bits and their D's at the end of the word: 60 live registers)
        DG_MATERIALISE(eof);
        // all in vector arithmetic: masks made in scalar registers (compare, s_and, v_cndmask) put a VALU -> SALU -> VALU
        // round trip into the word path four times per word.  `bad` collects sign bits: more than 32 bits in this group of
        // eight symbols, or more than the four staged words hold.
        bad |= (32u - (off - off_group)) | (95u - off);
        off_group = off;
        const uint32_t m32 = (uint32_t)((int32_t)(31u - off) >> 31), m64 = (uint32_t)((int32_t)(63u - off) >> 31); // off >= 32, off >= 64
        const uint32_t lo = select32(m64, w2, select32(m32, w1, w0)), hi = select32(m64, w3, select32(m32, w2, w1));
        ahead = (uint32_t)(((((uint64_t)lo << 32) | hi) << (off & 31u)) >> 32);
      }
    }
    bits_out = GENERAL ? out : (out ^ mm) & active;
    bp = (uint64_t)k0 * 32u + off;
    if (GENERAL)
      mps = mm & 1u;
    else if (ADAPTIVE)
      tot += MASKED ? part_hi - part_lo + 1u : 32u;
    return (int32_t)(eof | bad) >= 0;
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Signed exp-Golomb parser + prefix sum (seg.c:45-94, diff.c:25-37) over the lane's decoded-bit ring.
// ---------------------------------------------------------------------------------------------------------------------
// The decoded seg bits never go to memory: the arithmetic decoder appends them to a 64-bit register window (at most 32
// at a time, when the window holds <= 32), and the parser takes codewords off its top.  Codewords longer than the
// window (up to 65 bits) are parsed in two parts: first the zero prefix (possibly over several refills), then the
// delimiting one with its residual (<= 33 bits).
struct SegParser
{
  uint64_t win;    // valid bits left aligned
  uint32_t cnt;    // number of valid bits in win
  uint32_t zeros;  // zero-prefix bits of the current codeword consumed so far
  uint32_t need;   // 0: scanning a prefix; else: prefix done, need = its length + 1 bits are wanted next
  uint32_t last;   // diff.c:27
  // valuesize < 32 (the NARROW variants below): the zero prefix is capped at valuesize + 1 bits (seg.c:55-56,74), decode
  // diff reads the decoded value back as valuesize bits, sign extended (diff.c:31-32), and writes the low valuesize
  // bits of the running sum (diff.c:34)
  uint32_t pmax, vshift, vmask;

  DG_DEV void init(uint32_t valuesize = 32)
  {
    win = 0;
    cnt = 0;
    zeros = 0;
    need = 0;
    last = 0;
    pmax = valuesize < 15u ? valuesize : 15u;
    vshift = 32u - valuesize;
    vmask = valuesize >= 32u ? 0xFFFFFFFFu : (1u << valuesize) - 1u;
  }

  DG_DEV bool has_room() const // for 32 more bits
  {
    return cnt <= 32u;
  }

  DG_DEV bool pending() const // inside a codeword
  {
    return (need | zeros) != 0u;
  }

  DG_DEV void push(uint32_t bits, uint32_t n) // n <= 32 bits, right aligned in `bits`; cnt + n <= 64
  {
    if (n > 0)
    {
      win |= (uint64_t)bits << (64u - n) >> cnt;
      cnt += n;
    }
  }

  // n <= 32 bits LEFT aligned in `word`, the rest of it zero; cnt <= 32.  Branch free (n = 0 with word = 0 changes nothing).
  DG_DEV void push_word(uint32_t word, uint32_t n)
  {
    win |= ((uint64_t)word << 32) >> (cnt & 63u);
    cnt += n;
  }

  DG_DEV void drop(uint32_t n) // n <= cnt, n <= 63... (n == 64 only when cnt == 64)
  {
    win = n >= 64u ? 0 : win << n;
    cnt -= n;
  }

  template <bool NARROW = false>
  DG_DEV void emit(uint64_t w, uint32_t &sample) // w = code_number + 1
  {
    const uint32_t mag = (uint32_t)(w >> 1);       // (code_number + 1) / 2, seg.c:76
    uint32_t d = (w & 1u) ? 0u - mag : mag;        // odd w = even code number = negative (seg.c:77-78)
    if (NARROW)
      d = (uint32_t)((int32_t)(d << vshift) >> vshift);
    last += d;                                     // diff.c:32-35
    sample = NARROW ? last & vmask : last;
  }

  // Branch-free attempt at the common case: a whole codeword of <= 31 bits on top of the window, parser between
  // codewords, `allowed` (room in the sample ring, count below T, lane not finished).  Returns true and the sample if it
  // took one; otherwise changes nothing.
  template <bool NARROW = false>
  DG_DEV bool take_short(bool allowed, uint32_t &sample)
  {
    const uint32_t top = (uint32_t)(win >> 32);
    const uint32_t p = clz32(top | 0x8000u); // 16 = no short codeword here
    const uint32_t n = 2u * p + 1u;
    const bool ok = allowed && (need | zeros) == 0u && p <= (NARROW ? pmax : 15u) && n <= cnt;
    const uint32_t w = top >> ((32u - n) & 31u);
    const uint32_t mag = w >> 1;
    uint32_t d = (w & 1u) ? 0u - mag : mag;
    if (NARROW)
      d = (uint32_t)((int32_t)(d << vshift) >> vshift);
    const uint32_t m = ok ? n : 0u;
    last += ok ? d : 0u;
    sample = NARROW ? last & vmask : last;
    win <<= m;
    cnt -= m;
    return ok;
  }

  // The same for the parsing wave's steady pass, in as few instructions as it takes (arithmetic masks instead of
  // compares and selects; the parser stands between two codewords: need == zeros == 0).  `room`: the lane may produce a
  // sample now.  Returns all ones when a codeword was taken (then `sample` is its value), else zero and nothing changed.
  template <bool NARROW = false>
  DG_DEV uint32_t take_short_lean(bool room, uint32_t &sample)
  {
    const uint32_t top = (uint32_t)(win >> 32);
    const uint32_t p2 = 2u * clz32(top | 0x8000u);      // twice the prefix length; 32 = no short codeword here
    // the codeword has p2 + 1 bits: whole iff p2 < cnt; short iff p2 <= 30 (narrow values: p <= pmax)
    const uint32_t most = NARROW ? 2u * pmax + 1u : 31u;
    const uint32_t lim = room ? (cnt < most ? cnt : most) : 0u;
    const uint32_t okm = (uint32_t)((int32_t)(p2 - lim) >> 31); // all ones iff p2 < lim (both below 2^31)
    const uint32_t w = top >> ((31u - p2) & 31u);          // code_number + 1 (junk when not ok: masked below)
    const uint32_t sgn = 0u - (w & 1u);                  // odd w = even code number = negative (seg.c:77-78)
    uint32_t d = ((w >> 1) ^ sgn) - sgn;
    if (NARROW)
      d = (uint32_t)((int32_t)(d << vshift) >> vshift);
    const uint32_t m = (p2 | 1u) & okm;
    last += d & okm;
    sample = NARROW ? last & vmask : last;
    win <<= m;
    cnt -= m;
    return okm;
  }

  // Tries to parse one codeword.  Returns 1 and the sample if a complete codeword was available, 0 if more bits are
  // needed (and `final` is false), 2 at a clean end of stream, or a negative error code.
  template <bool NARROW = false>
  DG_DEV int32_t next(bool final, uint32_t &sample)
  {
    if (need == 0 && zeros == 0)
    {
      // common case: a whole codeword of <= 31 bits on top of the window
      const uint32_t top = (uint32_t)(win >> 32);
      const uint32_t p = top ? clz32(top) : 32u;
      const uint32_t n = 2u * p + 1u;
      if (p <= (NARROW ? pmax : 15u) && n <= cnt)
      {
        emit<NARROW>(top >> (32u - n), sample);
        drop(n);
        return 1;
      }
      if (cnt == 0)
        return final ? 2 : 0;
      // Fewer than 32 bits on hand and more to come: whatever the codeword is, it is not complete yet (a complete one
      // of <= 31 bits was taken above, longer ones need more than 32 bits).  Leave the window untouched, so that the
      // branch-free take_short() can have the codeword once it is whole.
      if (!final && cnt < 32u)
        return 0;
    }
    if (need == 0) // scanning the zero prefix (seg.c:50-57)
    {
      const uint32_t lz = win ? (uint32_t)__builtin_clzll(win) : 64u;
      const uint32_t z = lz < cnt ? lz : cnt;
      zeros += z;
      if (zeros >= 33u - (NARROW ? vshift : 0u))
        return ERR_INVALID_FORMAT; // prefix cap: valuesize + 1 (seg.c:55-56,74)
      drop(z);
      if (cnt == 0) // ran out inside the prefix
        return final ? 2 : 0; // seg.c:58-62: EOF inside a (non-empty) zero prefix ends the stream (padding)
      need = zeros + 1u; // the delimiting one + residual
      zeros = 0;
    }
    if (cnt < need)
    {
      // the stream ends inside this codeword.  Right after the delimiting one (cnt == 1) the reference sees EOF with a
      // non-empty prefix and takes it for padding (seg.c:58-62, checked after the loop of :50-57); with part of the
      // residual present its read comes up short (seg.c:64)
      if (!final)
        return 0;
      return cnt == 1u ? 2 : ERR_LIBRARY_CALL;
    }
    emit<NARROW>(win >> (64u - need), sample); // (1 << prefix) | residual = code_number + 1 (seg.c:64-66)
    drop(need);
    need = 0;
    return 1;
  }
};

// valuesize 33..64: the same parser over 64-bit values.  Codewords reach 127 bits here (the reference caps the prefix
// at min(valuesize + 1, 64) zeros, seg.c:74, so |v| = 2^63 cannot be decoded by it either), more than the window holds:
// the prefix is counted across refills, the delimiting one is taken on its own and the residual is gathered in as many
// pieces as it takes.  No short-codeword fast path: these sizes are for completeness, not for speed.
struct SegParser64
{
  uint64_t win;
  uint32_t cnt;
  uint32_t zeros;      // zero-prefix bits of the current codeword counted so far
  uint32_t resid_left; // residual bits still to come (valid while in_resid)
  bool in_resid;       // the delimiting one has been taken
  uint64_t resid;
  uint64_t last;       // diff.c:27
  uint32_t cap, vshift;
  uint64_t vmask;

  DG_DEV void init(uint32_t valuesize)
  {
    win = 0;
    cnt = 0;
    zeros = 0;
    resid_left = 0;
    in_resid = false;
    resid = 0;
    last = 0;
    cap = valuesize + 1u > 64u ? 64u : valuesize + 1u;
    vshift = 64u - valuesize;
    vmask = valuesize >= 64u ? ~0ull : (1ull << valuesize) - 1ull;
  }

  DG_DEV bool has_room() const
  {
    return cnt <= 32u;
  }

  DG_DEV bool pending() const
  {
    return in_resid || zeros != 0u;
  }

  DG_DEV void push(uint32_t bits, uint32_t n)
  {
    if (n > 0)
    {
      win |= (uint64_t)bits << (64u - n) >> cnt;
      cnt += n;
    }
  }

  DG_DEV void push_word(uint32_t word, uint32_t n) // as SegParser::push_word
  {
    win |= ((uint64_t)word << 32) >> (cnt & 63u);
    cnt += n;
  }

  DG_DEV void drop(uint32_t n)
  {
    win = n >= 64u ? 0 : win << n;
    cnt -= n;
  }

  // 1 = a sample, 0 = more bits needed, 2 = clean end of stream, negative = error (as SegParser::next)
  DG_DEV int32_t next(bool final, uint64_t &sample)
  {
    if (!in_resid)
    {
      const uint32_t lz = win ? (uint32_t)__builtin_clzll(win) : 64u;
      const uint32_t z = lz < cnt ? lz : cnt;
      zeros += z;
      if (zeros >= cap)
        return ERR_INVALID_FORMAT; // seg.c:55-56
      drop(z);
      if (cnt == 0)
        return final ? 2 : 0; // EOF inside the prefix, or nothing left: padding / clean end (seg.c:58-62)
      drop(1); // the delimiting one
      in_resid = true;
      resid_left = zeros;
      resid = 0;
    }
    if (resid_left > 0)
    {
      if (cnt == 0)
      {
        if (!final)
          return 0;
        return resid_left == zeros ? 2 : ERR_LIBRARY_CALL; // EOF right after the delimiter is padding; later a short read
      }
      const uint32_t k = cnt < resid_left ? cnt : resid_left; // 1..63 (zeros < cap <= 64)
      resid = (resid << k) | (win >> (64u - k));
      drop(k);
      resid_left -= k;
      if (resid_left > 0)
        return final ? ERR_LIBRARY_CALL : 0;
    }
    const uint64_t w = (1ull << zeros) | resid; // code_number + 1 (seg.c:64-66); zeros <= 63
    const uint64_t mag = w >> 1;
    uint64_t d = (w & 1ull) ? 0ull - mag : mag; // seg.c:76-78
    if (vshift != 0)
      d = (uint64_t)((int64_t)(d << vshift) >> vshift); // decode diff reads it back as valuesize bits (diff.c:31-32)
    last += d;
    sample = last & vmask;
    zeros = 0;
    in_resid = false;
    return 1;
  }
};

} // namespace dg
