// sim_main.cpp -- C entry points that run the real kernel source under the emulator (tests/sim/hipsim.hpp).
// TEST INFRASTRUCTURE ONLY; see hipsim.hpp.
#define DEGA_SIM 1
#define dg dgsim // keep the emulated kernels' symbols apart from libdega_hip.so's
#include "hipsim.hpp"

#include "../../data-compressor_amd/csrc/dega_kernels.hpp"
#include "../../data-compressor_amd/csrc/lzmh_kernels.hpp"

#include <vector>

using namespace dg;

static std::vector<uint32_t> make_table()
{
  std::vector<uint32_t> tab(DIV_TABLE_SIZE + 32, 0u); // + the look-ahead of BacEncoder::fetch_magics
  for (uint32_t t = 3; t < DIV_TABLE_SIZE; t++)
  {
    uint32_t L = 0;
    while ((1u << L) < t)
      L++;
    const unsigned __int128 num = (unsigned __int128)1 << (30 + L);
    tab[t] = (uint32_t)((num + t - 1) / t);
  }
  return tab;
}

extern "C" __attribute__((visibility("default"))) void sim_set_drag(int from_wave, int microseconds)
{
  sim::g_drag_from = from_wave;
  sim::g_drag_us = microseconds;
}

extern "C" __attribute__((visibility("default"))) int sim_encode_vs(const int32_t *x, size_t C, size_t T, size_t ld, int adaptive, int valuesize, uint8_t *out, size_t cap, uint64_t *bits, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  EncodeArgs a{x, C, T, ld, out, cap, bits, err, tab.data(), (uint32_t)valuesize};
  const dim3 grid((unsigned)((C + ENC_CHANNELS - 1) / ENC_CHANNELS));
  if (valuesize < 32)
  {
    if (adaptive)
      sim::launch(dega_encode_kernel<true, true>, grid, dim3(ENC_BLOCK), a);
    else
      sim::launch(dega_encode_kernel<false, true>, grid, dim3(ENC_BLOCK), a);
  }
  else if (adaptive)
    sim::launch(dega_encode_kernel<true>, grid, dim3(ENC_BLOCK), a);
  else
    sim::launch(dega_encode_kernel<false>, grid, dim3(ENC_BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_encode(const int32_t *x, size_t C, size_t T, size_t ld, int adaptive, uint8_t *out, size_t cap, uint64_t *bits, int32_t *err)
{
  return sim_encode_vs(x, C, T, ld, adaptive, 32, out, cap, bits, err);
}

extern "C" __attribute__((visibility("default"))) int sim_decode_var_vs(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld, int adaptive, int valuesize, int32_t *x, uint64_t *counts, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  DecodeArgs a{in, cap, in_bits, C, T, ld, x, err, tab.data(), counts, (uint32_t)valuesize};
  const dim3 grid((unsigned)((C + DEC_CHANNELS - 1) / DEC_CHANNELS));
  if (valuesize < 32)
  {
    if (adaptive)
      sim::launch(dega_decode_kernel<true, true>, grid, dim3(DEC_BLOCK), a);
    else
      sim::launch(dega_decode_kernel<false, true>, grid, dim3(DEC_BLOCK), a);
  }
  else if (adaptive)
    sim::launch(dega_decode_kernel<true>, grid, dim3(DEC_BLOCK), a);
  else
    sim::launch(dega_decode_kernel<false>, grid, dim3(DEC_BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_decode_vs(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld, int adaptive, int valuesize, int32_t *x, int32_t *err)
{
  return sim_decode_var_vs(in, cap, in_bits, C, T, ld, adaptive, valuesize, x, nullptr, err);
}

extern "C" __attribute__((visibility("default"))) int sim_decode(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld, int adaptive, int32_t *x, int32_t *err)
{
  return sim_decode_var_vs(in, cap, in_bits, C, T, ld, adaptive, 32, x, nullptr, err);
}

extern "C" __attribute__((visibility("default"))) int sim_normalize(const float *v, size_t C, size_t T, size_t ld, float factor, int32_t *x, int32_t *err)
{
  NormalizeArgs a{v, x, C, T, ld, factor, err, -2147483648.0f, 2147483648.0f, 0xFFFFFFFFu};
  sim::launch(dega_normalize_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK), 2), dim3(BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_denormalize(const int32_t *x, size_t C, size_t T, size_t ld, float factor, float *v)
{
  DenormalizeArgs a{x, v, C, T, ld, factor, 0u};
  sim::launch(dega_denormalize_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK), 2), dim3(BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_synth(int32_t *x, size_t C, size_t T, size_t ld, uint64_t seed, uint64_t c0, uint32_t S)
{
  SynthArgs a{x, C, T, ld, seed, c0, S};
  sim::launch(dega_synth_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK)), dim3(BLOCK), a);
  return 0;
}


// the channels' rows coded over several launches (rows [cuts[k], cuts[k+1]) each), the lanes' state saved in between
extern "C" __attribute__((visibility("default"))) int sim_encode_segments(const int32_t *x, size_t C, size_t T, size_t ld, int adaptive, const size_t *cuts, int ncuts, uint8_t *out,
                                                                       size_t cap, uint64_t *bits, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  std::vector<uint32_t> state(ENC_STATE_WORDS * C, 0xDEADBEEFu);
  const dim3 grid((unsigned)((C + ENC_CHANNELS - 1) / ENC_CHANNELS));
  for (int k = 0; k + 1 < ncuts; k++)
  {
    EncodeArgs a{x + cuts[k] * ld, C, cuts[k + 1] - cuts[k], ld, out, cap, bits, err, tab.data(), 32u};
    a.seg_state = state.data();
    a.seg_flags = (k > 0 ? ENC_SEG_CONTINUES : 0u) | (k + 2 < ncuts ? ENC_SEG_MORE : 0u);
    if (adaptive)
      sim::launch(dega_encode_kernel<true>, grid, dim3(ENC_BLOCK), a);
    else
      sim::launch(dega_encode_kernel<false>, grid, dim3(ENC_BLOCK), a);
  }
  (void)T;
  return 0;
}

// ---- direct test of the encoder's word paths against its bit-at-a-time path, on random (and nasty) states ------------
#include <random>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <bool ADAPTIVE>
static int fast_vs_slow(uint64_t seed, int rounds, int *word_taken, int *ripples)
{
  static const std::vector<uint32_t> tab = make_table();
  std::mt19937_64 rng(seed);
  int bad = 0;
  constexpr uint32_t RAWT = 128; // a raw "ring" that holds everything a word can produce: the writer runs afterwards
  for (int r = 0; r < rounds; r++)
  {
    std::vector<uint32_t> buf_a(64), buf_b(64);
    const bool nasty = (rng() % 4) == 0;
    for (int i = 0; i < 64; i++)
      buf_a[i] = buf_b[i] = (nasty && (rng() % 4)) ? 0xFFFFFFFFu : (uint32_t)rng();
    std::vector<uint32_t> ring_a(ENC_ORING * 64), ring_b(ENC_ORING * 64), raw_a(RAWT * 64), raw_b(RAWT * 64);
    uint32_t nothing_absorbed = 0;
    BacCoder<ADAPTIVE, RAWT> e;
    e.init(raw_a.data(), &nothing_absorbed);
    // a normalised interval: start < H <= end and not (start >= Q and end < 3Q)
    uint32_t s, en;
    do
    {
      s = (uint32_t)(rng() % 0x8000u);
      en = 0x8000u + (uint32_t)(rng() % 0x8000u);
    } while (s >= 0x4000u && en < 0xC000u);
    uint32_t A = s << 16;
    e.B = (65535u - en) << 16;
    if (rng() & 1)
    {
      A |= 0x80000000u; // the "spurious" top bits left by an E3 step
      e.B |= 0x80000000u;
    }
    e.L = ((uint64_t)2 << 32) | A; // at a word boundary the finished bits have just been dumped
    if (ADAPTIVE)
    {
      do
      {
        const unsigned kind = (unsigned)(rng() % 5);
        e.tot = kind == 0 ? 16383u - (uint32_t)(rng() % 40u) : kind == 1 ? 3u + (uint32_t)(rng() % 70u) : 3u + (uint32_t)(rng() % 16380u);
        e.c1 = 2 + (uint32_t)(rng() % ((e.tot - 1) / 2));                 // 1 <= f2 <= f1
        if (kind == 1 || (rng() % 8) == 0)
          e.c1 = (e.tot + 1) / 2 - (uint32_t)(rng() % 2 ? 0 : (e.tot > 8 ? rng() % 3 : 0)); // f2 == f1 or nearly: swaps
        if (kind == 4)
          e.c1 = 2 + (uint32_t)(rng() % (1 + e.tot / 64));                // skewed counts: long codes for the rare symbol
      } while (e.c1 < 2 && e.tot < 3);
      e.mps = (uint32_t)(rng() & 1);
    }
    // the writer's side of the lane: bits waiting in F, a held-back word that may be all ones, words already stored
    BacWriter<> we;
    we.init(buf_a.data(), 64, ring_a.data());
    we.fcnt = (uint32_t)(rng() % 32u);
    we.F = (nasty ? ~(uint64_t)0 : rng()) & (((uint64_t)1 << we.fcnt) - 1);
    we.prev = nasty ? 0xFFFFFFFFu - (uint32_t)(rng() % 2) : (uint32_t)rng();
    we.pos = 1 + (uint32_t)(rng() % 40u);
    we.drained = we.pos - 1;
    BacWriter<> wf = we;
    wf.dst = buf_b.data();
    wf.oring = ring_b.data();
    BacCoder<ADAPTIVE, RAWT> f = e;
    f.raw = raw_b.data();
    // the rare symbol really is rare in most words of the skewed states, frequent in the others
    uint32_t word = (rng() % 3) ? (uint32_t)rng() : (uint32_t)(rng() & rng() & rng());
    if ((rng() % 6) == 0)
      word = (rng() & 1) ? 0xFFFFFFFFu : 0xFFFFFFFFu << (rng() % 32); // the rare symbol 32 times over: the most bits a word can make
    if (e.mps)
      word = ~word;
    // (a) bit at a time
    for (uint32_t i = 0; i < 32; i++)
      e.encode_bit((word >> (31u - i)) & 1u, tab.data());
    e.end_bits_word();
    // (b) the word path of the lane's class -- or, two rounds out of three, of a more general class, as happens when
    //     another lane of the wave needs one
    f.classify();
    uint32_t cls = f.cls;
    if (cls <= CLS_GENERAL && ADAPTIVE && (r % 3) == 1)
    {
      cls = CLS_GENERAL;
      f.whole_word();
    }
    else if ((cls == CLS_FAST8 || cls == CLS_FAST4) && (r % 3) == 2)
      cls = CLS_FAST4;
    if (cls == CLS_SPLIT)
    {
      // a halving word: two masked fast steps (one when the halving belongs to the last symbol), as the kernel does it
      if constexpr (ADAPTIVE)
      {
        (*word_taken)++;
        for (int part = 0; part < 2; part++)
        {
          uint32_t Mg[32];
          f.fetch_magics_first(tab.data(), Mg);
          f.template encode_word<false, 8, true>(word, tab.data(), Mg);
          if (f.after_part(word))
            break;
        }
      }
    }
    else if (cls == CLS_BITS)
    {
      for (uint32_t i = 0; i < 32; i++)
        f.encode_bit((word >> (31u - i)) & 1u, tab.data());
      f.end_bits_word();
    }
    else
    {
      (*word_taken)++;
      uint32_t Mg[32];
      f.fetch_magics_first(tab.data(), Mg);
      if (cls == CLS_FAST8)
        f.template encode_word<false, 8>(word, tab.data(), Mg);
      else if (cls == CLS_FAST4)
        f.template encode_word<false, 4>(word, tab.data(), Mg);
      else if constexpr (ADAPTIVE)
        f.template encode_word<true, 4>(word, tab.data(), Mg);
    }
    // the writers absorb what the two coders dumped; a carry past the held-back word ripples into the stored words
    const uint32_t before_a = buf_a[we.pos >= 2 ? we.pos - 2 : 0], before_b = before_a;
    for (uint32_t k = 0; k < e.rwr; k++)
      we.absorb(raw_a[(k % RAWT) * 64u]);
    for (uint32_t k = 0; k < f.rwr; k++)
      wf.absorb(raw_b[(k % RAWT) * 64u]);
    (void)before_b;
    // Bring both to the same representation -- a carry may still wait above F's bits where the other path has already
    // added it to the held-back word (the same number either way) -- flush, and compare everything observable
    for (BacWriter<> *p : {&we, &wf})
    {
      const uint32_t carry = (uint32_t)(p->F >> p->fcnt);
      p->F &= ((uint64_t)1 << p->fcnt) - 1;
      const uint32_t sum = p->prev + carry;
      if (sum < carry)
      {
        (*ripples)++;
        p->ripple_carry_from(p->pos - 1u);
      }
      p->prev = sum;
    }
    we.drain_lane();
    wf.drain_lane();
    if (buf_a[we.pos >= 2 ? we.pos - 2 : 0] != before_a)
      (*ripples)++;
    const bool same = e.L == f.L && e.B == f.B && e.c1 == f.c1 && e.tot == f.tot && e.mps == f.mps && we.F == wf.F && we.fcnt == wf.fcnt &&
                      we.prev == wf.prev && we.pos == wf.pos && we.drained == wf.drained && we.err == wf.err && buf_a == buf_b && (f.rwr & 3u) == 0u;
    if (!same)
    {
      bad++;
      if (bad <= 6 && getenv("DEGA_SIM_VERBOSE") != nullptr)
        fprintf(stderr, "round %d cls %u nasty %d: L %d B %d c1 %d tot %d mps %d F %d (%llx %llx) fcnt %d (%u %u) prev %d (%x %x) pos %d (%u %u) drained %d err %d buf %d rwr %u\n", r, cls,
                (int)nasty, e.L == f.L, e.B == f.B, e.c1 == f.c1, e.tot == f.tot, e.mps == f.mps, we.F == wf.F, (unsigned long long)we.F, (unsigned long long)wf.F,
                we.fcnt == wf.fcnt, we.fcnt, wf.fcnt, we.prev == wf.prev, we.prev, wf.prev, we.pos == wf.pos, we.pos, wf.pos, we.drained == wf.drained, we.err == wf.err, buf_a == buf_b, f.rwr);
    }
  }
  return bad;
}

extern "C" __attribute__((visibility("default"))) int sim_fast_vs_slow(uint64_t seed, int rounds, int adaptive, int *fast_taken, int *redo_taken)
{
  *fast_taken = 0;
  *redo_taken = 0;
  return adaptive ? fast_vs_slow<true>(seed, rounds, fast_taken, redo_taken) : fast_vs_slow<false>(seed, rounds, fast_taken, redo_taken);
}

extern "C" __attribute__((visibility("default"))) int sim_lzmh_encode(const uint8_t *in, size_t stride, const uint64_t *in_len, size_t C, uint8_t *out, size_t cap, uint64_t *bits, int32_t *err)
{
  LzmhEncodeArgs a{in, stride, in_len, C, out, cap, bits, err};
  sim::launch(lzmh_encode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZ_ENC_THREADS), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_lzmh_render(const int32_t *x, size_t C, size_t T, size_t ld, uint8_t *out, size_t stride, uint64_t *out_len, int32_t *err)
{
  RenderArgs a{x, C, T, ld, out, stride, out_len, err};
  sim::launch(lzmh_render_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_lzmh_decode(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out, size_t stride, uint64_t *out_len, int32_t *err)
{
  LzmhDecodeArgs a{in, cap, in_bits, C, out, stride, out_len, err};
  sim::launch(lzmh_decode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZD_THREADS), a);
  return 0;
}

// the 8-pair workgroup shape that the library uses for decoding batches of more than 64 Ki channels, forced on a small batch
extern "C" __attribute__((visibility("default"))) int sim_decode_wide(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld, int adaptive, int32_t *x, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  DecodeArgs a{in, cap, in_bits, C, T, ld, x, err, tab.data(), nullptr, 32u};
  const dim3 grid((unsigned)((C + 511) / 512)); // 8 pairs of waves, 16-sample ring
  if (adaptive)
    sim::launch(dega_decode_kernel<true, false, false, false, 8, false>, grid, dim3(1024), a);
  else
    sim::launch(dega_decode_kernel<false, false, false, false, 8, false>, grid, dim3(1024), a);
  return 0;
}

// valuesize 33..64: int64 containers
extern "C" __attribute__((visibility("default"))) int sim_encode64(const int64_t *x, size_t C, size_t T, size_t ld, int adaptive, int valuesize, uint8_t *out, size_t cap, uint64_t *bits, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  EncodeArgs a{reinterpret_cast<const int32_t *>(x), C, T, ld, out, cap, bits, err, tab.data(), (uint32_t)valuesize};
  const dim3 grid((unsigned)((C + ENC_CHANNELS - 1) / ENC_CHANNELS));
  if (adaptive)
    sim::launch(dega_encode_kernel<true, false, 4, 32, 16, 32, true>, grid, dim3(ENC_BLOCK), a);
  else
    sim::launch(dega_encode_kernel<false, false, 4, 32, 16, 32, true>, grid, dim3(ENC_BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_decode64(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, size_t T, size_t ld, int adaptive, int valuesize, int64_t *x, uint64_t *counts, int32_t *err)
{
  static const std::vector<uint32_t> tab = make_table();
  DecodeArgs a{in, cap, in_bits, C, T, ld, reinterpret_cast<int32_t *>(x), err, tab.data(), counts, (uint32_t)valuesize};
  const dim3 grid((unsigned)((C + DEC_CHANNELS - 1) / DEC_CHANNELS));
  if (adaptive)
    sim::launch(dega_decode_kernel<true, false, true>, grid, dim3(DEC_BLOCK), a);
  else
    sim::launch(dega_decode_kernel<false, false, true>, grid, dim3(DEC_BLOCK), a);
  return 0;
}

// the half-filled-wave launch the library uses for up to 64 Ki channels
extern "C" __attribute__((visibility("default"))) int sim_lzmh_decode_half(const uint8_t *in, size_t cap, const uint64_t *in_bits, size_t C, uint8_t *out, size_t stride, uint64_t *out_len, int32_t *err)
{
  LzmhDecodeArgs a{in, cap, in_bits, C, out, stride, out_len, err};
  sim::launch(lzmh_decode_kernel, dim3((unsigned)((C + LZ_BLOCK - 1) / LZ_BLOCK)), dim3(LZD_THREADS), a); // (one shape since the pairs of waves)
  return 0;
}
