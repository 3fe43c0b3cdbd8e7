// sim_main.cpp -- C entry points that run the real kernel source under the emulator (tests/sim/hipsim.hpp).
// TEST INFRASTRUCTURE ONLY; see hipsim.hpp.
#define DEGA_SIM 1
#define dg dgsim // keep the emulated kernels' symbols apart from libdega_hip.so's
#include "hipsim.hpp"

#include "../../data-compressor_amd/csrc/dega_kernels.hpp"

#include <vector>

using namespace dg;

static std::vector<DivEntry> make_table()
{
  std::vector<DivEntry> tab(DIV_TABLE_SIZE, DivEntry{0u, 0u});
  for (uint32_t t = 3; t < DIV_TABLE_SIZE; t++)
  {
    uint32_t L = 0;
    while ((1u << L) < t)
      L++;
    const unsigned __int128 num = (unsigned __int128)1 << (30 + L);
    tab[t].magic = (uint32_t)((num + t - 1) / t);
    tab[t].shift = L - 2;
  }
  return tab;
}

extern "C" __attribute__((visibility("default"))) int sim_encode(const int32_t *x, size_t C, size_t T, size_t ld, int adaptive, uint8_t *out, size_t cap, uint64_t *bits, int32_t *err)
{
  static const std::vector<DivEntry> tab = make_table();
  EncodeArgs a{x, C, T, ld, out, cap, bits, err, tab.data()};
  const dim3 grid((unsigned)((C + BLOCK - 1) / BLOCK));
  if (adaptive)
    sim::launch(dega_encode_kernel<true>, grid, dim3(BLOCK), a);
  else
    sim::launch(dega_encode_kernel<false>, grid, dim3(BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_normalize(const float *v, size_t C, size_t T, size_t ld, float factor, int32_t *x, int32_t *err)
{
  NormalizeArgs a{v, x, C, T, ld, factor, err};
  sim::launch(dega_normalize_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK), 2), dim3(BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_denormalize(const int32_t *x, size_t C, size_t T, size_t ld, float factor, float *v)
{
  DenormalizeArgs a{x, v, C, T, ld, factor};
  sim::launch(dega_denormalize_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK), 2), dim3(BLOCK), a);
  return 0;
}

extern "C" __attribute__((visibility("default"))) int sim_synth(int32_t *x, size_t C, size_t T, size_t ld, uint64_t seed, uint64_t c0, uint32_t S)
{
  SynthArgs a{x, C, T, ld, seed, c0, S};
  sim::launch(dega_synth_kernel, dim3((unsigned)((C + BLOCK - 1) / BLOCK)), dim3(BLOCK), a);
  return 0;
}
