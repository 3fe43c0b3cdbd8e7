// hipsim.hpp -- a minimal thread-per-lane emulator of the HIP execution model, for DEBUGGING the kernels of
// data-compressor_amd/csrc/dega_kernels.hpp on a machine without a GPU.
//
// TEST INFRASTRUCTURE ONLY.  This is not a fallback and nothing in the product can reach it: it is compiled by
// tests/sim/Makefile into tests/sim/libdega_sim.so and used by tests/test_kernel_sim.py to check kernel LOGIC
// (ring indexing, phase control, termination) against the oracle before spending GPU minutes.  Speed: ~1e4x slower
// than the GPU.  Model: one OS thread per lane; a workgroup's threads run concurrently, workgroups run one after
// the other; __syncthreads() and the wave votes are barriers (valid because the kernels only vote in wave-uniform
// control flow).  The paired waves of the kernels (one codes, its partner fills / parses / searches) therefore run
// truly concurrently here and talk through their LDS words with release / acquire (dega_intrinsics.hpp: peer_store /
// peer_load); sim::drag() below slows one side down for tests that want a ring to run full.
#pragma once

#include <stdint.h>

#include <barrier>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

struct dim3
{
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace sim
{
inline thread_local dim3 t_threadIdx, t_blockIdx;
inline dim3 g_blockDim, g_gridDim;
inline std::unique_ptr<std::barrier<>> g_block_barrier;
inline std::vector<std::unique_ptr<std::barrier<>>> g_wave_barrier;
inline uint8_t g_votes[16][64];

// Test knob: the waves g_drag_from .. of a workgroup sleep g_drag_us microseconds whenever they read a word of their
// partner (peer_load) -- a slow consumer, so that the rings between paired waves run full.
inline int g_drag_from = 1 << 30, g_drag_us = 0;
inline void drag()
{
  if ((int)(t_threadIdx.x >> 6) >= g_drag_from && g_drag_us > 0)
    std::this_thread::sleep_for(std::chrono::microseconds(g_drag_us));
}

inline void block_barrier()
{
  g_block_barrier->arrive_and_wait();
}

inline uint32_t g_vals[16][64];

inline uint32_t wave_reduce(uint32_t v, bool want_min)
{
  const unsigned w = t_threadIdx.x >> 6, l = t_threadIdx.x & 63u;
  g_vals[w][l] = v;
  g_wave_barrier[w]->arrive_and_wait();
  uint32_t r = g_vals[w][0];
  for (unsigned i = 1; i < 64; i++)
    r = want_min ? (g_vals[w][i] < r ? g_vals[w][i] : r) : (g_vals[w][i] > r ? g_vals[w][i] : r);
  g_wave_barrier[w]->arrive_and_wait();
  return r;
}

inline bool wave_vote(bool p, bool want_all)
{
  const unsigned w = t_threadIdx.x >> 6, l = t_threadIdx.x & 63u;
  g_votes[w][l] = p ? 1 : 0;
  g_wave_barrier[w]->arrive_and_wait();
  unsigned n = 0;
  for (unsigned i = 0; i < 64; i++)
    n += g_votes[w][i];
  g_wave_barrier[w]->arrive_and_wait();
  return want_all ? n == 64 : n > 0;
}

template <typename K, typename A>
void launch(K kernel, dim3 grid, dim3 block, const A &args)
{
  g_blockDim = block;
  g_gridDim = grid;
  for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++)
    {
      g_block_barrier = std::make_unique<std::barrier<>>(block.x);
      g_wave_barrier.clear();
      for (unsigned w = 0; w < (block.x + 63) / 64; w++)
        g_wave_barrier.push_back(std::make_unique<std::barrier<>>(64));
      std::vector<std::thread> th;
      for (unsigned t = 0; t < block.x; t++)
        th.emplace_back([=]() {
          t_threadIdx = dim3(t);
          t_blockIdx = dim3(bx, by);
          kernel(args);
        });
      for (auto &x : th)
        x.join();
    }
}
} // namespace sim

#define threadIdx sim::t_threadIdx
#define blockIdx sim::t_blockIdx
#define blockDim sim::g_blockDim
#define gridDim sim::g_gridDim
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __syncthreads() sim::block_barrier()

namespace dg
{
inline bool wave_any(bool p)
{
  return sim::wave_vote(p, false);
}
inline bool wave_all(bool p)
{
  return sim::wave_vote(p, true);
}
inline uint32_t wave_min_u32(uint32_t v)
{
  return sim::wave_reduce(v, true);
}
inline uint32_t wave_max_u32(uint32_t v)
{
  return sim::wave_reduce(v, false);
}
} // namespace dg
