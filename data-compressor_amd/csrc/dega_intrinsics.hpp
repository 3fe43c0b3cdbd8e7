// dega_intrinsics.hpp -- every place where the DEGA / LZMH kernels name a gfx950 instruction or builtin directly.
//
// Each wrapper has up to three bodies: the shipped one (hipcc, gfx950), the one of the thread-per-lane emulator that
// compiles the same kernel source with g++ for debugging without a GPU (DEGA_SIM; tests/sim/, test tooling only), and --
// for a few -- an ablation variant of the timing builds (DEGA_DIAG; csrc/Makefile `diag`, never shipped).  Keeping them
// here lets the hot loops in dega_lane.hpp / dega_kernels.hpp read as the code that runs.
#pragma once

#include <stdint.h>
#if defined(DEGA_SIM)
#include <sched.h>
#endif

#if defined(DEGA_SIM)
#define DG_DEV inline
#define DG_MATERIALISE(x) ((void)0)
#else
#define DG_DEV __device__ __forceinline__
// the value must be in its register here: stops hipcc from sinking a load down to its first use (where its latency
// would sit in the per-symbol dependency chain)
#define DG_MATERIALISE(x) asm volatile("" : "+v"(x))
#endif
#define DG_COMPILER_BARRIER() asm volatile("" ::: "memory")

namespace dg
{

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

DG_DEV uint32_t mul24(uint32_t a, uint32_t b) // both < 2^24
{
#if defined(DEGA_SIM)
  return a * b;
#else
  return (uint32_t)__umul24(a, b);
#endif
}

DG_DEV uint32_t bswap32(uint32_t x)
{
  return __builtin_bswap32(x);
}

#if defined(DEGA_SIM)
inline float __uint_as_float(uint32_t u)
{
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
#endif

DG_DEV uint32_t select32(uint32_t mask, uint32_t if_set, uint32_t if_clear) // mask is all ones or all zeros
{
#if defined(DEGA_SIM)
  return (if_set & mask) | (if_clear & ~mask);
#else
  uint32_t r; // one v_bfi_b32; left to itself hipcc rebuilds the select from a compare, two v_cndmask and and/or
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(if_set), "v"(if_clear));
  return r;
#endif
}

// Any function of three words bit by bit in one instruction (v_bitop3_b32, gfx950); the table is the function applied
// to 0xF0, 0xCC, 0xAA.  hipcc finds these itself in plain expressions, but not across an inline-asm operand.
DG_DEV uint32_t xor_then_or(uint32_t a, uint32_t b, uint32_t c) // (a ^ b) | c
{
#if defined(DEGA_SIM)
  return (a ^ b) | c;
#else
  return __builtin_amdgcn_bitop3_b32(a, b, c, ((0xF0 ^ 0xCC) | 0xAA) & 0xFF);
#endif
}

DG_DEV uint32_t andn_and(uint32_t a, uint32_t b, uint32_t c) // ~a & b & c
{
#if defined(DEGA_SIM)
  return ~a & b & c;
#else
  return __builtin_amdgcn_bitop3_b32(a, b, c, (~0xF0 & 0xCC & 0xAA) & 0xFF);
#endif
}

// Index of the lowest set bit; all ones for x == 0 (what v_ffbl_b32 does by itself: __builtin_ctz adds a compare and a select)
DG_DEV uint32_t lowest_bit_or_ones(uint32_t x)
{
#if defined(DEGA_SIM)
  return x != 0 ? (uint32_t)__builtin_ctz(x) : 0xFFFFFFFFu;
#else
  uint32_t r;
  asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#endif
}

// Number of leading one bits of x, for x with bit 31 set and not all ones (v_ffbh_i32 counts the bits equal to the sign).
DG_DEV uint32_t leading_ones(uint32_t x)
{
#if defined(DEGA_SIM)
  return (uint32_t)__builtin_clz(~x);
#else
  uint32_t r;
  asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#endif
}

// (acc << 1) | (x >> 31) in one instruction (funnel shift)
DG_DEV uint32_t shift_in_msb(uint32_t acc, uint32_t x)
{
#if defined(DEGA_SIM)
  return (acc << 1) | (x >> 31);
#else
  return __builtin_amdgcn_alignbit(acc, x, 31);
#endif
}

// range = 65536 - (x >> 16) for x = A + B in one instruction (SDWA: the high word of x as the subtrahend); x = 0 is the
// full range 65536, which is why this is not simply -x >> 16
DG_DEV uint32_t range_from_sum(uint32_t x)
{
#if defined(DEGA_SIM) || (defined(DEGA_DIAG) && (DEGA_DIAG & 64))
  return 0x10000u - (x >> 16);
#else
  uint32_t r;
  const uint32_t full = 0x10000u;
  asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(full), "v"(x));
  return r;
#endif
}

// acc + (x << 16) over 64 bits for a 32-bit x, in one instruction: the multiply-add takes the 32-bit operand as it is
// (as a shift the same is two: the 64-bit shift-and-add only shifts by up to 4)
DG_DEV uint64_t add_shifted16(uint64_t acc, uint32_t x)
{
#if defined(DEGA_SIM)
  return acc + ((uint64_t)x << 16);
#else
  uint64_t r;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x10000u), "v"(acc) : "vcc");
  return r;
#endif
}

// The same written in C++, for the decoder: hipcc's SDWA peephole folds the shift into the subtraction and, unlike behind
// inline asm, adds no wait state (measured: the decoder is 0.4 % faster this way, the encoder 1.4 % slower)
DG_DEV uint32_t range_from_sum_plain(uint32_t x)
{
  return 0x10000u - (x >> 16);
}

#if !defined(DEGA_SIM)
DG_DEV bool wave_any(bool p)
{
  return __any((int)p) != 0;
}
DG_DEV bool wave_all(bool p)
{
  return __all((int)p) != 0;
}
DG_DEV uint32_t wave_min_u32(uint32_t v) // butterfly over the 64 lanes
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
  {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = o < v ? o : v;
  }
  return v;
}
DG_DEV uint32_t wave_max_u32(uint32_t v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
  {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}
#endif

#if defined(DEGA_DIAG) && (DEGA_DIAG & (32 | 256)) && !defined(DEGA_SIM)
// diagnostic build: per-wave cycle totals per section of a loop (s_memtime): 32 = the LZMH searching wave, dumped over
// out_bits[]; 256 = the LZMH reading wave, dumped over out_len[]
#define DG_STAMP_DECL uint64_t stamp_t0 = __builtin_amdgcn_s_memtime(), stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define DG_STAMP(k) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); stamp_sum[k] += now_ - stamp_t0; stamp_cnt[k]++; stamp_t0 = now_; } while (0)
#else
#define DG_STAMP_DECL
#define DG_STAMP(k)
#endif

#if !defined(DEGA_SIM)
// One dword per lane from global memory straight into LDS (LDS-DMA): lane l's dword lands at lds_row[l].
// lds_row must be wave uniform.  Completion is NOT tracked by hipcc: wait with wait_vector_memory() before reading.
DG_DEV void dma_row_to_lds(const int32_t *src, uint32_t *lds_row, uint32_t /*lane*/)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_row, 4, 0, 0);
}
// Four dwords per lane (16-byte aligned source): lane l's 16 bytes land at lds_base + 16*l.
DG_DEV void dma_x4_to_lds(const int32_t *src, uint32_t *lds_base, uint32_t /*lane*/)
{
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                   (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}
DG_DEV void wait_vector_memory()
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
DG_DEV void wait_lds() // every LDS read of this wave so far has returned
{
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
#else
inline void dma_x4_to_lds(const int32_t *src, uint32_t *lds_base, uint32_t lane)
{
  for (uint32_t k = 0; k < 4; k++)
    lds_base[lane * 4 + k] = (uint32_t)src[k];
}
inline void dma_row_to_lds(const int32_t *src, uint32_t *lds_row, uint32_t lane)
{
  lds_row[lane] = (uint32_t)*src;
}
inline void wait_vector_memory() {}
inline void wait_lds() {}
#endif

// ---- two waves of one workgroup talking through LDS (the paired-wave kernels: one wave codes, its partner parses) -----
// A wave's LDS instructions execute in the order it issued them, so "write the data, then write the counter" needs no
// fence in hardware; the wrappers only stop the COMPILER from reordering or caching the accesses.  The emulator's lanes
// are OS threads: release / acquire there.
DG_DEV void peer_store(uint32_t *lds_word, uint32_t v)
{
#if defined(DEGA_SIM)
  __atomic_store_n(lds_word, v, __ATOMIC_RELEASE);
#else
  // (a volatile access through a generic pointer would become a FLAT instruction -- the slow path to LDS, tracked by
  // vmcnt together with the wave's global traffic: name the address space)
  DG_COMPILER_BARRIER();
  *(volatile __attribute__((address_space(3))) uint32_t *)lds_word = v;
  DG_COMPILER_BARRIER();
#endif
}
DG_DEV uint32_t peer_load(const uint32_t *lds_word)
{
#if defined(DEGA_SIM)
  sim::drag(); // (emulator test knob: a slow partner, tests/sim/hipsim.hpp)
  return __atomic_load_n(lds_word, __ATOMIC_ACQUIRE);
#else
  DG_COMPILER_BARRIER();
  const uint32_t v = *(const volatile __attribute__((address_space(3))) uint32_t *)lds_word;
  DG_COMPILER_BARRIER();
  return v;
#endif
}
// A word in device memory that the HOST side (a copy engine, on another stream) writes while the kernel runs: read it
// past every cache (system scope, acquire).
DG_DEV uint32_t load_written_by_host(const uint32_t *global_word)
{
#if defined(DEGA_SIM)
  return __atomic_load_n(global_word, __ATOMIC_ACQUIRE);
#else
  return __hip_atomic_load(global_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
// A word the HOST reads while the kernel runs (pinned host memory): every store of this wave so far is written back and
// visible to the host side -- a copy engine included -- before the word is (system scope, release).
DG_DEV void store_read_by_host(uint32_t *host_word, uint32_t v)
{
#if defined(DEGA_SIM)
  __atomic_store_n(host_word, v, __ATOMIC_RELEASE);
#else
  __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
// nothing to do until the partner has moved: leave the SIMD's issue slots to it for about 64 * n cycles
template <int N>
DG_DEV void wave_sleep()
{
#if defined(DEGA_SIM)
  sched_yield();
#else
  __builtin_amdgcn_s_sleep(N);
#endif
}
template <int P>
DG_DEV void wave_priority() // 0 (default) .. 3: which of a SIMD's waves issues first when both are ready
{
#if !defined(DEGA_SIM)
  __builtin_amdgcn_s_setprio(P);
#endif
}
DG_DEV uint32_t wave_uniform(uint32_t v) // v is the same in every lane: tell the compiler (scalar register)
{
#if defined(DEGA_SIM)
  return v;
#else
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// float32 arithmetic with exactly one IEEE rounding per operation and no contraction into an FMA (normalize.c's
// value * factor +- 0.5 and value / factor as the reference's compiler emits them: mulss, addss/subss, divss)
DG_DEV float fmul_once(float a, float b)
{
#if defined(DEGA_SIM)
  volatile float r = a * b;
  return r;
#else
  return __fmul_rn(a, b);
#endif
}
DG_DEV float fadd_once(float a, float b)
{
#if defined(DEGA_SIM)
  volatile float r = a + b;
  return r;
#else
  return __fadd_rn(a, b);
#endif
}
DG_DEV float fdiv_once(float a, float b)
{
#if defined(DEGA_SIM)
  volatile float r = a / b;
  return r;
#else
  return __fdiv_rn(a, b);
#endif
}

// four dwords as one 16-byte store to a 4-byte aligned address
DG_DEV void store_x4(uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
#if defined(DEGA_SIM)
  dst[0] = a;
  dst[1] = b;
  dst[2] = c;
  dst[3] = d;
#else
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
  const u32x4 v = {a, b, c, d};
  *reinterpret_cast<u32x4 *>(dst) = v;
#endif
}

} // namespace dg
