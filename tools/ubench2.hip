// ubench2.hip -- VALU issue cost on gfx950 per opcode CLASS, with dependent and independent instruction streams, at
// 1, 2 and 4 waves per SIMD (one workgroup of 256 / 512 / 1024 threads per CU, 96 KiB of LDS each so that a CU takes one).
//
// Why: tools/ubench.hip (round 1) timed one dependent chain per opcode and found ~4 cycles per wave-instruction whatever
// the opcode -- except v_add_u32, which ran at 2.1 cycles aggregate with two waves on a SIMD.  The DEGA kernels run at two
// waves per SIMD, and their issue bound is priced at 4 cycles per instruction: this settles what the 4 depends on
// (encoding VOP1/VOP2 vs VOP3, operand width, SDWA / DPP, dependence), and what two DIFFERENT streams do to each other.
//
// Every wave times itself with s_memtime around ITERS x 64 instructions; the table prints the mean over waves of
// cycles per instruction per wave (so "aggregate cycles per instruction per SIMD" = that / waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2 && tools/ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 3000;

// operand forms; r = the rotating destination register (%0..%7), %8 %9 = loop-invariant sources, %10 = 64-bit invariant
#define F1(op, r) op " %" #r ", %" #r "\n\t"
#define F2(op, r) op " %" #r ", %" #r ", %8\n\t"
#define F2S(op, r) op " %" #r ", %8, %" #r "\n\t"
#define F2C(op, r) op " %" #r ", 5, %" #r "\n\t"
#define F2L(op, r) op " %" #r ", 0x12345678, %" #r "\n\t"
#define F2V(op, r) op " %" #r ", %" #r ", %8, vcc\n\t"
#define F2CO(op, r) op " %" #r ", vcc, %" #r ", %8\n\t"
#define F2CI(op, r) op " %" #r ", vcc, %" #r ", %8, vcc\n\t"
#define F3(op, r) op " %" #r ", %" #r ", %8, %9\n\t"
#define F3C(op, r) op " %" #r ", %" #r ", 3, %9\n\t"
#define FCMP(op, r) op " vcc, %" #r ", %8\n\t"
#define FCMPS(op, r) op " s[4:5], %" #r ", %8\n\t"
#define FSDWA(op, r) op " %" #r ", %8, %" #r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
#define FDPP(op, r) op " %" #r ", %" #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define FDPP2(op, r) op " %" #r ", %" #r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define FMAD64(op, r) op " %" #r ", vcc, %8, %9, %" #r "\n\t"
#define F64S(op, r) op " %" #r ", %8, %" #r "\n\t"
#define F64A(op, r) op " %" #r ", %" #r ", 1, %10\n\t"
#define FQSAD(op, r) op " %" #r ", %" #r ", %8, %" #r "\n\t"
#define FSAD(op, r) op " %" #r ", %" #r ", %8, %9\n\t"

#define ROT8(F, op) F(op, 0) F(op, 1) F(op, 2) F(op, 3) F(op, 4) F(op, 5) F(op, 6) F(op, 7)
#define DEP8(F, op) F(op, 0) F(op, 0) F(op, 0) F(op, 0) F(op, 0) F(op, 0) F(op, 0) F(op, 0)
#define R8(x) x x x x x x x x
// two different instructions alternating, each on its own four rotating registers
#define MIX8(FA, opa, FB, opb) FA(opa, 0) FB(opb, 4) FA(opa, 1) FB(opb, 5) FA(opa, 2) FB(opb, 6) FA(opa, 3) FB(opb, 7)

#define BODY32(STR)                                                                                                          \
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
  uint32_t b = seed | 1u, c = seed + 3u;                                                                                     \
  uint64_t q = seed;                                                                                                         \
  const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                         \
  for (int i = 0; i < ITERS; i++)                                                                                            \
    asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(q) : "vcc", "s4", "s5"); \
  const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                         \
  sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                               \
  dt = t1 - t0;

#define BODY64(STR)                                                                                                          \
  uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
  uint32_t b = seed | 1u, c = seed + 3u;                                                                                     \
  uint64_t q = seed;                                                                                                         \
  const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                         \
  for (int i = 0; i < ITERS; i++)                                                                                            \
    asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(q) : "vcc", "s4", "s5"); \
  const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                         \
  sink = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                                   \
  dt = t1 - t0;

// every wave runs stream A; or: waves of the first half of the workgroup run A and those of the second half B (wave w and
// w + n/2 share a SIMD: the CU deals a workgroup's waves out to its SIMDs in turn)
#define KERN(name, BODY, STRA, STRB)                                                          \
  __global__ void __launch_bounds__(1024) name(uint64_t *out, uint32_t seed, int split)      \
  {                                                                                           \
    extern __shared__ uint32_t lds_pad[];                                                     \
    uint32_t sink = 0;                                                                        \
    uint64_t dt = 0;                                                                          \
    const bool second = split != 0 && threadIdx.x >= blockDim.x / 2;                          \
    if (!second)                                                                              \
    {                                                                                         \
      BODY(STRA)                                                                              \
    }                                                                                         \
    else                                                                                      \
    {                                                                                         \
      BODY(STRB)                                                                              \
    }                                                                                         \
    if (sink == 0x7fffffffu)                                                                  \
      lds_pad[threadIdx.x] = sink;                                                            \
    if ((threadIdx.x & 63u) == 0)                                                             \
      out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = dt;                                 \
  }

#define K_IND(name, F, op) KERN(name##_ind, BODY32, R8(ROT8(F, op)), R8(ROT8(F, op)))
#define K_DEP(name, F, op) KERN(name##_dep, BODY32, R8(DEP8(F, op)), R8(DEP8(F, op)))
#define K_BOTH(name, F, op) K_IND(name, F, op) K_DEP(name, F, op)
#define K64_BOTH(name, F, op) KERN(name##_ind, BODY64, R8(ROT8(F, op)), R8(ROT8(F, op))) KERN(name##_dep, BODY64, R8(DEP8(F, op)), R8(DEP8(F, op)))

// ---- VOP1 / VOP2, 4-byte encodings ----
K_BOTH(add, F2, "v_add_u32")
K_BOTH(sub, F2, "v_sub_u32")
K_BOTH(and_, F2, "v_and_b32")
K_BOTH(or_, F2, "v_or_b32")
K_BOTH(xor_, F2, "v_xor_b32")
K_BOTH(xnor, F2, "v_xnor_b32")
K_BOTH(lshl, F2S, "v_lshlrev_b32")
K_BOTH(lshr, F2S, "v_lshrrev_b32")
K_BOTH(ashr, F2S, "v_ashrrev_i32")
K_BOTH(minu, F2, "v_min_u32")
K_BOTH(mul24, F2, "v_mul_u32_u24")
K_BOTH(cndmask, F2V, "v_cndmask_b32")
K_BOTH(addco, F2CO, "v_add_co_u32")
K_BOTH(addci, F2CI, "v_addc_co_u32")
K_BOTH(mov, F1, "v_mov_b32")
K_BOTH(not_, F1, "v_not_b32")
K_BOTH(ffbhu, F1, "v_ffbh_u32")
K_BOTH(ffbhi, F1, "v_ffbh_i32")
K_BOTH(bfrev, F1, "v_bfrev_b32")
K_BOTH(addc5, F2C, "v_add_u32")
K_BOTH(addlit, F2L, "v_add_u32")
K_BOTH(andlit, F2L, "v_and_b32")
K_BOTH(cmp, FCMP, "v_cmp_lt_u32")
// ---- VOP3, 8-byte encodings ----
K_BOTH(add_e64, F2, "v_add_u32_e64")
K_BOTH(and_e64, F2, "v_and_b32_e64")
K_BOTH(cmp_s, FCMPS, "v_cmp_lt_u32_e64")
K_BOTH(add3, F3, "v_add3_u32")
K_BOTH(bfi, F3, "v_bfi_b32")
K_BOTH(alignbit, F3, "v_alignbit_b32")
K_BOTH(lshl_or, F3C, "v_lshl_or_b32")
K_BOTH(lshl_add, F3C, "v_lshl_add_u32")
K_BOTH(and_or, F3, "v_and_or_b32")
K_BOTH(or3, F3, "v_or3_b32")
K_BOTH(xad, F3, "v_xad_u32")
K_BOTH(bfe, F3, "v_bfe_u32")
K_BOTH(perm, F3, "v_perm_b32")
K_BOTH(mul_hi, F2, "v_mul_hi_u32")
K_BOTH(mul_lo, F2, "v_mul_lo_u32")
K_BOTH(mad24, F3, "v_mad_u32_u24")
K_BOTH(sad_u8, FSAD, "v_sad_u8")
K_BOTH(msad_u8, FSAD, "v_msad_u8")
// ---- SDWA / DPP ----
K_BOTH(sub_sdwa, FSDWA, "v_sub_u32_sdwa")
K_BOTH(mov_dpp, FDPP, "v_mov_b32_dpp")
K_BOTH(add_dpp, FDPP2, "v_add_u32_dpp")
// ---- packed 16-bit ----
K_BOTH(pk_add16, F2, "v_pk_add_u16")
K_BOTH(pk_lshl16, F2S, "v_pk_lshlrev_b16")
K_BOTH(pk_mul16, F2, "v_pk_mul_lo_u16")
// ---- 64-bit destinations ----
K64_BOTH(lshl64, F64S, "v_lshlrev_b64")
K64_BOTH(lshr64, F64S, "v_lshrrev_b64")
K64_BOTH(lshl_add64, F64A, "v_lshl_add_u64")
K64_BOTH(mad64, FMAD64, "v_mad_u64_u32")
K64_BOTH(mqsad, FQSAD, "v_mqsad_pk_u16_u8")
// ---- selects by a lane mask in an SGPR pair: where the mask comes from ----
#define FCMPCND(op, r) "v_cmp_lt_u32 vcc, %" #r ", %8\n\tv_cndmask_b32 %" #r ", %" #r ", %9, vcc\n\t"
#define FCMPSCND(op, r) "v_cmp_lt_u32 vcc, %" #r ", %8\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %" #r ", %" #r ", %9, vcc\n\t"
#define FCNDS(op, r) "v_cndmask_b32 %" #r ", %" #r ", %9, s[4:5]\n\t"
#define FBITOP(op, r) "v_bitop3_b32 %" #r ", %" #r ", %8, %9 bitop3:0x69\n\t"
K_BOTH(cmp_cnd, FCMPCND, "")
K_BOTH(cnd_sgpr, FCNDS, "")
K_BOTH(bitop3, FBITOP, "")
// ---- mixed streams in one wave ----
KERN(mix_add_bfi, BODY32, R8(MIX8(F2, "v_add_u32", F3, "v_bfi_b32")), R8(MIX8(F2, "v_add_u32", F3, "v_bfi_b32")))
KERN(mix_add_mulhi, BODY32, R8(MIX8(F2, "v_add_u32", F2, "v_mul_hi_u32")), R8(MIX8(F2, "v_add_u32", F2, "v_mul_hi_u32")))
KERN(mix_add_xor, BODY32, R8(MIX8(F2, "v_add_u32", F2, "v_xor_b32")), R8(MIX8(F2, "v_add_u32", F2, "v_xor_b32")))
KERN(mix_add_lshl, BODY32, R8(MIX8(F2, "v_add_u32", F2S, "v_lshlrev_b32")), R8(MIX8(F2, "v_add_u32", F2S, "v_lshlrev_b32")))
// ---- two different waves on a SIMD: A = dependent chain (the coder), B = something else ----
KERN(pair_depadd_depadd, BODY32, R8(DEP8(F2, "v_add_u32")), R8(DEP8(F2, "v_add_u32")))
KERN(pair_depbfi_depbfi, BODY32, R8(DEP8(F3, "v_bfi_b32")), R8(DEP8(F3, "v_bfi_b32")))
KERN(pair_depadd_depbfi, BODY32, R8(DEP8(F2, "v_add_u32")), R8(DEP8(F3, "v_bfi_b32")))
KERN(pair_depbfi_indadd, BODY32, R8(DEP8(F3, "v_bfi_b32")), R8(ROT8(F2, "v_add_u32")))
KERN(pair_depmulhi_indadd, BODY32, R8(DEP8(F2, "v_mul_hi_u32")), R8(ROT8(F2, "v_add_u32")))
KERN(pair_depxor_depxor, BODY32, R8(DEP8(F2, "v_xor_b32")), R8(DEP8(F2, "v_xor_b32")))
KERN(pair_depadd_depxor, BODY32, R8(DEP8(F2, "v_add_u32")), R8(DEP8(F2, "v_xor_b32")))
KERN(pair_depmulhi_depmulhi, BODY32, R8(DEP8(F2, "v_mul_hi_u32")), R8(DEP8(F2, "v_mul_hi_u32")))

typedef void (*kern_t)(uint64_t *, uint32_t, int);
struct Test
{
  const char *name;
  kern_t k;
  int split;
};
#define T2(label, name) {label " ind", name##_ind, 0}, {label " dep", name##_dep, 0}

int main(int argc, char **argv)
{
  setvbuf(stdout, NULL, _IONBF, 0);
  uint64_t *out;
  const int max_waves = 256 * 16;
  CHECK(hipMalloc(&out, max_waves * sizeof(uint64_t)));
  std::vector<Test> tests = {
      T2("v_add_u32 (VOP2)", add), T2("v_sub_u32 (VOP2)", sub), T2("v_and_b32 (VOP2)", and_), T2("v_or_b32 (VOP2)", or_), T2("v_xor_b32 (VOP2)", xor_),
      T2("v_xnor_b32 (VOP2)", xnor), T2("v_lshlrev_b32 (VOP2)", lshl), T2("v_lshrrev_b32 (VOP2)", lshr), T2("v_ashrrev_i32 (VOP2)", ashr),
      T2("v_min_u32 (VOP2)", minu), T2("v_mul_u32_u24 (VOP2)", mul24), T2("v_cndmask_b32 vcc (VOP2)", cndmask), T2("v_add_co_u32 (VOP2)", addco),
      T2("v_addc_co_u32 (VOP2)", addci), T2("v_mov_b32 (VOP1)", mov), T2("v_not_b32 (VOP1)", not_), T2("v_ffbh_u32 (VOP1)", ffbhu),
      T2("v_ffbh_i32 (VOP1)", ffbhi), T2("v_bfrev_b32 (VOP1)", bfrev), T2("v_add_u32 inline const", addc5), T2("v_add_u32 literal (8 B)", addlit),
      T2("v_and_b32 literal (8 B)", andlit), T2("v_cmp_lt_u32 -> vcc (VOPC)", cmp),
      T2("v_add_u32_e64 (VOP3)", add_e64), T2("v_and_b32_e64 (VOP3)", and_e64), T2("v_cmp_lt_u32_e64 -> sgpr", cmp_s), T2("v_add3_u32", add3),
      T2("v_bfi_b32", bfi), T2("v_alignbit_b32", alignbit), T2("v_lshl_or_b32", lshl_or), T2("v_lshl_add_u32", lshl_add), T2("v_and_or_b32", and_or),
      T2("v_or3_b32", or3), T2("v_xad_u32", xad), T2("v_bfe_u32", bfe), T2("v_perm_b32", perm), T2("v_mul_hi_u32", mul_hi), T2("v_mul_lo_u32", mul_lo),
      T2("v_mad_u32_u24", mad24), T2("v_sad_u8", sad_u8), T2("v_msad_u8", msad_u8),
      T2("v_sub_u32_sdwa", sub_sdwa), T2("v_mov_b32_dpp quad_perm", mov_dpp), T2("v_add_u32_dpp row_shr", add_dpp),
      T2("v_pk_add_u16", pk_add16), T2("v_pk_lshlrev_b16", pk_lshl16), T2("v_pk_mul_lo_u16", pk_mul16),
      T2("v_lshlrev_b64", lshl64), T2("v_lshrrev_b64", lshr64), T2("v_lshl_add_u64", lshl_add64), T2("v_mad_u64_u32", mad64), T2("v_mqsad_pk_u16_u8", mqsad),
      T2("v_cmp + v_cndmask vcc (2 instr)", cmp_cnd), T2("v_cndmask_b32 mask in s[4:5]", cnd_sgpr),
      T2("v_bitop3_b32", bitop3),
      {"mix add/bfi ind", mix_add_bfi, 0}, {"mix add/mul_hi ind", mix_add_mulhi, 0}, {"mix add/xor ind", mix_add_xor, 0}, {"mix add/lshl ind", mix_add_lshl, 0},
  };
  std::vector<Test> pairs = {
      {"A dep add | B dep add", pair_depadd_depadd, 1},       {"A dep bfi | B dep bfi", pair_depbfi_depbfi, 1},
      {"A dep add | B dep bfi", pair_depadd_depbfi, 1},       {"A dep bfi | B ind add", pair_depbfi_indadd, 1},
      {"A dep mul_hi | B ind add", pair_depmulhi_indadd, 1}, {"A dep xor | B dep xor", pair_depxor_depxor, 1},
      {"A dep add | B dep xor", pair_depadd_depxor, 1},       {"A dep mul_hi | B dep mul_hi", pair_depmulhi_depmulhi, 1},
  };
  const size_t lds = 96 * 1024; // one workgroup per CU
  std::vector<uint64_t> h(max_waves);
  printf("cycles per instruction per wave (s_memtime, mean over waves); 256 workgroups, one per CU\n");
  printf("%-34s %12s %12s %12s\n", "stream", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  for (auto &t : tests)
  {
    if (argc > 1 && strstr(t.name, argv[1]) == NULL) // tools/ubench2 <substring>: only the streams whose name holds it
      continue;
    CHECK(hipFuncSetAttribute((const void *)t.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double res[3];
    for (int m = 0; m < 3; m++)
    {
      const int threads = 256 << m;
      hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), lds, 0, out, 1u, 0);
      hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), lds, 0, out, 2u, 0);
      CHECK(hipDeviceSynchronize());
      const int waves = 256 * threads / 64;
      CHECK(hipMemcpy(h.data(), out, waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
      double s = 0;
      for (int w = 0; w < waves; w++)
        s += (double)h[w];
      res[m] = s / waves / ((double)ITERS * 64.0);
    }
    printf("%-34s %12.2f %12.2f %12.2f\n", t.name, res[0], res[1], res[2]);
  }
  printf("\ntwo different streams on one SIMD (waves w and w + n/2 of a workgroup): cycles per instruction, stream A | stream B\n");
  printf("%-34s %25s %25s\n", "pair", "2 waves/SIMD (1 A + 1 B)", "4 waves/SIMD (2 A + 2 B)");
  for (auto &t : pairs)
  {
    CHECK(hipFuncSetAttribute((const void *)t.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double ra[2], rb[2];
    for (int m = 1; m < 3; m++)
    {
      const int threads = 256 << m;
      hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), lds, 0, out, 1u, 1);
      hipLaunchKernelGGL(t.k, dim3(256), dim3(threads), lds, 0, out, 2u, 1);
      CHECK(hipDeviceSynchronize());
      const int wpb = threads / 64, waves = 256 * wpb;
      CHECK(hipMemcpy(h.data(), out, waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
      double sa = 0, sb = 0;
      for (int w = 0; w < waves; w++)
        ((w % wpb) < wpb / 2 ? sa : sb) += (double)h[w];
      ra[m - 1] = sa / (waves / 2) / ((double)ITERS * 64.0);
      rb[m - 1] = sb / (waves / 2) / ((double)ITERS * 64.0);
    }
    printf("%-34s %11.2f | %-11.2f %11.2f | %-11.2f\n", t.name, ra[0], rb[0], ra[1], rb[1]);
  }
  return 0;
}
