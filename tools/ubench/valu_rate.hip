// Microbenchmark: issue rate of wave64 integer VALU instructions on gfx950 (MI355X), per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate > profiles/r04/valu_rate.txt
// Question it settles ("What's weak" 1): does a wave64 integer instruction hold its SIMD for FOUR cycles
// (16 lanes a cycle; the packed-fp32 figure of 157 TFLOP/s is then 2 x 16 lanes) or for TWO (SIMD-32, as
// /opt/skills/guides/MI355X_MICROARCH.md states for v_fma_f32)?  Every op is timed as K independent chains per wave
// (K = 8: no dependent-issue stalls) at 1, 2, 4 and 8 waves per SIMD, plus one dependent chain (latency), plus VALU
// mixed with SALU in the same wave and in neighbouring waves (does scalar issue overlap vector issue?).
// Cycles are s_memtime deltas (shader clock) taken by every wave; the slowest wave of the launch is reported, and
// the wall time by HIP events gives the clock the counter ran at.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Op {
  ADD_U32, AND_B32, LSHLREV_B32, ALIGNBYTE, PERM, BFE_U32, LSHLREV_B64, ADD_DPP, PK_ADD_U16, DOT4_U32_U8, SAD_U8,
  MUL_LO_U32, MAD_U32_U24, BCNT, FFBL, CNDMASK, CMP_CNDMASK, ADD3, LSHL_OR, AND_OR, BFI, MBCNT, ADDC_PAIR, XOR3_LOP3,
  PK_LSHL_U16, MOV_DPP_BCAST, READLANE_PAIR, SALU_ONLY, MIX_VALU_SALU, FMA_F32, PK_FMA_F32,
  OR_B32, XOR_B32, SUB_U32, MOV_B32, NOT_B32, MAX_U32, MIN_U32, LSHRREV_B32, ASHRREV_I32, AND_E64, ADD_LIT, CMP_ONLY, CNDMASK_SGPR, BFREV,
  MUL_U32_U24, LSHL_ADD, ADD_LSHL, OR3, XAD, MIN3, MED3, ALIGNBIT, BFM, SUBREV, MOV_SDWA, CVT_PK_U8, SAT_PK, XNOR, ADD_SDWA, PK_MAX_U16, PK_MAD_U16, MQSAD, MSAD, LERP, MBCNT_HI, READFIRSTLANE_ADD, DS_BPERMUTE, DS_SWIZZLE, PERMLANE32_SWAP, BITOP3, LSHL_ADD_U64, MIX_ADD_BFE, MIX_2ADD_BFE, MIX_3ADD_BFE, MIX_ADD_LDSREAD, OP_COUNT
};
static const char* op_name[OP_COUNT] = {
  "v_add_u32", "v_and_b32", "v_lshlrev_b32", "v_alignbyte_b32", "v_perm_b32", "v_bfe_u32", "v_lshlrev_b64",
  "v_add_u32 dpp row_shr:1", "v_pk_add_u16", "v_dot4_u32_u8", "v_sad_u8", "v_mul_lo_u32", "v_mad_u32_u24",
  "v_bcnt_u32_b32", "v_ffbl_b32", "v_cndmask_b32 (vcc fixed)", "v_cmp_lt_u32 + v_cndmask_b32 (2 instr)", "v_add3_u32",
  "v_lshl_or_b32", "v_and_or_b32", "v_bfi_b32", "v_mbcnt_lo_u32_b32", "v_add_co_u32 + v_addc_co_u32 (2 instr)",
  "v_xor_b32 + v_and_b32 (2 instr, dependent pair)", "v_pk_lshlrev_b16", "v_mov_b32 dpp row_bcast:15",
  "v_readlane_b32 + v_add (2 instr)", "s_add_u32 only (8 chains)", "v_add_u32 + s_add_u32 interleaved (2 instr)",
  "v_fma_f32", "v_pk_fma_f32",
  "v_or_b32", "v_xor_b32", "v_sub_u32", "v_mov_b32 (chain of 2 regs)", "v_not_b32", "v_max_u32", "v_min_u32", "v_lshrrev_b32", "v_ashrrev_i32",
  "v_and_b32_e64 (VOP3 encoding)", "v_add_u32 literal", "v_cmp_lt_u32 only (sgpr pair dst)", "v_cndmask_b32 (sgpr mask)", "v_bfrev_b32",
  "v_mul_u32_u24", "v_lshl_add_u32", "v_add_lshl_u32", "v_or3_b32", "v_xad_u32", "v_min3_u32", "v_med3_u32", "v_alignbit_b32", "v_bfm_b32",
  "v_subrev_u32", "v_mov_b32 sdwa byte1", "v_cvt_pk_u8_f32", "v_sat_pk_u8_i16", "v_xnor_b32", "v_add_u32 sdwa", "v_pk_max_u16", "v_pk_mad_u16",
  "v_mqsad_pk_u16_u8", "v_msad_u8", "v_lerp_u8", "v_mbcnt_hi_u32_b32", "v_readfirstlane_b32 + v_add (2 instr)", "ds_bpermute_b32 (+waitcnt per 8)",
  "ds_swizzle_b32 (+waitcnt per 8)", "v_permlane32_swap", "v_bitop3_b32", "v_lshl_add_u64",
  "v_add_u32 + v_bfe_u32 (2 instr: simple + complex)", "2 x v_add_u32 + v_bfe_u32 (3 instr)", "3 x v_add_u32 + v_bfe_u32 (4 instr)",
  "v_add_u32 + ds_read_b32 (2 instr, waitcnt per 8)"};
// instructions one unrolled step issues per chain
static int op_instrs(int op) {
  switch (op) { case CMP_CNDMASK: case ADDC_PAIR: case XOR3_LOP3: case READLANE_PAIR: case MIX_VALU_SALU: case MOV_B32: case READFIRSTLANE_ADD: case MIX_ADD_BFE: case MIX_ADD_LDSREAD: return 2; case MIX_2ADD_BFE: return 3; case MIX_3ADD_BFE: return 4; default: return 1; }
}

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k_rate(int iters, unsigned long long* cycles, uint32_t* sink) {
  uint32_t a[8], b = threadIdx.x * 2654435761u + 12345u, c = threadIdx.x | 1u;
  unsigned long long w[8];
  uint32_t s[8], s2[8];
  unsigned long long cm[8];
  const unsigned long long cm0 = 0x5555aaaa5555aaaaull;
  unsigned long long cm0v = cm0 + threadIdx.x;
  __shared__ uint32_t ldsbuf[256];
  ldsbuf[threadIdx.x] = threadIdx.x;
  const uint32_t ldsaddr = (threadIdx.x & 63) * 4;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pf[8];
  float ff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i * 77u; w[i] = a[i]; s[i] = blockIdx.x + i; s2[i] = a[i] ^ 5u; cm[i] = 0; ff[i] = a[i]; pf[i] = f2{ff[i], ff[i]}; }
  const float fb = 1.0001f, fc = 0.5f;
  const f2 pb = {fb, fb}, pc = {fc, fc};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) {
        if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
        if (OP == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(b));
        if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 17" : "+v"(a[i]));
        if (OP == LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(w[i]));
        if (OP == ADD_DPP) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == DOT4_U32_U8) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == SAD_U8) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
        if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
        if (OP == CMP_CNDMASK) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
        if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(b));
        if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == ADDC_PAIR) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a[i]), "+v"(s[i]) : "v"(b), "v"(c) : "vcc");
        if (OP == XOR3_LOP3) asm volatile("v_xor_b32 %0, %0, %1\n v_and_b32 %0, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == PK_LSHL_U16) asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(a[i]));
        if (OP == MOV_DPP_BCAST) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
        if (OP == READLANE_PAIR) { uint32_t t; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(a[i])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "s"(t)); }
        if (OP == SALU_ONLY) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s[i]) :: "scc");
        if (OP == MIX_VALU_SALU) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("s_add_u32 %0, %0, 7" : "+s"(s[i]) :: "scc"); }
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff[i]) : "v"(fb), "v"(fc));
        if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pf[i]) : "v"(pb), "v"(pc));
        if (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %0" : "+v"(a[i]), "+v"(s2[i]));
        if (OP == NOT_B32) asm volatile("v_not_b32 %0, %0" : "+v"(a[i]));
        if (OP == MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == LSHRREV_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
        if (OP == ASHRREV_I32) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i]));
        if (OP == AND_E64) asm volatile("v_and_b32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD_LIT) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a[i]));
        if (OP == CMP_ONLY) asm volatile("v_cmp_lt_u32 %0, %1, %2" : "=s"(cm[i]) : "v"(a[i]), "v"(b));
        if (OP == CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(cm0));
        if (OP == BFREV) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
        if (OP == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD_LSHL) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(a[i]) : "v"(b));
        if (OP == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MIN3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MED3) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 5" : "+v"(a[i]) : "v"(b));
        if (OP == BFM) asm volatile("v_bfm_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == SUBREV) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == MOV_SDWA) asm volatile("v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(a[i]));
        if (OP == CVT_PK_U8) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(fb));
        if (OP == SAT_PK) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(a[i]));
        if (OP == XNOR) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
        if (OP == PK_MAX_U16) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == PK_MAD_U16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MQSAD) asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0" : "+v"(w[i]) : "v"(b));
        if (OP == MSAD) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == LERP) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == MBCNT_HI) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
        if (OP == READFIRSTLANE_ADD) { uint32_t t; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t) : "v"(a[i])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "s"(t)); }
        if (OP == DS_BPERMUTE) { asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b)); if (i == CHAINS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == DS_SWIZZLE) { asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F" : "+v"(a[i])); if (i == CHAINS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(cm0v));
        if (OP == MIX_ADD_BFE) asm volatile("v_add_u32 %0, %0, %2\n v_bfe_u32 %1, %1, 3, 17" : "+v"(a[i]), "+v"(s2[i]) : "v"(b));
        if (OP == MIX_2ADD_BFE) asm volatile("v_add_u32 %0, %0, %2\n v_bfe_u32 %1, %1, 3, 17\n v_xor_b32 %0, %0, %2" : "+v"(a[i]), "+v"(s2[i]) : "v"(b));
        if (OP == MIX_3ADD_BFE) asm volatile("v_add_u32 %0, %0, %2\n v_bfe_u32 %1, %1, 3, 17\n v_xor_b32 %0, %0, %2\n v_sub_u32 %0, %0, %2" : "+v"(a[i]), "+v"(s2[i]) : "v"(b));
        if (OP == MIX_ADD_LDSREAD) { asm volatile("v_add_u32 %0, %0, %2\n ds_read_b32 %1, %3" : "+v"(a[i]), "=v"(s2[i]) : "v"(b), "v"(ldsaddr)); if (i == CHAINS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(s2[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i] + (uint32_t)w[i] + s[i] + s2[i] + (uint32_t)cm[i] + (uint32_t)ff[i] + (uint32_t)pf[i].x;
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

// half the waves of every workgroup run VALU, the other half SALU: does the scalar stream of one wave slow the vector
// stream of its SIMD neighbour?  (waves 0..3 of a 512-thread workgroup land on SIMDs 0..3, waves 4..7 again.)
__global__ void __launch_bounds__(512) k_split_roles(int iters, unsigned long long* cycles, uint32_t* sink) {
  uint32_t a[8], s[8], b = threadIdx.x * 2654435761u;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; s[i] = blockIdx.x + i; }
  const bool scalar_role = (threadIdx.x >> 6) >= 4;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (scalar_role) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, 7" : "+s"(s[i]) :: "scc");
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i] + s[i];
  if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

typedef void (*kern_t)(int, unsigned long long*, uint32_t*);
template <int OP> static kern_t pick(int chains) { return chains == 1 ? (kern_t)k_rate<OP, 1> : (kern_t)k_rate<OP, 8>; }
template <int... OPS> static void fill(kern_t (*tab[])(int), std::integer_sequence<int, OPS...>) { ((tab[OPS] = pick<OPS>), ...); }

int main(int argc, char** argv) {
  int dev = 0;
  CHK(hipSetDevice(dev));
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, dev));
  const int cus = p.multiProcessorCount;
  printf("# device %s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1000.0);
  printf("# cycles = s_memtime ticks; 'wave-instr/clk/SIMD' = (waves per SIMD x instructions per wave) / ticks of the slowest wave\n");
  kern_t (*tab[OP_COUNT])(int);
  fill(tab, std::make_integer_sequence<int, OP_COUNT>{});
  unsigned long long* d_cycles;
  uint32_t* d_sink;
  const int max_waves = cus * 32 * 2;
  CHK(hipMalloc(&d_cycles, max_waves * sizeof(unsigned long long)));
  CHK(hipMalloc(&d_sink, 64));
  std::vector<unsigned long long> h(max_waves);
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int iters = 2000;
  // counter frequency: a long v_add run, ticks against wall time
  double tick_ghz = 0;
  printf("\n%-46s %6s %5s %12s %12s %10s %10s\n", "op", "chains", "w/SIMD", "ticks/instr", "w-instr/clk/SIMD", "wall ms", "ticks/ns");
  for (int op = 0; op < OP_COUNT; ++op) {
    for (int chains : {8, 1}) {
      for (int wps : {1, 2, 4, 8}) {
        if (chains == 1 && wps != 1) continue;
        // one 256-thread workgroup = one wave per SIMD; wps workgroups per CU
        const int blocks = cus * wps;
        kern_t kf = tab[op](chains);
        hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, 10, d_cycles, d_sink);  // warm-up
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, iters, d_cycles, d_sink);
        CHK(hipEventRecord(e1));
        CHK(hipDeviceSynchronize());
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h.data(), d_cycles, blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long mx = 0, mn = ~0ull;
        for (int i = 0; i < blocks * 4; ++i) { mx = std::max(mx, h[i]); mn = std::min(mn, h[i]); }
        const double instr = (double)iters * 16 * chains * op_instrs(op);
        const double per = mx / instr;
        if (op == ADD_U32 && chains == 8 && wps == 8) tick_ghz = mx / (ms * 1e6);
        printf("%-46s %6d %5d %12.3f %12.3f %10.3f %10.3f\n", op_name[op], chains, wps, per, wps * instr / mx, ms, mx / (ms * 1e6));
      }
    }
  }
  // SALU beside VALU in neighbouring waves
  {
    const int blocks = cus;
    hipLaunchKernelGGL(k_split_roles, dim3(blocks), dim3(512), 0, 0, 10, d_cycles, d_sink);
    CHK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_split_roles, dim3(blocks), dim3(512), 0, 0, iters, d_cycles, d_sink);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(h.data(), d_cycles, blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long mv = 0, ms_ = 0;
    for (int i = 0; i < blocks * 8; ++i) { if ((i & 7) < 4) mv = std::max(mv, h[i]); else ms_ = std::max(ms_, h[i]); }
    const double instr = (double)iters * 16 * 8;
    printf("\n# one VALU wave and one SALU wave per SIMD (512-thread workgroup, one per CU):\n");
    printf("#   v_add_u32 wave: %.3f ticks/instr;  s_add_u32 wave: %.3f ticks/instr (alone: see rows above at 1 wave per SIMD)\n", mv / instr, ms_ / instr);
  }
  printf("\n# s_memtime ran at %.3f ticks/ns during the 8-waves-per-SIMD v_add_u32 run (the shader clock, if ticks are cycles)\n", tick_ghz);
  return 0;
}
