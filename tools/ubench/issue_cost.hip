// issue_cost.hip - issue cost of the instructions k_mb is made of, on gfx950.
//
// Every test body is 32 INDEPENDENT instructions written with asm volatile (the compiler can neither fold nor
// reorder them), looped ITERS times by W waves per SIMD on every SIMD of the chip; the cost is
//     wall time x clock / (instructions per wave x W)        [shader cycles per wave-instruction per SIMD]
// with the clock taken from s_memtime ticks of the same launch (tick = shader cycle).  W = 1 shows the dependent /
// single-wave rate, W = 4 and 8 the rate the macroblock kernel sees with its 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_cost.hip -o /tmp/issue_cost && /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 2000
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

#define BODY_BEGIN(NAME)                                                                     \
    __global__ __launch_bounds__(1024) void NAME(uint32_t *out, uint32_t seed)                \
    {                                                                                        \
        __shared__ uint32_t lds[4096];                                                       \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 0x55, d0, d1, d2, d3;  \
        unsigned long long q = ((unsigned long long)a << 32) | b, e0, e1;                    \
        uint32_t la = (threadIdx.x * 4) & 0x3ffc;                                            \
        lds[threadIdx.x] = a; __syncthreads();                                               \
        d0 = d1 = d2 = d3 = 0; e0 = e1 = 0;                                                  \
        long long t0 = __builtin_readcyclecounter();                                         \
        for (int it = 0; it < ITERS; ++it) {
#define BODY_END                                                                             \
        }                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)");                                       \
        long long t1 = __builtin_readcyclecounter();                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (uint32_t)(t1 - t0); }           \
        if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long *)(out + 4), (unsigned long long)t0); atomicMax((unsigned long long *)(out + 6), (unsigned long long)t1); } \
        if (a == 0x12345678 && d0 + d1 + d2 + d3 + (uint32_t)e0 + (uint32_t)e1 == 77) out[1] = d0 + lds[5];  \
    }

// 4 different destinations, sources never written inside the loop: no dependency between any two instructions
// (one asm statement per loop body: between two asm statements the compiler's hazard recogniser inserts an s_nop)
#define V32(INS) asm volatile(R8(INS(%0) INS(%1) INS(%2) INS(%3)) : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b), "v"(c), "s"(seed), "v"(q), "v"(la) : "vcc", "s20", "s21");
#define T(NAME, INS) BODY_BEGIN(NAME) V32(INS) BODY_END

#define I_ADD(D)      "v_add_u32 " #D ", %4, %5\n\t"
#define I_AND(D)      "v_and_b32 " #D ", %4, %5\n\t"
#define I_MOV(D)      "v_mov_b32 " #D ", %4\n\t"
#define I_MOVK(D)     "v_mov_b32 " #D ", 0x12345\n\t"
#define I_LSHL(D)     "v_lshlrev_b32 " #D ", 3, %4\n\t"
#define I_SUBS(D)     "v_sub_u32 " #D ", %7, %4\n\t"
#define I_ADD3(D)     "v_add3_u32 " #D ", %4, %5, %6\n\t"
#define I_LSHLADD(D)  "v_lshl_add_u32 " #D ", %4, 3, %5\n\t"
#define I_LSHLOR(D)   "v_lshl_or_b32 " #D ", %4, 8, %5\n\t"
#define I_ANDOR(D)    "v_and_or_b32 " #D ", %4, %5, %6\n\t"
#define I_BFEU(D)     "v_bfe_u32 " #D ", %4, 8, 8\n\t"
#define I_BFEI(D)     "v_bfe_i32 " #D ", %4, 8, 9\n\t"
#define I_MAD24(D)    "v_mad_i32_i24 " #D ", %4, %5, %6\n\t"
#define I_MADU24(D)   "v_mad_u32_u24 " #D ", %4, %5, %6\n\t"
#define I_MUL24(D)    "v_mul_i32_i24 " #D ", %4, %5\n\t"
#define I_MULLO(D)    "v_mul_lo_u32 " #D ", %4, %5\n\t"
#define I_MULHI(D)    "v_mul_hi_u32 " #D ", %4, %5\n\t"
#define I_MED3(D)     "v_med3_i32 " #D ", %4, %5, %6\n\t"
#define I_MAXI(D)     "v_max_i32 " #D ", %4, %5\n\t"
#define I_PERM(D)     "v_perm_b32 " #D ", %4, %5, %6\n\t"
#define I_ALIGN(D)    "v_alignbyte_b32 " #D ", %4, %5, 1\n\t"
#define I_ALIGNV(D)   "v_alignbyte_b32 " #D ", %4, %5, %6\n\t"
#define I_LERP(D)     "v_lerp_u8 " #D ", %4, %5, %6\n\t"
#define I_SAD(D)      "v_sad_u8 " #D ", %4, %5, 0\n\t"
#define I_SADA(D)     "v_sad_u8 " #D ", %4, %5, %6\n\t"
#define I_CNDV(D)     "v_cndmask_b32 " #D ", %4, %5, vcc\n\t"
#define I_CNDS(D)     "v_cndmask_b32 " #D ", %4, %5, s[20:21]\n\t"
#define I_CNDV64(D)   "v_cndmask_b32_e64 " #D ", %4, %5, vcc\n\t"
#define I_CNDVK(D)    "v_cndmask_b32 " #D ", 0, %5, vcc\n\t"
#define I_CMPCND(D)   "v_cmp_lt_u32 vcc, %4, %5\n\tv_cndmask_b32 " #D ", %4, %5, vcc\n\t"
#define I_CMPCNDS(D)  "v_cmp_lt_u32 s[20:21], %4, %5\n\tv_cndmask_b32 " #D ", %4, %5, s[20:21]\n\t"
#define I_CMPV(D)     "v_cmp_lt_u32 vcc, %4, %5\n\t"
#define I_CMPS(D)     "v_cmp_lt_u32 s[20:21], %4, %5\n\t"
#define I_DOT2(D)     "v_dot2_i32_i16 " #D ", %4, %5, %6\n\t"
#define I_DOT2C(D)    "v_dot2c_i32_i16 " #D ", %4, %5\n\t"
#define I_DOT4(D)     "v_dot4_i32_i8 " #D ", %4, %5, %6\n\t"
#define I_DOT4C(D)    "v_dot4c_i32_i8 " #D ", %4, %5\n\t"
#define I_PKADD(D)    "v_pk_add_i16 " #D ", %4, %5\n\t"
#define I_PKSUB(D)    "v_pk_sub_i16 " #D ", %4, %5\n\t"
#define I_PKMAX(D)    "v_pk_max_i16 " #D ", %4, %5\n\t"
#define I_PKMAD(D)    "v_pk_mad_i16 " #D ", %4, %5, %6\n\t"
#define I_PKMUL(D)    "v_pk_mul_lo_u16 " #D ", %4, %5\n\t"
#define I_PKASHR(D)   "v_pk_ashrrev_i16 " #D ", 3, %4\n\t"
#define I_SDWA(D)     "v_add_u32_sdwa " #D ", %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"
#define I_DPPADD(D)   "v_add_u32_dpp " #D ", %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define I_DPPMOV(D)   "v_mov_b32_dpp " #D ", %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define I_DPPBC(D)    "v_add_u32_dpp " #D ", %4, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
#define I_MBCNT(D)    "v_mbcnt_lo_u32_b32 " #D ", %7, %4\n\t"
#define I_FFBH(D)     "v_ffbh_u32 " #D ", %4\n\t"
#define I_BCNT(D)     "v_bcnt_u32_b32 " #D ", %4, %5\n\t"
#define I_XAD(D)      "v_xad_u32 " #D ", %4, %5, %6\n\t"
#define I_ADDLSHL(D)  "v_add_lshl_u32 " #D ", %4, %5, 2\n\t"
#define I_SADHI(D)    "v_sad_hi_u8 " #D ", %4, %5, %6\n\t"
#define I_BITOP3(D)   "v_bitop3_b32 " #D ", %4, %5, %6 bitop3:0xde\n\t"
// round 4: which of the cheap-looking integer forms share the ~2-cycle rate of v_add_u32 / v_and_b32 / v_mov_b32 / v_bitop3_b32?
#define I_SUBV(D)     "v_sub_u32 " #D ", %4, %5\n\t"
#define I_SUBREVV(D)  "v_subrev_u32 " #D ", %4, %5\n\t"
#define I_OR(D)       "v_or_b32 " #D ", %4, %5\n\t"
#define I_XOR(D)      "v_xor_b32 " #D ", %4, %5\n\t"
#define I_XNOR(D)     "v_xnor_b32 " #D ", %4, %5\n\t"
#define I_NOT(D)      "v_not_b32 " #D ", %4\n\t"
#define I_ADDK(D)     "v_add_u32 " #D ", 5, %4\n\t"
#define I_ADDLIT(D)   "v_add_u32 " #D ", 0x808080, %4\n\t"
#define I_ADDS(D)     "v_add_u32 " #D ", %7, %4\n\t"
#define I_ANDLIT(D)   "v_and_b32 " #D ", 0xff00ff, %4\n\t"
#define I_ASHR(D)     "v_ashrrev_i32 " #D ", 3, %4\n\t"
#define I_LSHR(D)     "v_lshrrev_b32 " #D ", 3, %4\n\t"
#define I_MINU(D)     "v_min_u32 " #D ", %4, %5\n\t"
#define I_ADDCO(D)    "v_add_co_u32 " #D ", vcc, %4, %5\n\t"
#define I_ADDE64(D)   "v_add_u32_e64 " #D ", %4, %5\n\t"
#define I_ADDSDWA(D)  "v_add_u32_sdwa " #D ", %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define I_OR3(D)      "v_or3_b32 " #D ", %4, %5, %6\n\t"
#define I_XOR3(D)     "v_bitop3_b32 " #D ", %4, %5, %6 bitop3:0x96\n\t"
#define I_CVTF(D)     "v_cvt_f32_i32 " #D ", %4\n\t"
#define I_ADDF(D)     "v_add_f32 " #D ", %4, %5\n\t"
#define I_FMA(D)      "v_fma_f32 " #D ", %4, %5, %6\n\t"
#define I_MULF(D)     "v_mul_f32 " #D ", %4, %5\n\t"
#define I_PKADDF(D)   "v_pk_add_f32 " #D ", %8, %8\n\t"
#define I_DSR32(D)    "ds_read_b32 " #D ", %9\n\t"
#define I_DSW32(D)    "ds_write_b32 %9, %4\n\t"
#define I_DSW16(D)    "ds_write_b16 %9, %4\n\t"
#define I_DSOR(D)     "ds_or_b32 %9, %4\n\t"

T(k_add, I_ADD) T(k_and, I_AND) T(k_mov, I_MOV) T(k_movk, I_MOVK) T(k_lshl, I_LSHL) T(k_subs, I_SUBS) T(k_add3, I_ADD3)
T(k_lshladd, I_LSHLADD) T(k_lshlor, I_LSHLOR) T(k_andor, I_ANDOR) T(k_bfeu, I_BFEU) T(k_bfei, I_BFEI) T(k_mad24, I_MAD24)
T(k_madu24, I_MADU24) T(k_mul24, I_MUL24) T(k_mullo, I_MULLO) T(k_mulhi, I_MULHI) T(k_med3, I_MED3) T(k_maxi, I_MAXI)
T(k_perm, I_PERM) T(k_align, I_ALIGN) T(k_alignv, I_ALIGNV) T(k_lerp, I_LERP) T(k_sad, I_SAD) T(k_sada, I_SADA)
T(k_cndv, I_CNDV) T(k_cnds, I_CNDS) T(k_cndv64, I_CNDV64) T(k_cndvk, I_CNDVK) T(k_cmpcnd, I_CMPCND) T(k_cmpcnds, I_CMPCNDS) T(k_cmpv, I_CMPV) T(k_cmps, I_CMPS) T(k_dot2, I_DOT2) T(k_dot2c, I_DOT2C)
T(k_dot4, I_DOT4) T(k_dot4c, I_DOT4C) T(k_pkadd, I_PKADD) T(k_pksub, I_PKSUB) T(k_pkmax, I_PKMAX) T(k_pkmad, I_PKMAD)
T(k_pkmul, I_PKMUL) T(k_pkashr, I_PKASHR) T(k_sdwa, I_SDWA) T(k_dppadd, I_DPPADD) T(k_dppmov, I_DPPMOV) T(k_dppbc, I_DPPBC)
T(k_mbcnt, I_MBCNT) T(k_ffbh, I_FFBH) T(k_bcnt, I_BCNT) T(k_xad, I_XAD) T(k_addlshl, I_ADDLSHL)
T(k_sadhi, I_SADHI) T(k_bitop3, I_BITOP3)
T(k_subv, I_SUBV) T(k_subrevv, I_SUBREVV) T(k_or, I_OR) T(k_xor, I_XOR) T(k_xnor, I_XNOR) T(k_not, I_NOT) T(k_addk, I_ADDK) T(k_addlit, I_ADDLIT)
T(k_adds, I_ADDS) T(k_andlit, I_ANDLIT) T(k_ashr, I_ASHR) T(k_lshr, I_LSHR) T(k_minu, I_MINU) T(k_addco, I_ADDCO) T(k_adde64, I_ADDE64)
T(k_addsdwa, I_ADDSDWA) T(k_or3, I_OR3) T(k_xor3, I_XOR3) T(k_cvtf, I_CVTF) T(k_addf, I_ADDF) T(k_fma, I_FMA) T(k_mulf, I_MULF)
T(k_dsr32, I_DSR32) T(k_dsw32, I_DSW32) T(k_dsw16, I_DSW16) T(k_dsor, I_DSOR)

// 64-bit destinations
#define Q32(INS) asm volatile(R8(INS(%0) INS(%1) INS(%0) INS(%1)) : "=v"(e0), "=v"(e1) : "v"(a), "v"(b), "v"(q), "v"(la) : "vcc");
#define TQ(NAME, INS) BODY_BEGIN(NAME) Q32(INS) BODY_END
#define I_QSAD(D)     "v_qsad_pk_u16_u8 " #D ", %4, %2, 0\n\t"
#define I_MQSAD(D)    "v_mqsad_pk_u16_u8 " #D ", %4, %2, 0\n\t"
#define I_LSHL64(D)   "v_lshlrev_b64 " #D ", 3, %4\n\t"
#define I_LSHLADD64(D) "v_lshl_add_u64 " #D ", %4, 3, %4\n\t"
#define I_DSR64(D)    "ds_read_b64 " #D ", %5\n\t"
#define I_MADU64(D)   "v_mad_u64_u32 " #D ", vcc, %2, %3, %4\n\t"
TQ(k_qsad, I_QSAD) TQ(k_mqsad, I_MQSAD) TQ(k_lshl64, I_LSHL64) TQ(k_lshladd64, I_LSHLADD64) TQ(k_dsr64, I_DSR64) TQ(k_madu64, I_MADU64)

// lane swaps (gfx950): both operands are written
BODY_BEGIN(k_swap32) asm volatile(R8("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); BODY_END
BODY_BEGIN(k_swap16) asm volatile(R8("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); BODY_END
// readlane: scalar destination
BODY_BEGIN(k_readlane) asm volatile(R32("v_readlane_b32 s20, %0, 5\n\t") :: "v"(a) : "s20"); BODY_END
BODY_BEGIN(k_readfirst) asm volatile(R32("v_readfirstlane_b32 s20, %0\n\t") :: "v"(a) : "s20"); BODY_END
BODY_BEGIN(k_salu) asm volatile(R32("s_add_u32 s20, s21, s22\n\t") ::: "s20", "scc"); BODY_END
// a VALU and a SALU instruction alternating: does the scalar unit issue beside the vector unit?
BODY_BEGIN(k_valu_salu) asm volatile(R32("v_add_u32 %0, %1, %2\n\ts_add_u32 s20, s21, s22\n\t") : "=v"(d0) : "v"(a), "v"(b) : "s20", "scc"); BODY_END
// a VALU and an LDS read alternating
BODY_BEGIN(k_valu_dsr) asm volatile(R32("v_add_u32 %0, %2, %3\n\tds_read_b32 %1, %4\n\t") : "=v"(d0), "=v"(d1) : "v"(a), "v"(b), "v"(la)); BODY_END

// MFMA issue rate (independent accumulators)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(1024) void k_mfma16x16x32(uint32_t *out, uint32_t seed)
{
    long a = seed + threadIdx.x, b = seed * 7;
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
    if (c0[0] + c1[1] + c2[2] + c3[3] == 0x7fffffff) out[1] = 1;
}
// MFMA and VALU interleaved: 1 MFMA + 7 v_add per group; does the matrix pipe hide behind the VALU stream?
__global__ __launch_bounds__(1024) void k_mfma_valu(uint32_t *out, uint32_t seed)
{
    long a = seed + threadIdx.x, b = seed * 7;
    uint32_t x = seed + threadIdx.x, y = seed, d0 = 0;
    v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 1; ++r) {
#define VA7 asm volatile("v_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\tv_add_u32 %0, %1, %2\n\t" : "=v"(d0) : "v"(x), "v"(y));
            c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c0, 0, 0, 0); VA7
            c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c1, 0, 0, 0); VA7
            c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c2, 0, 0, 0); VA7
            c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c3, 0, 0, 0); VA7
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (uint32_t)(t1 - t0);
    if (c0[0] + c1[1] + c2[2] + c3[3] + d0 == 0x7fffffff) out[1] = 1;
}

struct Test { const char *name; void (*fn)(uint32_t *, uint32_t); int per_iter; };

int main()
{
    uint32_t *out;
    hipMalloc(&out, 64);
    hipMemset(out, 0, 64);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<Test> tests = {
#define E(k, n) {#k, k, n}
        E(k_add, 32), E(k_and, 32), E(k_mov, 32), E(k_movk, 32), E(k_lshl, 32), E(k_subs, 32), E(k_add3, 32), E(k_lshladd, 32),
        E(k_lshlor, 32), E(k_andor, 32), E(k_bfeu, 32), E(k_bfei, 32), E(k_mad24, 32), E(k_madu24, 32), E(k_mul24, 32), E(k_mullo, 32),
        E(k_mulhi, 32), E(k_med3, 32), E(k_maxi, 32), E(k_perm, 32), E(k_align, 32), E(k_alignv, 32), E(k_lerp, 32), E(k_sad, 32),
        E(k_sada, 32), E(k_cndv, 32), E(k_cnds, 32), E(k_cndv64, 32), E(k_cndvk, 32), E(k_cmpcnd, 64), E(k_cmpcnds, 64), E(k_cmpv, 32), E(k_cmps, 32), E(k_dot2, 32), E(k_dot2c, 32), E(k_dot4, 32),
        E(k_dot4c, 32), E(k_pkadd, 32), E(k_pksub, 32), E(k_pkmax, 32), E(k_pkmad, 32), E(k_pkmul, 32), E(k_pkashr, 32), E(k_sdwa, 32),
        E(k_dppadd, 32), E(k_dppmov, 32), E(k_dppbc, 32), E(k_mbcnt, 32), E(k_ffbh, 32), E(k_bcnt, 32), E(k_xad, 32), E(k_addlshl, 32),
        E(k_sadhi, 32), E(k_bitop3, 32),
        E(k_subv, 32), E(k_subrevv, 32), E(k_or, 32), E(k_xor, 32), E(k_xnor, 32), E(k_not, 32), E(k_addk, 32), E(k_addlit, 32), E(k_adds, 32), E(k_andlit, 32),
        E(k_ashr, 32), E(k_lshr, 32), E(k_minu, 32), E(k_addco, 32), E(k_adde64, 32), E(k_addsdwa, 32), E(k_or3, 32), E(k_xor3, 32), E(k_cvtf, 32),
        E(k_addf, 32), E(k_fma, 32), E(k_mulf, 32),
        E(k_swap32, 32), E(k_swap16, 32), E(k_qsad, 32), E(k_mqsad, 32), E(k_lshl64, 32), E(k_lshladd64, 32), E(k_madu64, 32),
        E(k_readlane, 32), E(k_readfirst, 32), E(k_salu, 32), E(k_valu_salu, 64), E(k_valu_dsr, 64),
        E(k_dsr32, 32), E(k_dsr64, 32), E(k_dsw32, 32), E(k_dsw16, 32), E(k_dsor, 32),
        E(k_mfma16x16x32, 32), E(k_mfma_valu, 32),
    };
    printf("%d CUs; cost = shader cycles per wave-instruction per SIMD (wall time x measured clock / instructions), W waves per SIMD\n", cus);
    printf("%-16s %8s %8s %8s %10s %8s  %s\n", "test", "W=1", "W=4", "W=8(wall@2.4GHz)", "W=8(ticks)", "MHz", "(W=8 ticks: first start to last end of all wavefronts in s_memtime ticks - clock independent; MHz = ticks / wall)");
    for (auto &t : tests) {
        double cost[3], cost_ticks = 0, mhz = 0;
        int wi = 0;
        for (int W : {1, 4, 8}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            // W = 1: one 256-thread block per CU (one wave per SIMD); W = 4: one 1024-thread block per CU (its 16 waves are
            // co-resident by construction, 4 per SIMD); W = 8: two of those per CU
            const int threads = W == 1 ? 256 : 1024;
            const int blocks = cus * (W == 8 ? 2 : 1);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u);     // warm
            hipDeviceSynchronize();
            float best = 1e9f;
            uint32_t ticks = 0;
            unsigned long long span = 0;
            for (int rep = 0; rep < 5; ++rep) {
                const unsigned long long init[2] = {~0ull, 0ull};
                hipMemcpy(out + 4, init, 16, hipMemcpyHostToDevice);
                hipEventRecord(e0);
                hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) {
                    best = ms; hipMemcpy(&ticks, out, 4, hipMemcpyDeviceToHost);
                    unsigned long long mm[2];
                    hipMemcpy(mm, out + 4, 16, hipMemcpyDeviceToHost);
                    span = mm[1] > mm[0] ? mm[1] - mm[0] : 0;
                }
            }
            // cycles per instruction from the wave-local tick count of block 0 (tick = shader cycle): all W waves of a
            // SIMD run concurrently for `ticks`, issuing W * per_iter * ITERS instructions between them
            const int per = !strcmp(t.name, "k_mfma_valu") ? 32 : t.per_iter;
            // W = 8 (two blocks per CU): the blocks may not overlap completely; the wall-clock figure is the honest one there
            const double wall_cycles = best * 1e-3 * 2.4e9;
            cost[wi++] = W == 8 ? wall_cycles / ((double)per * ITERS * W) : (double)ticks / ((double)per * ITERS * W);
            if (W == 8) { cost_ticks = (double)span / ((double)per * ITERS * W); mhz = (double)span / (best * 1e-3) * 1e-6; }
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        printf("%-16s %8.2f %8.2f %8.2f %10.2f %8.0f\n", t.name, cost[0], cost[1], cost[2], cost_ticks, mhz);
    }
    return 0;
}
