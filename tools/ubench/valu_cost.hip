// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU opcodes the
// macroblock kernel uses.  One wave per SIMD x 4 SIMDs x 256 CUs, 8 independent chains per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
#define ITERS 200

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a[8];
    unsigned long long q[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; q[i] = a[i]; }
    const uint32_t b = seed * 3 + threadIdx.x, c = seed ^ 0x55;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] + b;                                                          // v_add_u32
                if (OP == 1) a[i] = a[i] * b;                                                          // v_mul_lo_u32
                if (OP == 2) a[i] = (uint32_t)__mul24((int)a[i], (int)b);                              // v_mul_i32_i24
                if (OP == 3) a[i] = __builtin_amdgcn_sad_u8(a[i], b, a[i]);                            // v_sad_u8
                if (OP == 4) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(q[i], b, q[i]);                    // v_qsad_pk_u16_u8
                if (OP == 5) a[i] = __builtin_amdgcn_alignbyte(a[i], b, c);                            // v_alignbyte_b32
                if (OP == 6) a[i] += __builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xF, 0xF, true); // v_add_u32_dpp
                if (OP == 7) q[i] = (q[i] << 3) + b;                                                   // v_lshl_add_u64
                if (OP == 8) a[i] = a[i] > b ? a[i] - c : a[i] + c;                                    // cmp + cndmask/sub
                if (OP == 9) a[i] = (a[i] >> 3) & b;                                                   // shift + and (v_and_b32 / bfe)
                if (OP == 10) a[i] = __builtin_amdgcn_perm(a[i], b, 0x07050301);                       // v_perm_b32
                if (OP == 11) a[i] = (uint32_t)__builtin_amdgcn_sad_u16(a[i], b, a[i]);                // v_sad_u16
                if (OP == 12) a[i] = (uint32_t)((int)a[i] >> 5) + (a[i] & c);                          // ashr + and + add
                if (OP == 13) q[i] = q[i] << (c & 31);                                                 // v_lshlrev_b64
                if (OP == 14) a[i] = (uint32_t)__builtin_amdgcn_mqsad_pk_u16_u8(q[i], b, q[i]);        // v_mqsad_pk_u16_u8
                if (OP == 15) a[i] = __builtin_amdgcn_msad_u8(a[i], b, a[i]);                          // v_msad_u8
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)q[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint32_t)(t1 - t0);
}

template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4 * waves_per_simd;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t cyc; hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    const double instr = (double)ITERS * REP;
    printf("%-22s waves/SIMD %d: %7.2f clk/instr (wave-local s_memtime), wall %.3f ms -> %.2f ns per wave-instr per SIMD\n",
           name, waves_per_simd, cyc / instr, ms, ms * 1e6 / (instr * waves_per_simd));
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, ((1 << 20) + 16) * 4);
    for (int w : {1, 4}) {
        run<0>("v_add_u32", d, w);       run<1>("v_mul_lo_u32", d, w);   run<2>("v_mul_i32_i24", d, w);
        run<3>("v_sad_u8", d, w);        run<4>("v_qsad_pk_u16_u8", d, w); run<5>("v_alignbyte_b32", d, w);
        run<6>("v_add_u32_dpp", d, w);   run<7>("shl3+add u64", d, w);   run<8>("cmp+select+addsub", d, w);
        run<9>("shr+and", d, w);         run<10>("v_perm_b32", d, w);    run<11>("v_sad_u16", d, w);
        run<12>("ashr+and+add", d, w);   run<13>("v_lshlrev_b64", d, w); run<14>("v_mqsad_pk_u16_u8", d, w);
        run<15>("v_msad_u8", d, w);
    }
    return 0;
}
