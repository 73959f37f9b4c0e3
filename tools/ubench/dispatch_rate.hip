// dispatch_rate.hip - what it costs to START the wavefronts of k_mb's launch geometry on gfx950, with no work in them.
//
// k_mb launches one 64-thread workgroup (one wavefront) per macroblock: 86 400 workgroups per P-frame launch of config c3,
// 64 VGPRs and 4 288 bytes of LDS each.  This measures the time of launches of that shape whose wavefronts end at once
// (variant 0), or idle for N shader cycles first (s_sleep: occupies the wave slot, no issue), for 1, 2 and 4 wavefronts per
// workgroup: the workgroup dispatcher's rate, and how much of a real launch it can hide behind.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/dispatch_rate.hip -o /tmp/dispatch_rate && /tmp/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WAVES, int LDS>
__global__ __launch_bounds__(64 * WAVES, 8 / WAVES > 0 ? 8 / WAVES : 1) void k_empty(uint32_t *out, int sleep_units)
{
    __shared__ uint32_t lds[LDS / 4];
    // 64 vector registers in the kernel descriptor, like k_mb
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    for (int i = 0; i < sleep_units; ++i) asm volatile("s_sleep 127");       // 127 * 64 cycles each
    if (out != nullptr && threadIdx.x == 1000) { lds[threadIdx.x] = 1; out[blockIdx.x] = lds[0]; }
}

template <int WAVES, int LDS>
static void run(const char *name, int mbs, int sleep_units)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid((unsigned)(mbs / WAVES)), block(64 * WAVES);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_empty<WAVES, LDS>), grid, block, 0, 0, (uint32_t *)nullptr, sleep_units);
    hipDeviceSynchronize();
    const int reps = 50;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_empty<WAVES, LDS>), grid, block, 0, 0, (uint32_t *)nullptr, sleep_units);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1000.0 / reps;
    printf("%-44s %6d workgroups x %d waves, idle %6d cycles: %8.2f us per launch = %6.2f ns per workgroup\n", name, mbs / WAVES, WAVES,
           sleep_units * 127 * 64, us, us * 1000.0 / (mbs / WAVES));
}

int main()
{
    const int mbs = 86400;
    for (int sl : {0, 1, 3}) {
        run<1, 4288>("1 wave / workgroup, 4288 B LDS", mbs, sl);
        run<1, 16>("1 wave / workgroup, 16 B LDS", mbs, sl);
        run<2, 8576>("2 waves / workgroup, 8576 B LDS", mbs, sl);
        run<4, 17152>("4 waves / workgroup, 17152 B LDS", mbs, sl);
    }
    run<1, 4288>("1 wave / workgroup, 10x the grid", mbs * 10, 0);
    return 0;
}
