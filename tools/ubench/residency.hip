// residency.hip - how many workgroups of k_mb's shape (one wavefront, 64 VGPRs, 4 288 B LDS) are RESIDENT AT ONCE on gfx950.
// Every workgroup notes the constant-rate clock (s_memrealtime, 100 MHz) when it starts and idles ~40 us: the spread of the start
// times is how long the dispatcher takes to bring the whole grid in; the (xcd, cu) census comes from HW_ID.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/residency.hip -o /tmp/residency && /tmp/residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

template <int LDS>
__global__ __launch_bounds__(64, 8) void k_count(uint32_t *counter, uint32_t *seen, uint32_t *hw)
{
    __shared__ uint32_t lds[LDS / 4];
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    unsigned long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < 12; ++i) asm volatile("s_sleep 127");       // ~ 97 500 cycles
    if (threadIdx.x == 0) {
        seen[blockIdx.x] = (uint32_t)t0;
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        hw[blockIdx.x] = id;
        if (seen[blockIdx.x] == 0xFFFFFFFFu) lds[0] = 1;
    }
}

template <int LDS>
static void run(unsigned grid)
{
    uint32_t *d_counter, *d_seen, *d_hw;
    hipMalloc(&d_counter, 4); hipMalloc(&d_seen, grid * 4); hipMalloc(&d_hw, grid * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d_counter, 0, 4);
        hipLaunchKernelGGL((k_count<LDS>), dim3(grid), dim3(64), 0, 0, d_counter, d_seen, d_hw);
        hipDeviceSynchronize();
    }
    std::vector<uint32_t> seen(grid), hw(grid);
    hipMemcpy(seen.data(), d_seen, grid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hw.data(), d_hw, grid * 4, hipMemcpyDeviceToHost);
    std::sort(seen.begin(), seen.end());
    // waves per (xcc, se, cu): HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9 layout)
    std::vector<int> per_cu(1 << 12, 0);
    for (unsigned i = 0; i < grid; ++i) per_cu[((hw[i] >> 8) & 0xFF) | ((i & 7u) << 8)]++;
    int used = 0, mx = 0, mn = 1 << 30;
    for (int c : per_cu) if (c) { ++used; mx = std::max(mx, c); mn = std::min(mn, c); }
    printf("LDS %5d B  grid %6u: start times after the first (us)  median %.2f  90 %% %.2f  99 %% %.2f  last %.2f   | distinct (xcd, cu) %d, workgroups per CU %d .. %d\n", LDS, grid,
           (seen[grid / 2] - seen[0]) * 0.01, (seen[grid * 9 / 10] - seen[0]) * 0.01, (seen[grid * 99 / 100] - seen[0]) * 0.01, (seen[grid - 1] - seen[0]) * 0.01, used, mn, mx);
    hipFree(d_counter); hipFree(d_seen); hipFree(d_hw);
}

int main()
{
    for (unsigned g : {4096u, 6144u, 8192u, 10240u}) run<4288>(g);
    for (unsigned g : {8192u}) { run<16>(g); run<5056>(g); run<2048>(g); }
    return 0;
}
