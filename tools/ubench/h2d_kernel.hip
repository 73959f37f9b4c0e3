// h2d_kernel.hip - what does ONE upload of a GOP's frames cost from page-locked host memory, by the copy engine and by a kernel?
// (tools/ubench, not part of the library.)  The port path's blocking contract (m2v_push_frames returns when its frames have been read)
// pays the transfer's fixed cost once per call; this probe separates the fixed cost from the bytes:
//   for each size: K back-to-back { start transfer; wait for it } by (a) hipMemcpyAsync + hipStreamSynchronize, (b) a copy kernel that reads
//   the host memory through its device pointer (W workgroups of 256 lanes, 16 bytes per lane per turn) + hipStreamSynchronize.
// build: hipcc --offload-arch=gfx950 -O2 -o h2d_kernel h2d_kernel.hip        run: ./h2d_kernel [MiB ...]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy(u32x4 *__restrict__ dst, const u32x4 *__restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    std::vector<size_t> sizes;
    for (int i = 1; i < argc; ++i) sizes.push_back((size_t)atoi(argv[i]) << 20);
    if (sizes.empty()) sizes = {1u << 20, 8u << 20, (size_t)57 << 20, (size_t)114 << 20};
    const size_t maxb = 128u << 20;
    unsigned char *h = nullptr, *d = nullptr;
    CHK(hipHostMalloc((void **)&h, maxb, hipHostMallocDefault));
    CHK(hipMalloc((void **)&d, maxb));
    for (size_t i = 0; i < maxb; i += 4096) h[i] = (unsigned char)(i >> 12);
    void *hd = nullptr;
    CHK(hipHostGetDevicePointer(&hd, h, 0));
    hipStream_t s;
    CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int K = 10;
    for (size_t bytes : sizes) {
        if (bytes > maxb) continue;
        // (a) copy engine
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            for (int k = 0; k < K; ++k) { CHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CHK(hipStreamSynchronize(s)); }
            const double dt = (now() - t0) / K;
            if (rep) printf("%4zu MiB  hipMemcpyAsync + sync      %8.1f us  %6.2f GB/s\n", bytes >> 20, dt * 1e6, bytes / dt * 1e-9);
        }
        // (b) kernel, several grid sizes
        for (int wgs : {16, 32, 64, 128, 256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                for (int k = 0; k < K; ++k) {
                    hipLaunchKernelGGL(k_copy<4>, dim3(wgs), dim3(256), 0, s, (u32x4 *)d, (const u32x4 *)hd, bytes / 16);
                    CHK(hipStreamSynchronize(s));
                }
                const double dt = (now() - t0) / K;
                if (rep) printf("%4zu MiB  kernel %3d x 256, 4 x 16 B   %8.1f us  %6.2f GB/s\n", bytes >> 20, wgs, dt * 1e6, bytes / dt * 1e-9);
            }
        }
        // (c) the same transfer cut in two halves on two streams (two copy engines, if the runtime deals them so)
        hipStream_t s2;
        CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; ++rep) {
            const double t0 = now();
            for (int k = 0; k < K; ++k) {
                const size_t half = (bytes / 2) & ~(size_t)255;
                CHK(hipMemcpyAsync(d, h, half, hipMemcpyHostToDevice, s));
                CHK(hipMemcpyAsync(d + half, h + half, bytes - half, hipMemcpyHostToDevice, s2));
                CHK(hipStreamSynchronize(s)); CHK(hipStreamSynchronize(s2));
            }
            const double dt = (now() - t0) / K;
            if (rep) printf("%4zu MiB  two halves on two streams  %8.1f us  %6.2f GB/s\n", bytes >> 20, dt * 1e6, bytes / dt * 1e-9);
        }
        // (d) the same upload with a 3.4 MB read-back (one chunk's stream bytes) issued beside it on another stream: does the copy engine
        //     serve both directions at once?  (e) ... with an event recorded behind the upload and the host waiting for the STREAM
        {
            unsigned char *h2 = nullptr, *d2 = nullptr;
            const size_t back = (size_t)3400 << 10;
            CHK(hipHostMalloc((void **)&h2, back, hipHostMallocDefault));
            CHK(hipMalloc((void **)&d2, back));
            hipEvent_t ev;
            CHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            hipEvent_t ev_nf;
            CHK(hipEventCreateWithFlags(&ev_nf, hipEventDisableTiming | hipEventDisableSystemFence));
            for (int mode = 0; mode < 6; ++mode)
                for (int rep = 0; rep < 2; ++rep) {
                    const double t0 = now();
                    double up = 0;
                    for (int k = 0; k < K; ++k) {
                        const double a = now();
                        if (mode == 1) CHK(hipMemcpyAsync(h2, d2, back, hipMemcpyDeviceToHost, s2));
                        CHK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
                        if (mode == 2) CHK(hipMemcpyAsync(h2, d2, back, hipMemcpyDeviceToHost, s2));
                        if (mode == 3) { CHK(hipEventRecord(ev, s)); CHK(hipStreamWaitEvent(s2, ev, 0)); }
                        if (mode == 4) { CHK(hipEventRecord(ev_nf, s)); CHK(hipStreamWaitEvent(s2, ev_nf, 0)); }
                        if (mode == 5) { CHK(hipEventRecord(ev_nf, s)); CHK(hipStreamWaitEvent(s2, ev_nf, 0)); CHK(hipEventSynchronize(ev_nf)); }
                        CHK(hipStreamSynchronize(s));
                        up += now() - a;
                        CHK(hipStreamSynchronize(s2));
                    }
                    const double dt = (now() - t0) / K;
                    static const char *const names[6] = {"upload alone", "3.4 MB read-back issued first", "3.4 MB read-back issued behind", "event behind the upload, stream waited for",
                                                                "... an event without system fence", "... the same, the EVENT waited for"};
                    if (rep) printf("%4zu MiB  %-44s upload %8.1f us  (both %8.1f us)\n", bytes >> 20, names[mode], up / K * 1e6, dt * 1e6);
                }
            CHK(hipEventDestroy(ev)); CHK(hipEventDestroy(ev_nf));
            CHK(hipFree(d2)); CHK(hipHostFree(h2));
        }
        CHK(hipStreamDestroy(s2));
        // correctness of the kernel copy (sampled)
        CHK(hipMemset(d, 0, bytes));
        hipLaunchKernelGGL(k_copy<4>, dim3(64), dim3(256), 0, s, (u32x4 *)d, (const u32x4 *)hd, bytes / 16);
        CHK(hipStreamSynchronize(s));
        std::vector<unsigned char> back(bytes);
        CHK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
        printf("          kernel copy %s\n", memcmp(back.data(), h, bytes) == 0 ? "identical" : "DIFFERS");
    }
    return 0;
}
