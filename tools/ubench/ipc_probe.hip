// ipc_probe.hip - does the peer transport's mechanism work between two PROCESSES on this box?  (tools/ubench, not part of the library)
//   process A allocates fine-grained device memory (landing buffer + flag), exports a hipIpcMemHandle, and runs a kernel that polls the
//   flag (bounded), then checks the payload;
//   process B opens the handle and runs a kernel whose blocks store their payload write-through (system-scope relaxed atomic stores =
//   global_store ... sc0 sc1), drain, and add 1 to the flag with a system-scope atomic.
// Both kernels are in flight at the same time on the same GPU (or on two GPUs: ipc_probe <devA> <devB>).
// build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", (int)getpid(), #x, hipGetErrorString(e_)); _exit(3); } } while (0)

constexpr int kBlocks = 256, kWords = 64;        // 256 blocks x 256 B
typedef __attribute__((address_space(1))) unsigned int *gu32;

__global__ void producer(unsigned int *payload, unsigned int *flag, unsigned int salt)
{
    const int b = blockIdx.x, t = threadIdx.x;
    __hip_atomic_store((gu32)(payload + b * kWords + t), salt + (unsigned)(b * kWords + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_fetch_add((gu32)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void consumer(const unsigned int *payload, unsigned int *flag, unsigned int need, unsigned int salt, unsigned int *result, long long budget)
{
    const int t = threadIdx.x;
    const long long t0 = wall_clock64();
    unsigned int seen = 0;
    for (;;) {
        seen = __hip_atomic_load((gu32)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (seen >= need || wall_clock64() - t0 > budget) break;
        __builtin_amdgcn_s_sleep(32);
    }
    unsigned int bad = 0;
    if (seen >= need)
        for (int i = t; i < kBlocks * kWords; i += blockDim.x)
            bad += __hip_atomic_load((gu32)(payload + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != salt + (unsigned)i;
    atomicAdd(&result[1], bad);
    if (t == 0) { result[0] = seen; result[2] = (unsigned int)((wall_clock64() - t0) / 100); }      // 100 MHz -> us
}

int main(int argc, char **argv)
{
    const int devA = argc > 1 ? atoi(argv[1]) : 0, devB = argc > 2 ? atoi(argv[2]) : devA;
    const int fine = argc > 3 ? atoi(argv[3]) : 1;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    const pid_t pid = fork();               // before anything touches HIP
    if (pid == 0) {
        // ---- process A: owner + consumer ----
        CHK(hipSetDevice(devA));
        unsigned int *mem = nullptr, *result = nullptr;
        const size_t bytes = (size_t)(kBlocks * kWords + 64) * 4;
        hipError_t ae = fine ? hipExtMallocWithFlags((void **)&mem, bytes, hipDeviceMallocFinegrained) : hipMalloc((void **)&mem, bytes);
        printf("A: %s -> %s\n", fine ? "hipExtMallocWithFlags(finegrained)" : "hipMalloc", hipGetErrorString(ae));
        if (ae != hipSuccess) _exit(3);
        CHK(hipMemset(mem, 0, bytes));
        CHK(hipMalloc((void **)&result, 16));
        CHK(hipMemset(result, 0, 16));
        CHK(hipDeviceSynchronize());
        hipIpcMemHandle_t h;
        hipError_t ie = hipIpcGetMemHandle(&h, mem);
        printf("A: hipIpcGetMemHandle -> %s\n", hipGetErrorString(ie));
        fflush(stdout);
        if (ie != hipSuccess) _exit(3);
        for (int round = 0; round < 3; ++round) {
            const unsigned salt = 1000u * (round + 1);
            hipLaunchKernelGGL(consumer, dim3(1), dim3(256), 0, 0, mem, mem + kBlocks * kWords, (unsigned)kBlocks * (round + 1), salt, result, 200000000ll);   // 2 s
            CHK(hipGetLastError());
            if (round == 0) { if (write(c2p[1], &h, sizeof h) != (ssize_t)sizeof h) _exit(4); }
            else { char c = 'g'; if (write(c2p[1], &c, 1) != 1) _exit(4); }
            CHK(hipDeviceSynchronize());
            unsigned int r[4];
            CHK(hipMemcpy(r, result, 16, hipMemcpyDeviceToHost));
            CHK(hipMemset(result, 0, 16));
            printf("A: round %d: flag seen %u of %u, wrong payload words %u, waited %u us -> %s\n", round, r[0], kBlocks * (round + 1), r[1], r[2],
                   r[0] >= (unsigned)kBlocks * (round + 1) && r[1] == 0 ? "OK" : "FAILED");
            fflush(stdout);
            char c;
            if (read(p2c[0], &c, 1) != 1) _exit(4);          // B has finished its round
        }
        _exit(0);
    }
    // ---- process B: producer ----
    hipIpcMemHandle_t h;
    if (read(c2p[0], &h, sizeof h) != (ssize_t)sizeof h) { fprintf(stderr, "B: no handle\n"); int st; waitpid(pid, &st, 0); return 3; }
    CHK(hipSetDevice(devB));
    unsigned int *mem = nullptr;
    hipError_t oe = hipIpcOpenMemHandle((void **)&mem, h, hipIpcMemLazyEnablePeerAccess);
    printf("B: hipIpcOpenMemHandle -> %s\n", hipGetErrorString(oe));
    fflush(stdout);
    if (oe != hipSuccess) { kill(pid, SIGKILL); return 3; }
    for (int round = 0; round < 3; ++round) {
        if (round) { char c; if (read(c2p[0], &c, 1) != 1) return 4; }
        usleep(20000);                          // A's consumer is polling by now
        hipLaunchKernelGGL(producer, dim3(kBlocks), dim3(kWords), 0, 0, mem, mem + kBlocks * kWords, 1000u * (round + 1));
        CHK(hipGetLastError());
        CHK(hipDeviceSynchronize());
        char c = 'd';
        if (write(p2c[1], &c, 1) != 1) return 4;
    }
    int st = 0;
    waitpid(pid, &st, 0);
    CHK(hipIpcCloseMemHandle(mem));
    printf("B: done, A exited with %d\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return WIFEXITED(st) ? WEXITSTATUS(st) : 5;
}
