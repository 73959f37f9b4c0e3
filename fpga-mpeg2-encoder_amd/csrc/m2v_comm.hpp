// m2v_comm.hpp — the exchange step of strip mode (BASELINE config c5) behind one small interface.
//
// A strip needs, per GOP step, the neighbours' +-2*VECTOR_LEVEL luma / +-VECTOR_LEVEL chroma rows of the reconstruction it
// just wrote (window geometry RTL:1446-1448), once per sequence everybody's per-frame strip sizes, and the strips themselves on
// the rank that owns the output.  xGMI is point-to-point and so is this traffic: send / recv pairs with the two neighbours
// and 7 sends into the output rank - nothing is reduced.  Two implementations:
//
//   RcclComm   one process per GPU; ncclSend / ncclRecv / ncclAllGather of librccl, which is dlopen()ed on first use (the
//              encoder library itself has no link-time dependency on RCCL; a torch process has librccl mapped already and
//              gets that copy).  The communicator is created from a ncclUniqueId that the caller distributes
//              (m2v_comm_unique_id on rank 0, any broadcast, m2v_comm_init_rccl on every rank).
//   LocalComm  `world` handles of ONE process - one host thread each, on one GPU or several - exchange through mailboxes:
//              the sender posts a device pointer and an event, the receiver makes its stream wait for the event and copies
//              device to device.  This is what runs the N-rank code path on a 1-GPU box (tests, tools/step_timeline.py).
//
// All calls enqueue on the stream they are given and return; nothing here synchronises a stream except LocalComm's
// size exchange (which has to hand bytes from one thread to another through the host).
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the symbols are resolved with dlsym

#include <dlfcn.h>

#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/m2v_mi355x.h"

namespace m2v { struct PeerState; }

struct m2v_comm {
    int world = 1;
    virtual ~m2v_comm() {}
    // the peer transport's state when this communicator offers it (PeerComm), else null
    virtual m2v::PeerState *peer() { return nullptr; }
    // halo rows of one GOP step: nbytes to / from the rank above (rank - 1) and the rank below (rank + 1); a null pair = no neighbour
    virtual void halo(int rank, const void *send_up, void *recv_up, const void *send_down, void *recv_down, size_t nbytes, hipStream_t s) = 0;
    // every rank's `count` values into all[world][count] on every rank (device memory on both sides)
    virtual void allgather_u64(int rank, const unsigned long long *d_src, unsigned long long *d_all, size_t count, hipStream_t s) = 0;
    // rank r != dst sends sizes[r] bytes of d_strip to dst, which receives them in bufs[r]
    virtual void gather(int rank, int dst, const void *d_strip, const size_t *sizes, void *const *bufs, hipStream_t s) = 0;
    // self-test: nbytes from d_send to d_recv through the transport's own send / recv pair addressed to this very rank
    virtual void loopback(int rank, const void *d_send, void *d_recv, size_t nbytes, hipStream_t s) = 0;
    // a rank has failed: ranks blocked in (or arriving at) an exchange give up with an error instead of waiting for it (in-process
    // communicator; RCCL has no cheap equivalent - a failed rank of a multi-process job takes the job down, bench.py's launcher does that)
    virtual void abort() {}
    // true when every call above only ENQUEUES on the stream it is given, so that a whole strip sequence can be recorded into a
    // hipGraph (m2v_strip_encode) - false for the in-process communicator, whose calls block on the other threads
    virtual bool capturable() const { return false; }
    // ... and whether m2v_strip_encode does so without being asked (option strip_graph = -1, the default): the single-GPU timing
    // communicators yes; a real multi-rank RCCL communicator only when the caller opts in (strip_graph = 1)
    virtual bool graph_by_default() const { return false; }
    virtual const char *kind() const = 0;
};

namespace m2v {

struct CommError : std::runtime_error { using std::runtime_error::runtime_error; };

#define M2V_COMM_HIP(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) throw CommError(std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// librccl through dlopen
// ---------------------------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    std::string err;

    static RcclApi &get()
    {
        static RcclApi api;
        static std::once_flag once;
        std::call_once(once, [] { api.load(); });
        return api;
    }
    bool ok() const { return h != nullptr && err.empty(); }

private:
    void load()
    {
        // a process that imported torch has its librccl mapped under this soname already; otherwise the ROCm copy
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) { err = std::string("librccl not found: ") + dlerror(); return; }
#define M2V_RCCL_SYM(field, sym)                                         \
        field = (decltype(field))dlsym(h, #sym);                         \
        if (!field) { err = "librccl lacks " #sym; return; }
        M2V_RCCL_SYM(GetUniqueId, ncclGetUniqueId)
        M2V_RCCL_SYM(CommInitRank, ncclCommInitRank)
        M2V_RCCL_SYM(CommDestroy, ncclCommDestroy)
        M2V_RCCL_SYM(GetErrorString, ncclGetErrorString)
        M2V_RCCL_SYM(GroupStart, ncclGroupStart)
        M2V_RCCL_SYM(GroupEnd, ncclGroupEnd)
        M2V_RCCL_SYM(Send, ncclSend)
        M2V_RCCL_SYM(Recv, ncclRecv)
        M2V_RCCL_SYM(AllGather, ncclAllGather)
#undef M2V_RCCL_SYM
    }
};

struct RcclComm final : m2v_comm {
    RcclApi &api;
    ncclComm_t comm = nullptr;
    int rank;
    RcclComm(const ncclUniqueId &id, int rank_, int world_) : api(RcclApi::get()), rank(rank_)
    {
        world = world_;
        if (!api.ok()) throw CommError(api.err);
        chk(api.CommInitRank(&comm, world_, id, rank_), "ncclCommInitRank");
    }
    ~RcclComm() override { if (comm) (void)api.CommDestroy(comm); }
    const char *kind() const override { return "rccl"; }
    bool capturable() const override { return true; }
    bool graph_by_default() const override { return world == 1; }
    void chk(ncclResult_t r, const char *what) const
    {
        if (r != ncclSuccess) throw CommError(std::string(what) + ": " + api.GetErrorString(r));
    }
    // ncclGroupStart; ncclSend / ncclRecv x 2; ncclGroupEnd - one fused point-to-point step per GOP step
    void halo(int r, const void *send_up, void *recv_up, const void *send_down, void *recv_down, size_t n, hipStream_t s) override
    {
        if (!n) return;
        chk(api.GroupStart(), "ncclGroupStart");
        if (r > 0 && send_up && recv_up) {
            chk(api.Send(send_up, n, ncclUint8, r - 1, comm, s), "ncclSend(up)");
            chk(api.Recv(recv_up, n, ncclUint8, r - 1, comm, s), "ncclRecv(up)");
        }
        if (r < world - 1 && send_down && recv_down) {
            chk(api.Send(send_down, n, ncclUint8, r + 1, comm, s), "ncclSend(down)");
            chk(api.Recv(recv_down, n, ncclUint8, r + 1, comm, s), "ncclRecv(down)");
        }
        chk(api.GroupEnd(), "ncclGroupEnd");
    }
    void allgather_u64(int, const unsigned long long *d_src, unsigned long long *d_all, size_t count, hipStream_t s) override
    {
        chk(api.AllGather(d_src, d_all, count, ncclUint64, comm, s), "ncclAllGather");
    }
    void loopback(int r, const void *d_send, void *d_recv, size_t n, hipStream_t s) override
    {
        chk(api.GroupStart(), "ncclGroupStart");
        chk(api.Send(d_send, n, ncclUint8, r, comm, s), "ncclSend(self)");
        chk(api.Recv(d_recv, n, ncclUint8, r, comm, s), "ncclRecv(self)");
        chk(api.GroupEnd(), "ncclGroupEnd");
    }
    void gather(int r, int dst, const void *d_strip, const size_t *sizes, void *const *bufs, hipStream_t s) override
    {
        chk(api.GroupStart(), "ncclGroupStart");
        if (r != dst) {
            if (sizes[r]) chk(api.Send(d_strip, sizes[r], ncclUint8, dst, comm, s), "ncclSend(strip)");
        } else {
            for (int k = 0; k < world; ++k)
                if (k != dst && sizes[k]) chk(api.Recv(bufs[k], sizes[k], ncclUint8, k, comm, s), "ncclRecv(strip)");
        }
        chk(api.GroupEnd(), "ncclGroupEnd");
    }
};

// ---------------------------------------------------------------------------------------------
// ranks = threads of this process
// ---------------------------------------------------------------------------------------------
struct LocalComm final : m2v_comm {
    static constexpr int kMax = 16;
    struct Slot {
        const void *ptr = nullptr;
        size_t n = 0;
        int dev = -1;                   // the sender's device
        // `ready` is recorded by the sender on its stream and `consumed` by the receiver on its own: with the two handles on
        // different GPUs an event must belong to the device of the stream that records it, so each side creates its own
        hipEvent_t ready = nullptr, consumed = nullptr;
        bool full = false, taken = false;
    };
    bool peer_on[kMax][kMax] = {};      // (under mu) hipDeviceEnablePeerAccess done for [reader][owner]
    std::mutex mu;
    std::condition_variable cv;
    Slot halo_slot[kMax][2];            // [sender][0 = to the rank above, 1 = to the rank below]
    Slot strip_slot[kMax];              // [sender]: its strip for the output rank
    std::vector<unsigned long long> stage[kMax];
    int bar_count = 0;
    unsigned long long bar_gen = 0;
    bool aborted = false;               // (under mu)

    void abort() override
    {
        std::unique_lock<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
    void check(const std::unique_lock<std::mutex> &) const
    {
        if (aborted) throw CommError("local exchange: another rank of this communicator has failed");
    }

    explicit LocalComm(int w) { world = w; }
    ~LocalComm() override
    {
        auto drop = [](Slot &sl) { if (sl.ready) (void)hipEventDestroy(sl.ready); if (sl.consumed) (void)hipEventDestroy(sl.consumed); };
        for (auto &pair : halo_slot) for (auto &sl : pair) drop(sl);
        for (auto &sl : strip_slot) drop(sl);
    }
    const char *kind() const override { return "local"; }

    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long gen = bar_gen;
        check(lk);
        if (++bar_count == world) { bar_count = 0; ++bar_gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return bar_gen != gen || aborted; });
        check(lk);
    }
    // sender: the bytes at ptr are final once everything enqueued on `s` so far has run
    void post(Slot &sl, const void *ptr, size_t n, hipStream_t s)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return (!sl.full && !sl.taken) || aborted; });
        check(lk);
        if (!sl.ready) M2V_COMM_HIP(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
        M2V_COMM_HIP(hipGetDevice(&sl.dev));
        M2V_COMM_HIP(hipEventRecord(sl.ready, s));
        sl.ptr = ptr; sl.n = n; sl.full = true;
        cv.notify_all();
    }
    // receiver: device-to-device copy on its own stream, behind the sender's event
    void take(Slot &sl, void *dst, size_t n, hipStream_t s)
    {
        const void *src;
        int mine = -1, theirs = -1;
        M2V_COMM_HIP(hipGetDevice(&mine));
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return sl.full || aborted; });
            check(lk);
            if (sl.n != n) throw CommError("local exchange: the two sides disagree about the size");
            src = sl.ptr;
            theirs = sl.dev;
            if (!sl.consumed) M2V_COMM_HIP(hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));    // on the receiver's device
            // two GPUs: direct access over xGMI where the devices offer it (the copy below works without, through a staging hop)
            if (theirs != mine && mine >= 0 && theirs >= 0 && mine < kMax && theirs < kMax && !peer_on[mine][theirs]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, mine, theirs) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(theirs, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) throw CommError(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
                }
                (void)hipGetLastError();
                peer_on[mine][theirs] = true;
            }
        }
        M2V_COMM_HIP(hipStreamWaitEvent(s, sl.ready, 0));
        if (n) {
            if (theirs == mine) M2V_COMM_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
            else M2V_COMM_HIP(hipMemcpyPeerAsync(dst, mine, src, theirs, n, s));
        }
        M2V_COMM_HIP(hipEventRecord(sl.consumed, s));
        std::unique_lock<std::mutex> lk(mu);
        sl.full = false; sl.taken = true;
        cv.notify_all();
    }
    // sender again: its buffer may be rewritten by work enqueued on `s` after this
    void release(Slot &sl, hipStream_t s)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return sl.taken || aborted; });
        check(lk);
        M2V_COMM_HIP(hipStreamWaitEvent(s, sl.consumed, 0));
        sl.taken = false;
        cv.notify_all();
    }

    void halo(int r, const void *send_up, void *recv_up, const void *send_down, void *recv_down, size_t n, hipStream_t s) override
    {
        if (!n) return;
        const bool up = r > 0 && send_up && recv_up, down = r < world - 1 && send_down && recv_down;
        // all posts first, then the receives (which only wait for the neighbours' posts), then the releases: no cycle
        if (up) post(halo_slot[r][0], send_up, n, s);
        if (down) post(halo_slot[r][1], send_down, n, s);
        if (up) take(halo_slot[r - 1][1], recv_up, n, s);            // the bottom rows of the rank above
        if (down) take(halo_slot[r + 1][0], recv_down, n, s);        // the top rows of the rank below
        if (up) release(halo_slot[r][0], s);
        if (down) release(halo_slot[r][1], s);
    }
    void allgather_u64(int r, const unsigned long long *d_src, unsigned long long *d_all, size_t count, hipStream_t s) override
    {
        barrier();                                  // the previous round's readers are done with the staging
        stage[r].resize(count);
        M2V_COMM_HIP(hipMemcpyAsync(stage[r].data(), d_src, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        M2V_COMM_HIP(hipStreamSynchronize(s));
        barrier();
        for (int k = 0; k < world; ++k)
            M2V_COMM_HIP(hipMemcpyAsync(d_all + (size_t)k * count, stage[k].data(), count * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        M2V_COMM_HIP(hipStreamSynchronize(s));      // pageable staging: read before anyone resizes it
    }
    void loopback(int r, const void *d_send, void *d_recv, size_t n, hipStream_t s) override
    {
        post(strip_slot[r], d_send, n, s);
        take(strip_slot[r], d_recv, n, s);
        release(strip_slot[r], s);
    }
    void gather(int r, int dst, const void *d_strip, const size_t *sizes, void *const *bufs, hipStream_t s) override
    {
        if (r != dst) {
            post(strip_slot[r], d_strip, sizes[r], s);
            release(strip_slot[r], s);
        } else {
            for (int k = 0; k < world; ++k)
                if (k != dst) take(strip_slot[k], bufs[k], sizes[k], s);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// ONE rank of `world`, alone on its GPU, with nobody to talk to: the halo it "receives" is its own (a device copy of the same
// size), everybody's sizes are its own, and the other ranks' strips are never sent.  The stream that comes out is NOT a valid
// encoding (the neighbour rows are wrong) - this exists to TIME what one rank of an N-GPU job does per GOP step on a real
// GPU (kernels of a 1/N strip, launch gaps, host time) when only one GPU is at hand: tools/strip_solo.py.
// ---------------------------------------------------------------------------------------------
struct SoloComm final : m2v_comm {
    // with_rccl: the rows travel through a 1-rank RCCL communicator as ncclSend / ncclRecv pairs addressed to this very rank inside
    // ncclGroupStart / ncclGroupEnd - the call pattern of RcclComm::halo, RCCL's own kernels on the stream - so that one GPU can
    // show what RCCL costs per GOP step and that it can be recorded into a hipGraph.  Otherwise plain device copies.
    RcclComm *self = nullptr;
    SoloComm(int w, bool with_rccl)
    {
        world = w;
        if (with_rccl) {
            RcclApi &api = RcclApi::get();
            if (!api.ok()) throw CommError(api.err);
            ncclUniqueId id;
            const ncclResult_t r = api.GetUniqueId(&id);
            if (r != ncclSuccess) throw CommError(std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
            self = new RcclComm(id, 0, 1);
        }
    }
    ~SoloComm() override { delete self; }
    const char *kind() const override { return self ? "solo-rccl" : "solo"; }
    bool capturable() const override { return true; }
    bool graph_by_default() const override { return true; }
    void halo(int r, const void *send_up, void *recv_up, const void *send_down, void *recv_down, size_t n, hipStream_t s) override
    {
        if (!n) return;
        const bool up = r > 0 && send_up && recv_up, down = r < world - 1 && send_down && recv_down;
        if (self) {
            // sends and receives to oneself pair up in the order they were issued
            RcclApi &api = self->api;
            self->chk(api.GroupStart(), "ncclGroupStart");
            if (up) {
                self->chk(api.Send(send_up, n, ncclUint8, 0, self->comm, s), "ncclSend(self, up)");
                self->chk(api.Recv(recv_up, n, ncclUint8, 0, self->comm, s), "ncclRecv(self, up)");
            }
            if (down) {
                self->chk(api.Send(send_down, n, ncclUint8, 0, self->comm, s), "ncclSend(self, down)");
                self->chk(api.Recv(recv_down, n, ncclUint8, 0, self->comm, s), "ncclRecv(self, down)");
            }
            self->chk(api.GroupEnd(), "ncclGroupEnd");
            return;
        }
        if (up) M2V_COMM_HIP(hipMemcpyAsync(recv_up, send_up, n, hipMemcpyDeviceToDevice, s));
        if (down) M2V_COMM_HIP(hipMemcpyAsync(recv_down, send_down, n, hipMemcpyDeviceToDevice, s));
    }
    void allgather_u64(int, const unsigned long long *d_src, unsigned long long *d_all, size_t count, hipStream_t s) override
    {
        if (self) {                             // the collective itself on the 1-rank communicator (into row 0), the other rows copied
            self->allgather_u64(0, d_src, d_all, count, s);
            for (int k = 1; k < world; ++k)
                M2V_COMM_HIP(hipMemcpyAsync(d_all + (size_t)k * count, d_all, count * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
            return;
        }
        for (int k = 0; k < world; ++k)
            M2V_COMM_HIP(hipMemcpyAsync(d_all + (size_t)k * count, d_src, count * sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    }
    void gather(int r, int dst, const void *d_strip, const size_t *sizes, void *const *bufs, hipStream_t s) override
    {
        if (r != dst) return;
        for (int k = 0; k < world; ++k)       // the output rank "receives" copies of its own strip: the same bytes moved as in a real gather
            if (k != dst && sizes[k]) M2V_COMM_HIP(hipMemcpyAsync(bufs[k], d_strip, sizes[k], hipMemcpyDeviceToDevice, s));
    }
    void loopback(int, const void *d_send, void *d_recv, size_t n, hipStream_t s) override
    {
        if (self) { self->loopback(0, d_send, d_recv, n, s); return; }
        M2V_COMM_HIP(hipMemcpyAsync(d_recv, d_send, n, hipMemcpyDeviceToDevice, s));
    }
};

// ---------------------------------------------------------------------------------------------
// The exchange supplied by the CALLER (m2v_comm_init_callbacks): three functions of the host program - its MPI, its torch.distributed
// process group, a test's pipes - behind the same interface.  Each gets device pointers and the stream the data is ordered on and
// must leave its effect ordered on that stream (enqueue there, or synchronise it, move the bytes, and return).  0 = success.
// ---------------------------------------------------------------------------------------------
struct CallbackComm final : m2v_comm {
    m2v_comm_callbacks cb;
    explicit CallbackComm(int w, const m2v_comm_callbacks &c) : cb(c) { world = w; }
    const char *kind() const override { return "callbacks"; }
    static void chk(int r, const char *what) { if (r != 0) throw CommError(std::string(what) + ": the caller's function returned " + std::to_string(r)); }
    void halo(int r, const void *send_up, void *recv_up, const void *send_down, void *recv_down, size_t n, hipStream_t s) override
    {
        if (!n) return;
        const bool up = r > 0 && send_up && recv_up, down = r < world - 1 && send_down && recv_down;
        chk(cb.halo(cb.user, r, up ? send_up : nullptr, up ? recv_up : nullptr, down ? send_down : nullptr, down ? recv_down : nullptr, n, (void *)s), "halo");
    }
    void allgather_u64(int r, const unsigned long long *d_src, unsigned long long *d_all, size_t count, hipStream_t s) override
    {
        chk(cb.allgather_u64(cb.user, r, d_src, d_all, count, (void *)s), "allgather_u64");
    }
    void gather(int r, int dst, const void *d_strip, const size_t *sizes, void *const *bufs, hipStream_t s) override
    {
        chk(cb.gather(cb.user, r, dst, d_strip, sizes, bufs, (void *)s), "gather");
    }
    void loopback(int, const void *d_send, void *d_recv, size_t n, hipStream_t s) override
    {
        M2V_COMM_HIP(hipMemcpyAsync(d_recv, d_send, n, hipMemcpyDeviceToDevice, s));
    }
};

// ---------------------------------------------------------------------------------------------
// The PEER transport: no exchange step at all.  Every rank owns a LANDING BLOCK in fine-grained device memory - four buffers (rows
// from the rank above / below, two GOP-step parities) and a few counters -, its two neighbours map it (the same process: the pointer,
// plus hipDeviceEnablePeerAccess across GPUs; another process: hipIpcOpenMemHandle), and the macroblock kernel of the strip's edge
// rows stores their outer rows straight into the neighbour's block and counts its arrival there (k_mb<.., EDGE, PEER>, PeerStep).
// "or peer-to-peer stores over xGMI".  Sizes and strips still travel through the BASE communicator this one wraps
// (RCCL, local, callbacks), which is also what the halo falls back to - for good - when a wait runs out of budget.
//
//   landing block:  [control][buf(parity 0, from above)][buf(0, from below)][buf(1, from above)][buf(1, from below)]   cap bytes each
//   control:        arrival counters cnt[set][GOP of the sequence][side] on 128-byte lines (set = sequence parity: a sequence clears
//                   the set of the NEXT one, which nobody touches meanwhile - every rank is past the previous sequence's all-gather -,
//                   so no rank ever clears a counter a neighbour may be adding to), then the give-up word
// ---------------------------------------------------------------------------------------------
struct PeerDesc {                    // what a rank tells its neighbours (m2v_comm_peer_export): plain bytes, M2V_PEER_DESC_BYTES
    uint32_t magic, version;
    int32_t pid, device;
    unsigned long long ptr, bytes, cap;
    unsigned long long nonce;        // drawn once per process: "the same process" is decided by this, not by the pid (ranks in a container each may all be pid 1)
    hipIpcMemHandle_t ipc;
    uint8_t pad[M2V_PEER_DESC_BYTES - 48 - sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(PeerDesc) == M2V_PEER_DESC_BYTES, "the descriptor is M2V_PEER_DESC_BYTES of plain data");

inline unsigned long long process_nonce()
{
    static const unsigned long long n = [] {
        std::random_device rd;
        unsigned long long v = ((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ ((unsigned long long)getpid() << 17);
        return v ? v : 1ull;
    }();
    return n;
}

struct PeerState {
    static constexpr uint32_t kMagic = 0x4D325650u;       // "M2VP"
    static constexpr uint32_t kVersion = 2;               // 2: the descriptor carries the process nonce
    static constexpr size_t kCntBytes = (size_t)kPeerCntStride * 4 / 2;                                   // one counter's line: 128 bytes
    static constexpr size_t kCtl = 2 * (size_t)kPeerSlots * 2 * kCntBytes + 1024;                          // the counters, then the give-up word's KB
    int rank = 0, world = 1, device = 0;
    uint8_t *block = nullptr;         // own landing block
    size_t bytes = 0, cap = 0;        // its size; capacity of one landing buffer
    bool fine = false;                // allocated fine-grained (the cross-GPU requirement); false only with M2V_PEER_COARSE=1 (one GPU)
    uint8_t *nb[2] = {nullptr, nullptr};   // the neighbours' blocks as mapped into this process: [0] the rank above, [1] the rank below
    bool nb_ipc[2] = {false, false};
    bool mirror = false;              // solo timing: both "neighbours" are this rank itself, rows come back on the side they left from
    bool connected = false;
    bool degraded = false;            // a wait once ran out of budget: this communicator exchanges through its base from then on
    unsigned long long seq = 0;       // peer sequences so far (selects the counter set)
    unsigned int budget = 20000000u;  // bound of one wait in 10 ns ticks: 200 ms (M2V_PEER_BUDGET_US overrides)
    unsigned long long sequences = 0, giveups = 0;
    int lines_used = 0;               // counter lines (2 per GOP) the longest sequence so far counted on: what a sequence clears of the next set

    static size_t off_cnt(unsigned set, int side) { return ((size_t)set * kPeerSlots * 2 + (size_t)side) * kCntBytes; }     // GOP 0's; GOP g: + g * 2 * kCntBytes
    static size_t off_gaveup() { return kCtl - 1024; }
    size_t off_buf(unsigned parity, int side) const { return kCtl + (size_t)(parity * 2u + (unsigned)side) * cap; }
    // own side: rows that arrived from the rank above (side 0) / below (side 1)
    const uint8_t *got(int side, unsigned parity) const { return block + off_buf(parity, side); }
    const unsigned int *seen(int side, unsigned set) const { return (const unsigned int *)(block + off_cnt(set, side)); }
    unsigned int *gaveup() const { return (unsigned int *)(block + off_gaveup()); }
    // neighbour n (0 above, 1 below): where this rank's rows land there - in ITS buffer "from below" / "from above"
    uint8_t *put(int n, unsigned parity) const { return nb[n] + off_buf(parity, mirror ? n : 1 - n); }
    unsigned int *cnt(int n, unsigned set) const { return (unsigned int *)(nb[n] + off_cnt(set, mirror ? n : 1 - n)); }
};

struct PeerComm final : m2v_comm {
    m2v_comm *base;                   // not owned: sizes, strips, and the halo when the peer form cannot run
    PeerState st;
    std::string kind_text;
    PeerComm(m2v_comm *b, int rank, int device, size_t halo_bytes) : base(b)
    {
        world = b->world;
        st.rank = rank; st.world = b->world; st.device = device;
        kind_text = std::string("peer+") + b->kind();
        M2V_COMM_HIP(hipSetDevice(device));
        st.cap = (halo_bytes + 255) & ~(size_t)255;
        st.bytes = PeerState::kCtl + 4 * st.cap;
        const char *coarse = getenv("M2V_PEER_COARSE");
        st.fine = !(coarse && coarse[0] == '1');
        if (st.fine) M2V_COMM_HIP(hipExtMallocWithFlags((void **)&st.block, st.bytes, hipDeviceMallocFinegrained));
        else M2V_COMM_HIP(hipMalloc((void **)&st.block, st.bytes));
        try {                               // (no destructor runs for a constructor that throws: the block is given back here)
            M2V_COMM_HIP(hipMemset(st.block, 0, st.bytes));
            M2V_COMM_HIP(hipDeviceSynchronize());
        } catch (...) {
            (void)hipFree(st.block);
            st.block = nullptr;
            throw;
        }
        if (const char *b_us = getenv("M2V_PEER_BUDGET_US")) {
            const long long us = atoll(b_us);
            st.budget = (unsigned int)std::min<long long>(std::max<long long>(us, 0) * 100, 0x7FFFFFFFll);
        }
    }
    ~PeerComm() override
    {
        (void)hipSetDevice(st.device);
        (void)hipDeviceSynchronize();
        for (int n = 0; n < 2; ++n)
            if (st.nb[n] && st.nb_ipc[n]) (void)hipIpcCloseMemHandle(st.nb[n]);
        if (st.block) (void)hipFree(st.block);
    }
    PeerState *peer() override { return &st; }
    const char *kind() const override { return kind_text.c_str(); }
    void abort() override { base->abort(); }
    void halo(int r, const void *su, void *ru, const void *sd, void *rd, size_t n, hipStream_t s) override { base->halo(r, su, ru, sd, rd, n, s); }
    void allgather_u64(int r, const unsigned long long *a, unsigned long long *b, size_t c, hipStream_t s) override { base->allgather_u64(r, a, b, c, s); }
    void gather(int r, int dst, const void *p, const size_t *sizes, void *const *bufs, hipStream_t s) override { base->gather(r, dst, p, sizes, bufs, s); }
    void loopback(int r, const void *a, void *b, size_t n, hipStream_t s) override { base->loopback(r, a, b, n, s); }

    void export_desc(PeerDesc &d) const
    {
        memset(&d, 0, sizeof d);
        d.magic = PeerState::kMagic; d.version = PeerState::kVersion;
        d.pid = (int32_t)getpid(); d.device = st.device;
        d.nonce = process_nonce();
        d.ptr = (unsigned long long)(uintptr_t)st.block; d.bytes = st.bytes; d.cap = st.cap;
        M2V_COMM_HIP(hipSetDevice(st.device));
        M2V_COMM_HIP(hipIpcGetMemHandle(&d.ipc, st.block));
    }
    uint8_t *map(const PeerDesc &d, bool &ipc)
    {
        if (d.magic != PeerState::kMagic || d.version != PeerState::kVersion) throw CommError("peer transport: not a landing-block descriptor");
        if (d.cap != st.cap || d.bytes != st.bytes) throw CommError("peer transport: the ranks were created with different halo capacities");
        M2V_COMM_HIP(hipSetDevice(st.device));
        if (d.nonce == process_nonce() && d.pid == (int32_t)getpid()) {     // the same process (by its nonce; pids repeat across pid namespaces): its pointer is good here
            ipc = false;
            if (d.device != st.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, st.device, d.device) != hipSuccess || !can) throw CommError("peer transport: no peer access between the two GPUs");
                const hipError_t pe = hipDeviceEnablePeerAccess(d.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) throw CommError(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe));
                (void)hipGetLastError();
            }
            return (uint8_t *)(uintptr_t)d.ptr;
        }
        void *p = nullptr;
        M2V_COMM_HIP(hipIpcOpenMemHandle(&p, d.ipc, hipIpcMemLazyEnablePeerAccess));
        ipc = true;
        return (uint8_t *)p;
    }
    // the neighbours' descriptors (null where there is none); both null on a rank that has neighbours = solo timing: itself, mirrored
    void connect(const PeerDesc *up, const PeerDesc *down, bool mirror_self)
    {
        if (st.connected) throw CommError("peer transport: already connected");
        if (mirror_self) {
            st.nb[0] = st.nb[1] = st.block;
            st.mirror = true;
        } else {
            if ((st.rank > 0) != (up != nullptr) || (st.rank < world - 1) != (down != nullptr))
                throw CommError("peer transport: a descriptor for every neighbour, and only for neighbours");
            if (up) st.nb[0] = map(*up, st.nb_ipc[0]);
            if (down) st.nb[1] = map(*down, st.nb_ipc[1]);
        }
        st.connected = true;
    }
    // export + all-gather through the base communicator + connect: collective over the base
    void connect_all()
    {
        const std::string bk = base->kind();
        if (bk.rfind("solo", 0) == 0) { connect(nullptr, nullptr, true); return; }
        constexpr size_t kWords = sizeof(PeerDesc) / 8;
        PeerDesc mine;
        export_desc(mine);
        hipStream_t s = nullptr;
        unsigned long long *d_src = nullptr, *d_all = nullptr;
        std::vector<PeerDesc> all((size_t)world);
        M2V_COMM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        try {
            M2V_COMM_HIP(hipMalloc((void **)&d_src, sizeof mine));
            M2V_COMM_HIP(hipMalloc((void **)&d_all, sizeof mine * (size_t)world));
            M2V_COMM_HIP(hipMemcpyAsync(d_src, &mine, sizeof mine, hipMemcpyHostToDevice, s));
            M2V_COMM_HIP(hipStreamSynchronize(s));
            base->allgather_u64(st.rank, d_src, d_all, kWords, s);
            M2V_COMM_HIP(hipStreamSynchronize(s));
            M2V_COMM_HIP(hipMemcpy(all.data(), d_all, sizeof mine * (size_t)world, hipMemcpyDeviceToHost));
        } catch (...) {
            if (d_src) (void)hipFree(d_src);
            if (d_all) (void)hipFree(d_all);
            (void)hipStreamDestroy(s);
            throw;
        }
        (void)hipFree(d_src); (void)hipFree(d_all); (void)hipStreamDestroy(s);
        connect(st.rank > 0 ? &all[(size_t)st.rank - 1] : nullptr, st.rank < world - 1 ? &all[(size_t)st.rank + 1] : nullptr, false);
    }
};

}  // namespace m2v
