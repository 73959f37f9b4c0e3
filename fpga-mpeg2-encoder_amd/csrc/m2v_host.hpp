// m2v_host.hpp — internals shared by the host translation units of libm2v_mi355x.so (not installed; the ABI is include/m2v_mi355x.h).
//
//   m2v_launch.hip    the ONLY unit that includes the device code (m2v_kernels.hpp): constant-table upload and one plain C++ launch
//                     function per kernel (device globals live in the code object of the unit that defines them, so every launch
//                     has to come from there)
//   m2v_core.hip      handle life cycle, options, geometry (RTL:985-1006), the chunk plan (GOP segments, reconstruction slots, launch
//                     lists) and its execution: plan_chunk -> run_step* -> finish_chunk
//   m2v_port.hip      the port path: beats in (RTL:1027-1095), 32-byte words out (RTL:2961-2994), double-buffered staging
//   m2v_resident.hip  whole sequences resident in HBM (what bench.py times), one or several per call
//   m2v_strips.hip    strip mode (BASELINE config c5) and the communicators of m2v_comm.hpp
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/m2v_mi355x.h"
#include "m2v_types.hpp"

namespace m2v {

struct HipError { hipError_t e; const char *what; };

#define HIPCHK(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) throw ::m2v::HipError{_e, #expr};             \
    } while (0)


// Bumped whenever device or pinned memory of any handle is (re)allocated or freed: a recorded hipGraph holds raw pointers, and a
// graph recorded under an older generation is recorded again instead of launched (m2v_strip_encode).
inline std::atomic<unsigned long long> &alloc_generation()
{
    static std::atomic<unsigned long long> gen{0};
    return gen;
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool recorded = true;       // false: no recorded graph ever references this buffer (its reallocation invalidates none)
    void ensure(size_t count)
    {
        if (count <= n) return;
        if (recorded) ++alloc_generation();
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void release() { if (p) { if (recorded) ++alloc_generation(); (void)hipFree(p); } p = nullptr; n = 0; }
};

struct KStat { int launches = 0; double ms = 0, units = 0; };
struct StripFlight;


struct TimedLaunch { hipEvent_t a, b; int kernel; double units; int count; };

}  // namespace m2v

using namespace m2v;      // (an internal header: every unit that includes it is part of the library)

struct m2v_enc {
    // module parameters (RTL:11-14)
    int XL, YL, VL, Q;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // options
    size_t batch_frames = 96;
    bool profile = false;
    int ablate = 0;               // profiling aid, see Geom::ablate
    bool keep_recon = false;      // debug: every frame keeps its own reconstruction buffer, levels are dumped

    // sequence state (RTL:1017-1022)
    enum State { IDLE, DURING, ENDED } state = IDLE;
    Geom g{};
    uint32_t pframes = 0;
    size_t frames_total = 0;      // frames of this sequence handed to the GPU so far
    bool first_chunk = true;

    // Host staging, double buffered: while the GPU works on the chunk submitted from one stage the caller
    // fills the other one.  Each stage owns everything the host and the device touch asynchronously:
    // the pinned frames, the pinned launch plan, the control read-back, the chunk's stream buffer.
    struct HostStage {
        uint8_t *h_in = nullptr;              // pinned planar 4:4:4 frames of the chunk being filled
        size_t h_in_cap = 0;                  // bytes
        StreamCtl *h_ctl = nullptr;           // pinned: [0] read-back, [1] initial values
        FrameJob *h_jobs = nullptr;           // pinned staging of the per-frame jobs
        size_t h_jobs_cap = 0;
        FrameJob *h_joblist = nullptr;        // pinned staging of the jobs in launch-list order
        int *h_lists = nullptr;               // pinned staging of the launch lists
        size_t h_lists_cap = 0;
        uint8_t *h_out = nullptr;             // pinned read-back buffer
        size_t h_out_cap = 0;
        DevBuf<uint8_t> d_in;                 // the chunk's frames on the device: the upload of chunk k+1 (up_stream) runs
                                              // while the kernels of chunk k read the other stage's buffer
        DevBuf<uint8_t> d_out;                // chunk output when it goes to the host
        // Frames that arrived as PACKED 4:4:4 samples (m2v_push_packed) keep the caller's byte order until they are on the device: one
        // linear run of bytes per chunk, frame after frame in arrival order, staged in pinned memory (or uploaded straight from
        // page-locked caller memory) and de-interleaved into d_in by k_unpack444 in front of the chunk's kernels
        struct PkFrame { uint32_t frame; int layout; size_t off; };       // chunk frame index, M2V_PACKED_*, where its bytes start in h_pk / d_pk
        std::vector<PkFrame> pk;
        uint8_t *h_pk = nullptr;              // pinned staging (only when packed beats come from ordinary memory)
        size_t h_pk_cap = 0;
        DevBuf<uint8_t> d_pk;
        size_t pk_used = 0;                   // bytes reserved for the packed frames of the chunk being filled (whole frames)
        size_t pk_valid = 0;                  // ... of which the caller has delivered this many (the frame in progress ends here)
        size_t pk_up = 0;                     // ... of which this many are on the device (or on their way) already
        hipEvent_t ev_ctl = nullptr, ev_out = nullptr, ev_up = nullptr;
        size_t uploaded = 0;                  // leading frames of the chunk being filled that are already in d_in (page-locked
                                              // caller memory goes to the device directly, without the pinned staging copy)
        int stage = 0;                        // 0 free, 1 encode submitted, 2 stream read-back submitted
        bool last = false;
        size_t bytes = 0;
    } hs[2];
    int cur = 0;                  // stage being filled by m2v_push_*
    HostStage &st() { return hs[cur]; }
    std::deque<int> pending;      // submitted stages, oldest first
    int cu_pack = 5;              // option "cu_pack": log2 of the CUs an XCD deals its workgroups to in turn (xcd_remap; 0 = plain XCD remap)
    bool dct_mfma = true;         // option "dct_mfma": luma DCT through the matrix cores (k_mb<.., MFMA = true>); 0 = integer VALU / LDS
                                  // path.  Same results; kept by the rocprofv3 number (profiles/archive/r02_mfma_*: 138.3 vs 140.3 us per launch)
    bool conformant = false;      // option "conformant": ISO reconstruction loop instead of the RTL's (NOT byte-identical to the reference)
    int copy_threads = 8;         // option "copy_threads": threads that copy m2v_push_frames input into pinned memory
    bool direct_upload = true;    // option "direct_upload": page-locked caller memory is uploaded without the staging copy
    bool direct_upload_deferred = false;   // ... = 2: and m2v_push_frames returns while its frames are still being read (see include/m2v_mi355x.h)
    hipEvent_t ev_upl[2] = {nullptr, nullptr}, ev_up2 = nullptr;
    bool upl_pending[2] = {false, false};
    int up_parity = 0;
    // Blocking m2v_push_frames from page-locked memory: the chunk's kernels wait for the upload through an event WITHOUT system fence, and
    // the call waits for that event itself (not for the upload stream) - m2v_port.hip, flush_buffered
    bool up_unsynced = false;            // a direct upload has been issued and not yet waited for on the host
    hipEvent_t up_wait_ev = nullptr;     // the event behind the LAST transfer issued on the upload stream, if one was recorded there (else: wait for the stream)
    void *call_sink = nullptr;           // m2v_push_frames_pull: the call's destination (a PullSink), seen by every progress() inside the call
    int split_streams = 2;        // GOP segments of a chunk run as this many independent groups on as many streams (encode_chunk)
    static constexpr int kMaxSplit = 8;
    hipStream_t side[kMaxSplit - 1] = {};            // group 0 runs on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSplit - 1] = {};
    hipStream_t up_stream = nullptr;     // host -> device uploads of the port path
    hipStream_t up_stream2 = nullptr;    // ... the second one of option direct_upload = 2
    bool async = true;            // option "async": 0 = every chunk is completed before m2v_push_* returns
    hipStream_t copy_stream = nullptr;   // stream read-back, concurrent with the next chunk's kernels
    size_t buffered = 0;          // complete frames waiting in st().h_in
    size_t beat_pos = 0;          // beats received of the frame in progress
    int cur_kind = 0;             // the form the frame in progress is kept in (that of its first beats): 0 = three planes in h_in, 1 + layout = packed
    size_t cur_pk_off = 0;        // ... a packed one: where it starts in the stage's packed bytes
    uint32_t last_frame_valid_beats = 0;   // for a black-filled last frame

    // host output FIFO (32-byte words are handed out by m2v_pull)
    std::vector<uint8_t> fifo;
    size_t fifo_rd = 0;
    bool end_pending = false;     // the data in the FIFO ends with the o_last word

    // device buffers
    DevBuf<int16_t> d_coef;               // debug only: quantised levels
    DevBuf<MbAux> d_mbaux;
    DevBuf<uint32_t> d_slots;             // per-macroblock VLC bit segments (kSlotWords each), used on overflow only
    DevBuf<uint32_t> d_slots_small;       // compact 128-byte slots (kSmallSlotWords each): the common case
    DevBuf<uint32_t> d_mbinfo, d_mblen, d_slice_bytes;
    DevBuf<unsigned long long> d_slice_off, d_frame_off;
    DevBuf<FrameJob> d_jobs;
    DevBuf<int> d_lists;
    DevBuf<FrameJob> d_joblist;           // the jobs again, in launch-list order (k_mb reads its frame's job with ONE dependent scalar load)
    // block -> macroblock tables of the k_mb launches (MbMap), one per launch shape seen, built by the launch functions (m2v_launch.hip)
    struct MbMapKey { int row0, row1, mbw, mbh, cu_pack, mode, rstride, n_edge; };
    struct MbMapCache { MbMapKey key; DevBuf<MbMap> d; hipStream_t filled_on = nullptr; hipEvent_t ev = nullptr; std::vector<hipStream_t> waited; };
    std::deque<MbMapCache> mbmaps;
    DevBuf<StreamCtl> d_ctl;
    int ctl_init = 0;                     // how the next k_frame_scan sets the control word up (ctl_begin): 0 leaves it, 1 new stream, 2 continues
    unsigned long long ctl_cap = 0;
    // strip mode, peer transport: what the next k_frame_scan does about the give-up word and the next sequence's arrival counters (PeerScan)
    unsigned int *scan_peer_gaveup = nullptr, *scan_peer_clear = nullptr;
    int scan_peer_lines = 0;
    unsigned long long scan_peer_mark = 0;
    std::vector<uint8_t *> rec_pool;      // reconstruction buffers (4:2:0 planar), each ysz + 2*csz
    size_t rec_bytes = 0;
    size_t rec_pool_bytes = 0;            // allocation size of every buffer in rec_pool
    int persist_slot = -1;                // slot holding recon of the last encoded frame (GOP continues)
    unsigned long long stream_bytes = 0;  // bytes of the current sequence already moved to the FIFO

    // plan of the chunk being encoded (plan_chunk -> run_step* -> finish_chunk)
    struct Step { int off_i, n_i, off_p, n_p, off_h, n_h; int cut_i[kMaxSplit + 1], cut_p[kMaxSplit + 1]; };   // cut_*[k]: first list entry of segment group k
    // what d_jobs / d_lists / d_joblist hold: a caller that encodes sequence after sequence of one shape from the same buffers (the
    // resident entry in a loop) gets the same plan every time, and three small host-to-device copies in front of the first kernel
    // of every call are ~25 us of latency the GPU spends idle
    std::vector<FrameJob> dev_jobs;
    std::vector<int> dev_lists;
    const void *dev_jobs_p = nullptr, *dev_lists_p = nullptr, *dev_joblist_p = nullptr;
    int plan_groups = 1;                  // groups the launch lists of the current plan are cut into
    int plan_gf[kMaxSplit + 1] = {};      // chunk-frame index where each group's frames start (its GOP segments are consecutive frames)
    bool resident_inflight = false;       // between m2v_encode_resident_begin and m2v_encode_resident_end
    bool resident_empty = false;          // ... of a sequence without frames
    hipStream_t resident_stream = nullptr;
    bool slice_scan_done = false;         // the groups ran k_slice_scan on their own streams (encode_chunk): finish_chunk skips it
    std::vector<Step> plan_steps;
    size_t plan_nf = 0;
    bool strip_active = false;            // between m2v_strip_begin and m2v_strip_finish
    bool strip_inflight = false;          // between m2v_strip_encode_begin and m2v_strip_encode_end
    StripFlight *flight = nullptr;        // ... what the second half needs to know of the first (m2v_strips.hip)
    hipStream_t strip_stream = nullptr;
    DevBuf<uint8_t> d_segs;               // CopySeg table of the strip assembly (written by k_strip_layout)
    DevBuf<unsigned long long> d_frame_pos;   // where every frame's headers start in the assembled stream (k_strip_layout)
    DevBuf<unsigned long long> d_alloff;  // [ranks][frames + 1] frame offsets of every rank's strip
    uint8_t *h_asm = nullptr;             // pinned staging of the offsets (m2v_strip_assemble: up; m2v_strip_encode: the all-gathered sizes down)
    size_t h_asm_cap = 0;
    hipEvent_t ev_asm = nullptr;          // the staging may be rewritten once this has been reached
    uint8_t *h_strip = nullptr;           // pinned: this strip's frame offsets + control word (m2v_strip_finish_async -> m2v_strip_offsets)
    size_t h_strip_cap = 0;
    size_t strip_nf = 0;
    hipEvent_t ev_strip = nullptr;
    // m2v_strip_encode: the whole strip sequence in one call
    DevBuf<uint8_t> d_halo, d_strip_own, d_gather;      // (d_gather: sized from the other ranks' strips AFTER the host wait; no recording references it)
    hipStream_t comm_stream = nullptr;    // send / recv with the neighbours, beside the interior rows on the main stream
    hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_interior = nullptr, ev_done = nullptr;
    // m2v_strip_encode as a recorded hipGraph (option "strip_graph"): everything one call enqueues before its one host wait
    struct StripGraph {
        hipGraphExec_t exec = nullptr;
        std::vector<unsigned long long> key;       // what the recording depends on (shape, ranks, buffers' generation)
        std::vector<unsigned long long> seen;      // the key of the previous call: a shape is recorded when it comes a second time
        bool broken = false;                       // recording failed once on this handle: not tried again
        int launches = 0, captures = 0;
    } strip_graph;
    // -1 = automatic: recorded with world == 1 and with the single-GPU timing communicators; call by call between the ranks of a real
    // RCCL job (a recording with cross-rank ncclSend / ncclRecv inside has never run on hardware: opt in with 1); 0 = never
    int strip_graph_opt = -1;
    struct StripStats { double halo_total_ms = 0, halo_exposed_ms = 0, gather_ms = 0, host_us_per_step = 0, comm_us_per_step = 0; int steps = 0; int graph = 0; int peer = 0; } strip_stats;

    // debug bookkeeping of the last resident encode
    size_t dbg_frames = 0;
    std::vector<int> dbg_rec_slot;

    // profiling
    KStat stats[5];
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> ev_pool;      // timing events, reused from step to step
    size_t ev_used = 0;
    hipEvent_t chain_ev = nullptr;        // stop event of the previous timer while nothing else was enqueued after it
    hipStream_t chain_stream = nullptr;
    // the timer that stopped last and has not recorded its stop event yet: consecutive launches of ONE kind on one stream are timed as
    // one interval (an event between two kernels costs the second one ~3 us: the eight P launches of a sequence read 110.8 us each with
    // an event in every gap and 107 under rocprofv3)
    struct { bool on = false; hipEvent_t a = nullptr; int kernel = 0; double units = 0; int count = 0; hipStream_t s = nullptr; } open_t;
    bool timer_merge = false;             // only where launches follow each other on one stream with nothing in between (encode_chunk's step loop)

    void set_err(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
    }
};

namespace m2v {

// ---- m2v_core.hip ----
Geom make_geom(const m2v_enc *e, uint32_t xs, uint32_t ys);
hipEvent_t pool_event(m2v_enc *e);
void collect_timers(m2v_enc *e);
void plan_chunk(m2v_enc *e, hipStream_t s, const uint8_t *d_frames, size_t nf, bool last, uint32_t last_valid_beats);
void run_step(m2v_enc *e, hipStream_t s, size_t j);
void run_step_rows(m2v_enc *e, hipStream_t s, size_t j, int r0, int r1);
void run_step_edges_fused(m2v_enc *e, hipStream_t s, size_t j, uint8_t *up, uint8_t *down, const uint8_t *nb_up, const uint8_t *nb_down);
void run_step_peer(m2v_enc *e, hipStream_t s, size_t j, int group, uint8_t *put_up, uint8_t *put_down, const uint8_t *got_up, const uint8_t *got_down, PeerStep ps);
void finish_chunk(m2v_enc *e, hipStream_t s, bool first, bool last, uint8_t *d_stream, bool advance = false);
void encode_chunk(m2v_enc *e, hipStream_t s, const uint8_t *d_frames, size_t nf, bool first, bool last, uint32_t last_valid_beats,
                  uint8_t *d_stream, bool advance = false);
// pinned host memory of at least `bytes`, kept with the handle
void ensure_pinned(uint8_t *&p, size_t &cap, size_t bytes);
// every C-ABI entry runs its body through this: selects the handle's device, turns exceptions into M2V_E_* + the handle's error text
int guard(m2v_enc *e, int (*fn)(m2v_enc *, void *), void *arg);

// HIP-event timers of option "profile": events come from a pool that lives as long as the handle, and a timer
// that starts right where the previous one stopped (same stream, nothing enqueued in between) reuses that
// event, so a step of n back-to-back launches costs n + 1 event records and no create / destroy.
// closes the open timer: its stop event goes into its stream HERE (call it before anything untimed is enqueued there)
inline void timer_close(m2v_enc *e)
{
    if (!e->open_t.on) return;
    hipEvent_t b = pool_event(e);
    HIPCHK(hipEventRecord(b, e->open_t.s));
    e->timed.push_back(TimedLaunch{e->open_t.a, b, e->open_t.kernel, e->open_t.units, e->open_t.count});
    e->chain_ev = b;
    e->chain_stream = e->open_t.s;
    e->open_t.on = false;
}
// ... and the next timer records a start event of its own (untimed work follows)
inline void timer_break(m2v_enc *e)
{
    if (e->profile) timer_close(e);
    e->chain_ev = nullptr;
}

struct Timer {
    m2v_enc *e; hipStream_t s; int kernel; double units; hipEvent_t a = nullptr; bool merged = false;
    Timer(m2v_enc *e_, hipStream_t s_, int k, double u) : e(e_), s(s_), kernel(k), units(u)
    {
        if (e->profile) {
            if (e->open_t.on && e->open_t.kernel == kernel && e->open_t.s == s) { merged = true; return; }     // one more launch of the open interval
            timer_close(e);
            if (e->chain_ev && e->chain_stream == s) a = e->chain_ev;
            else { a = pool_event(e); HIPCHK(hipEventRecord(a, s)); }
            e->chain_ev = nullptr;
        }
    }
    void stop()
    {
        if (e->profile) {
            if (merged) { e->open_t.units += units; ++e->open_t.count; return; }
            e->open_t.on = true; e->open_t.a = a; e->open_t.kernel = kernel; e->open_t.units = units; e->open_t.count = 1; e->open_t.s = s;
            if (!e->timer_merge) timer_close(e);        // everywhere else the stop event follows its launch at once
        }
    }
};

// ---- m2v_strips.hip ----
void strip_flight_release(m2v_enc *e);

// ---- m2v_launch.hip: everything that touches device code ----
// The constant tables live in each device's copy of the code object: uploaded once per device, whichever thread creates the first
// handle there (config c4 creates 8 handles from 8 threads).
void upload_tables(int device);
template <bool P> void launch_mb(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g);
template <bool P> void launch_mb_edges(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g, uint8_t *up, uint8_t *down,
                                       const uint8_t *nb_up, const uint8_t *nb_down);
template <bool P> void launch_mb_peer(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g, uint8_t *put_up, uint8_t *put_down,
                                      const uint8_t *got_up, const uint8_t *got_down, const PeerStep &ps);
extern template void launch_mb_peer<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *, const PeerStep &);
extern template void launch_mb_peer<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *, const PeerStep &);
extern template void launch_mb<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &);
extern template void launch_mb<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &);
extern template void launch_mb_edges<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *);
extern template void launch_mb_edges<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *);
// start of a chunk's stream: the bytes of the sequence that precede it are the previous chunk's prior + total (still in *ctl: one
// stream, in order); only the padding rule needs them.  No launch and no copy: the chunk's k_frame_scan sets the control word up
// itself (its ctl_init argument), this only notes how
void ctl_begin(m2v_enc *e, unsigned long long cap, bool first);
void launch_plan_upload(m2v_enc *e, hipStream_t s, const FrameJob *h_jobs, size_t nf, const int *h_lists, const FrameJob *h_joblist, size_t nlist);
// neighbour-dependent codes + bit offsets of the slices of frames [f0, f1) of the chunk (one block per slice)
void launch_slice_scan(m2v_enc *e, hipStream_t s, const Geom &g, int f0, int f1);
void launch_frame_scan(m2v_enc *e, hipStream_t s, const Geom &g, size_t nf, bool first, bool last, bool advance, uint8_t *d_stream);
void launch_assemble(m2v_enc *e, hipStream_t s, const Geom &g, size_t nf, bool first, bool last, uint8_t *d_stream);
void launch_halo_pack(m2v_enc *e, hipStream_t s, const int *d_list, int count, uint8_t *up, uint8_t *down);
void launch_halo_unpack(m2v_enc *e, hipStream_t s, const int *d_list, int count, const uint8_t *from_up, const uint8_t *from_down);
// strip mode, output rank: where every (frame, rank) piece goes + the copy itself, headers and trailer (k_strip_layout, k_strip_assemble)
void launch_strip_assemble(m2v_enc *e, hipStream_t s, const Geom &g, uint32_t gop, size_t nf, int nranks, const StripSrc &src,
                           const unsigned long long *d_all_off, uint8_t *d_out, unsigned long long cap);
int debug_table(int which, int i, int j);

}  // namespace m2v
