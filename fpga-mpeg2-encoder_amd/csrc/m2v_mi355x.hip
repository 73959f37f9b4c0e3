// m2v_mi355x.hip — host side of libm2v_mi355x.so: the mpeg2encoder port contract
// (RTL/mpeg2encoder.v:10-38) as a C-ABI over the HIP kernels in m2v_kernels.hpp.
//
// Sequence control mirrors stage A of the RTL (RTL:1027-1095): the configuration is latched on
// the first beat, beats fill raster-order frames, i_sequence_stop black-fills the frame in
// progress, and the stream ends with sequence_end_code + one final zero-padded 32-byte word.
// Unlike the 64-clock/macroblock RTL pipeline, frames are buffered and encoded in chunks:
// closed GOPs (closed_gop = 1, RTL:2656) are independent, so frame k of every GOP in a chunk
// runs in the same launch — that is what fills 256 CUs with one wavefront per macroblock.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <deque>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/m2v_mi355x.h"
#include "m2v_kernels.hpp"
#include "m2v_comm.hpp"

using namespace m2v;

namespace {

struct HipError { hipError_t e; const char *what; };

#define HIPCHK(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) throw HipError{_e, #expr};                    \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    void ensure(size_t count)
    {
        if (count <= n) return;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        HIPCHK(hipMalloc((void **)&p, count * sizeof(T)));
        n = count;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct KStat { int launches = 0; double ms = 0, units = 0; };

// why the last m2v_create on this thread failed: there is no handle yet to carry the text (m2v_last_error(NULL))
thread_local std::string t_create_err;
thread_local std::string t_comm_err;          // the same for the m2v_comm_* constructors

struct TimedLaunch { hipEvent_t a, b; int kernel; double units; };

}  // namespace

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
    bool dct_mfma = true;         // option "dct_mfma": luma DCT through the matrix cores (k_mb<.., MFMA = true>); 0 = integer VALU / LDS
                                  // path.  Same results; kept by the rocprofv3 number (profiles/r02_mfma_*: 138.3 vs 140.3 us per launch)
    bool conformant = false;      // option "conformant": ISO reconstruction loop instead of the RTL's (NOT byte-identical to the reference)
    int copy_threads = 8;         // option "copy_threads": threads that copy m2v_push_frames input into pinned memory
    bool direct_upload = true;    // option "direct_upload": page-locked caller memory is uploaded without the staging copy
    int split_streams = 2;        // GOP segments of a chunk run as this many independent groups on as many streams (encode_chunk)
    static constexpr int kMaxSplit = 8;
    hipStream_t side[kMaxSplit - 1] = {};            // group 0 runs on the caller's stream
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSplit - 1] = {};
    hipStream_t up_stream = nullptr;     // host -> device uploads of the port path
    bool async = true;            // option "async": 0 = every chunk is completed before m2v_push_* returns
    hipStream_t copy_stream = nullptr;   // stream read-back, concurrent with the next chunk's kernels
    size_t buffered = 0;          // complete frames waiting in st().h_in
    size_t beat_pos = 0;          // beats received of the frame in progress
    uint32_t last_frame_valid_beats = 0;   // for a black-filled last frame

    // host output FIFO (32-byte words are handed out by m2v_pull)
    std::vector<uint8_t> fifo;
    size_t fifo_rd = 0;
    bool end_pending = false;     // the data in the FIFO ends with the o_last word

    // device buffers
    DevBuf<int16_t> d_coef;               // debug only: quantised levels
    DevBuf<MbAux> d_mbaux;
    DevBuf<MbDepRec> d_mbdep;             // neighbour-dependent codes of every macroblock (k_slice_scan -> k_assemble)
    DevBuf<uint32_t> d_slots;             // per-macroblock VLC bit segments (kSlotWords each), used on overflow only
    DevBuf<uint32_t> d_slots_small;       // compact 128-byte slots (kSmallSlotWords each): the common case
    DevBuf<uint32_t> d_mbinfo, d_mblen, d_mboff, d_slice_bytes;
    DevBuf<unsigned long long> d_slice_off, d_frame_off;
    DevBuf<FrameJob> d_jobs;
    DevBuf<int> d_lists;
    DevBuf<FrameJob> d_joblist;           // the jobs again, in launch-list order (k_mb reads its frame's job with ONE dependent scalar load)
    DevBuf<StreamCtl> d_ctl;
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
    DevBuf<uint8_t> d_halo, d_strip_own, d_gather;
    hipStream_t comm_stream = nullptr;    // send / recv with the neighbours, beside the interior rows on the main stream
    hipEvent_t ev_edges = nullptr, ev_halo = nullptr, ev_interior = nullptr, ev_done = nullptr;
    struct StripStats { double halo_total_ms = 0, halo_exposed_ms = 0, gather_ms = 0, host_us_per_step = 0, comm_us_per_step = 0; int steps = 0; } strip_stats;

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

namespace {

// ---------------------------------------------------------------------------------------------
// geometry (RTL:985-1006)
// ---------------------------------------------------------------------------------------------
int clamp_size16(uint32_t s, int L)
{
    const uint32_t lim = 1u << L;
    if (s > lim) return (int)lim - 1;
    if (s < 4) return 3;
    return (int)s - 1;
}

Geom make_geom(const m2v_enc *e, uint32_t xs, uint32_t ys)
{
    Geom g{};
    xs &= (2u << e->XL) - 1u;       // the ports are XL+1 / YL+1 bits wide (RTL:20-21)
    ys &= (2u << e->YL) - 1u;
    g.mbw = clamp_size16(xs, e->XL) + 1;
    g.mbh = clamp_size16(ys, e->YL) + 1;
    g.W = 16 * g.mbw;
    g.H = 16 * g.mbh;
    g.cw = g.W / 2;
    g.ch = g.H / 2;
    g.Q = e->Q;
    g.mbs = g.mbw * g.mbh;
    g.ysz = (uint32_t)g.W * g.H;
    g.csz = (uint32_t)g.cw * g.ch;
    g.row0 = 0;
    g.row1 = g.mbh;
    g.strip = 0;
    g.ablate = e->ablate;
    geom_finish(g);
    return g;
}

// The constant tables live in each device's copy of the code object: uploaded once per device, whichever thread
// creates the first handle there (config c4 creates 8 handles from 8 threads).  call_once leaves the flag unset when
// the upload throws, so a later m2v_create retries.
constexpr int kMaxDevices = 64;
std::once_flag g_tables_once[kMaxDevices];

void upload_tables_now()
{
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_dct), kDctBasis, sizeof kDctBasis));
    int8_t dct_neg[64];
    for (int i = 0; i < 64; ++i) dct_neg[i] = (int8_t)-kDctBasis[i];       // |basis| <= 89: the negative fits
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_dct_neg), dct_neg, sizeof dct_neg));
    int32_t dct32[64];
    for (int i = 0; i < 64; ++i) dct32[i] = kDctBasis[i];
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_dct32), dct32, sizeof dct32));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_intra_w), kIntraW, sizeof kIntraW));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_zigzag), kZigzagPos, sizeof kZigzagPos));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_motion_code), kMotionCode, sizeof kMotionCode));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_cbp_code), kCbpCode, sizeof kCbpCode));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_dc_code), kDcSizeCode, sizeof kDcSizeCode));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_dc_len), kDcSizeLen, sizeof kDcSizeLen));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_ac_code), kAcCode, sizeof kAcCode));
    // per-call staging (this function runs once per DEVICE, possibly on several threads at once: nothing shared, nothing static);
    // the copies below are synchronised before it goes out of scope
    std::vector<uint16_t> ac2v(2 * kAcRuns * kAcLevels, 0);
    uint16_t *const ac2 = ac2v.data();
    const size_t ac2_bytes = ac2v.size() * sizeof(uint16_t);
    for (int bank = 0; bank < 2; ++bank)
        for (int run = 0; run < 32; ++run)
            for (int lev = 1; lev <= 40; ++lev) ac2[(bank * kAcRuns + run) * kAcLevels + lev - 1] = kAcCode[run * 40 + lev - 1];
    ac2[kAcRuns * kAcLevels] = (1u << 8) | 1u;               // bank 1, run 0, level 1: '1' + sign
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_ac_code2), ac2, ac2_bytes));
    uint32_t recip[64];
    for (int i = 0; i < 64; ++i) recip[i] = ((1u << 21) + kIntraW[i] - 1u) / kIntraW[i];      // ceil(2^21 / W)
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_intra_recip), recip, sizeof recip));
    // per-lane operands of the DCT-as-GEMM variant (k_mb<.., MFMA = true>, see MfmaLane)
    MfmaLane ml[64];
    MfmaLaneIntra mi[64];
    SearchLane sl[64];
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        MfmaLane &m = ml[lane];
        MfmaLaneIntra &n = mi[lane];
        memset(&m, 0, sizeof m);
        memset(&n, 0, sizeof n);
        if ((c >> 3) == (g & 1))
            for (int b = 0; b < 8; ++b) {
                const int8_t w = (int8_t)(g < 2 ? kDctBasis[(c & 7) * 8 + b] : -kDctBasis[(c & 7) * 8 + b]);
                m.b1[b >> 2] |= (uint32_t)(uint8_t)w << (8 * (b & 3));
            }
        if ((c >> 3) == (g >> 1))
            for (int b = 0; b < 4; ++b) m.a2[0] |= (uint32_t)(uint8_t)kDctBasis[(c & 7) * 8 + 4 * (g & 1) + b] << (8 * b);
        m.a2[3] = m.a2[0];
        const int tile = ((g >> 1) << 1) | (c >> 3);
        for (int v = 0; v < 4; ++v) {
            const int raster = (4 * (g & 1) + v) * 8 + (c & 7);
            m.zoff[v] = (uint32_t)(tile * 128 + kZigzagPos[raster] * 2);
            n.wq |= (uint32_t)kIntraW[raster] << (8 * v);
            n.recip[v] = recip[raster];
        }
        // the reference pairs (w0,w1) (w2,w3) start at dword gq of window row dy', the pairs (w1,w2) (w3,w4) at gq + 1: one of the
        // two starts is even in copy A, the other in copy B (which holds dword j + 1 at index j); VECTOR_LEVEL 3 geometry
        const int dyi = s3_dy(lane), gq = s3_group(lane), gap = win_b_gap(16 + 4 * 3);
        SearchLane &q = sl[lane];
        memset(&q, 0, sizeof q);
        // candidate j of the lane has dx = 4 gq - 8 + j
        const uint32_t cbase = 255u - (uint32_t)((dyi << 4) | (4 * gq));
        q.cb4 = cbase | ((cbase - 1u) << 8) | ((cbase - 2u) << 16) | ((cbase - 3u) << 24);
        q.dead_lo = dyi > 12 || gq == 0 ? 0xFFFFFFFFu : 0u;                    // dx = -8, -7; the helper lanes own no candidates
        q.dead_hi = dyi > 12 ? 0xFFFFFFFFu : gq == 3 ? 0xFFFF0000u : 0u;       // dx = +7
        q.even = (uint32_t)kS3Win + 4u * (uint32_t)((gq & 1) ? gap + dyi * kWinStride + gq - 1 : dyi * kWinStride + gq);
        q.odd = (uint32_t)kS3Win + 4u * (uint32_t)((gq & 1) ? dyi * kWinStride + gq + 1 : gap + dyi * kWinStride + gq);
        if (dyi <= 12) {                                         // owner of the candidates (dy', 4 gq - 8 .. + 3)
            const int t = dyi % 3, k = dyi / 3;
            q.cur = (uint32_t)kS3Cur;
            q.cur12 = (uint32_t)kS3Cur + 12 * 16;
            q.plus = dyi < 12 ? (uint32_t)kS3Flush + 8u * (uint32_t)((t * 4 + gq) * 4 + k) : (uint32_t)kS3Sum12 + 8u * (uint32_t)gq;
            q.minus = dyi < 12 && k > 0 ? q.plus - 8u : (uint32_t)kS3Zero;
        } else {                                                 // helper t = dy' - 13: rows 13..15 of dy' = t, t + 3, t + 6, t + 9, one row of 12
            const int t = dyi - 13;
            q.cur = (uint32_t)kS3Rep;
            q.cur12 = (uint32_t)kS3Cur + (uint32_t)(13 + t) * 16;
            q.plus = (uint32_t)kS3Flush + 8u * (uint32_t)((t * 4 + gq) * 4);
            q.minus = (uint32_t)kS3Sum12 + 8u * (uint32_t)gq;
        }
    }
    // quad-major on the device (see MfmaLane): [quad][lane][4 dwords]
    typedef uint32_t Quad[64][4];
    std::vector<uint32_t> slqv(sizeof(SearchLane) / 16 * 64 * 4), mlqv(sizeof(MfmaLane) / 16 * 64 * 4);
    Quad *const slq = (Quad *)slqv.data(), *const mlq = (Quad *)mlqv.data();
    const size_t slq_bytes = slqv.size() * 4, mlq_bytes = mlqv.size() * 4;
    uint32_t dcl[12];
    for (int i = 0; i < 12; ++i) dcl[i] = (uint32_t)kDcSizeCode[0][i] | ((uint32_t)kDcSizeLen[0][i] << 16);
    for (int lane = 0; lane < 64; ++lane) {
        for (size_t q = 0; q < sizeof(SearchLane) / 16; ++q) memcpy(slq[q][lane], (const uint8_t *)&sl[lane] + 16 * q, 16);
        for (size_t q = 0; q < sizeof(MfmaLane) / 16; ++q) memcpy(mlq[q][lane], (const uint8_t *)&ml[lane] + 16 * q, 16);
    }
    for (int vl = 0; vl < 3; ++vl)
        for (int pf = 0; pf < 2; ++pf) {
            const size_t blk = ((size_t)(vl * 2 + pf) * kQuadsPerBlock) * 64 * 16;
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), slq, slq_bytes, blk + (size_t)kQuadSearch0 * 64 * 16));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), mlq, mlq_bytes, blk + (size_t)kQuadMfma0 * 64 * 16));
            const size_t cst = blk + (size_t)kQuadConst0 * 64 * 16;
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), dct32, sizeof dct32, cst + kConstDct32));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), kDctBasis, sizeof kDctBasis, cst + kConstDct));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), dct_neg, sizeof dct_neg, cst + kConstDctNeg));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), kCbpCode, sizeof kCbpCode, cst + kConstCbp));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), dcl, sizeof dcl, cst + kConstDcLuma));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), ac2, ac2_bytes, blk + (size_t)kQuadAc0 * 64 * 16));
        }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_mfma_intra), mi, sizeof mi));
    HIPCHK(hipDeviceSynchronize());         // the copies read stack arrays: complete before they go out of scope
    // the lane tables d_lanek[VL - 1][P]: every instantiation of the macroblock kernel writes its own (LaneK, FILL = true); they
    // read the constant tables uploaded above
    {
        Geom g0{};
        const dim3 one(1), wave(64);
#define M2V_FILL(VLV, PV) hipLaunchKernelGGL((k_mb<VLV, PV, false, false, true>), one, wave, 0, 0, (const FrameJob *)nullptr, (const int *)nullptr, g0, \
                                             (uint32_t *)nullptr, (MbAux *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int16_t *)nullptr)
        M2V_FILL(1, false); M2V_FILL(1, true); M2V_FILL(2, false); M2V_FILL(2, true); M2V_FILL(3, false); M2V_FILL(3, true);
#undef M2V_FILL
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
}

void upload_tables(int device)
{
    if (device >= 0 && device < kMaxDevices) std::call_once(g_tables_once[device], upload_tables_now);
    else upload_tables_now();
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
// HIP-event timers of option "profile": events come from a pool that lives as long as the handle, and a timer
// that starts right where the previous one stopped (same stream, nothing enqueued in between) reuses that
// event, so a step of n back-to-back launches costs n + 1 event records and no create / destroy.
hipEvent_t pool_event(m2v_enc *e)
{
    if (e->ev_used == e->ev_pool.size()) {
        hipEvent_t ev = nullptr;
        HIPCHK(hipEventCreate(&ev));
        e->ev_pool.push_back(ev);
    }
    return e->ev_pool[e->ev_used++];
}

struct Timer {
    m2v_enc *e; hipStream_t s; int kernel; double units; hipEvent_t a = nullptr, b = nullptr;
    Timer(m2v_enc *e_, hipStream_t s_, int k, double u) : e(e_), s(s_), kernel(k), units(u)
    {
        if (e->profile) {
            if (e->chain_ev && e->chain_stream == s) a = e->chain_ev;
            else { a = pool_event(e); HIPCHK(hipEventRecord(a, s)); }
            e->chain_ev = nullptr;
        }
    }
    void stop()
    {
        if (e->profile) {
            b = pool_event(e);
            HIPCHK(hipEventRecord(b, s));
            e->timed.push_back(TimedLaunch{a, b, kernel, units});
            e->chain_ev = b;
            e->chain_stream = s;
        }
    }
};

void collect_timers(m2v_enc *e)
{
    for (auto &t : e->timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            e->stats[t.kernel].launches++;
            e->stats[t.kernel].ms += ms;
            e->stats[t.kernel].units += t.units;
        }
    }
    e->timed.clear();
    e->ev_used = 0;
    e->chain_ev = nullptr;
}

// strip mode, m2v_strip_encode: the strip's first and last macroblock row in ONE launch of the EDGE instantiation, which also
// writes their outer rows of the reconstruction into the halo buffers (no pack kernel).  gg: row0 = first row, rstride = distance
// to the last one, row1 - row0 = 1 or 2 local rows.
template <bool P>
void launch_mb_edges(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g, uint8_t *up, uint8_t *down,
                     const uint8_t *nb_up, const uint8_t *nb_down)
{
    if (count <= 0) return;
    const dim3 grid((unsigned)((size_t)count * (size_t)(g.row1 - g.row0) * g.mbw)), block(64);
    Timer t(e, s, P ? 0 : 1, (double)count * (g.row1 - g.row0) * g.mbw * 256.0);
    int16_t *dbg = e->keep_recon ? e->d_coef.p : nullptr;
    const FrameJob *const jl = e->d_joblist.p + (d_list - e->d_lists.p);
#define M2V_LAUNCH_EDGE(VLV) \
    hipLaunchKernelGGL((k_mb<VLV, P, false, true, false, true>), grid, block, 0, s, jl, d_list, g, e->d_mbinfo.p, e->d_mbaux.p, \
                       e->d_slots_small.p, e->d_slots.p, dbg, up, down, nb_up, nb_down)
    switch (e->VL) {
        case 1: M2V_LAUNCH_EDGE(1); break;
        case 2: M2V_LAUNCH_EDGE(2); break;
        default: M2V_LAUNCH_EDGE(3); break;
    }
#undef M2V_LAUNCH_EDGE
    HIPCHK(hipGetLastError());
    t.stop();
}

template <bool P>
void launch_mb(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g)
{
    if (count <= 0) return;
    const dim3 grid((unsigned)((size_t)count * (size_t)(g.row1 - g.row0) * g.mbw)), block(64);      // one wavefront per macroblock
    Timer t(e, s, P ? 0 : 1, (double)count * g.ysz);
    int16_t *dbg = e->keep_recon ? e->d_coef.p : nullptr;
    const FrameJob *const jl = e->d_joblist.p + (d_list - e->d_lists.p);      // the same launch list, as jobs
#define M2V_LAUNCH_MB(VLV, PV, CV) \
    do { \
        if (e->dct_mfma && !(CV)) \
            hipLaunchKernelGGL((k_mb<VLV, PV, false, true>), grid, block, 0, s, jl, d_list, g, e->d_mbinfo.p, e->d_mbaux.p, \
                               e->d_slots_small.p, e->d_slots.p, dbg); \
        else \
            hipLaunchKernelGGL((k_mb<VLV, PV, CV, false>), grid, block, 0, s, jl, d_list, g, e->d_mbinfo.p, e->d_mbaux.p, \
                               e->d_slots_small.p, e->d_slots.p, dbg); \
    } while (0)
    if (P) {
        if (e->conformant) {
            switch (e->VL) {
                case 1: M2V_LAUNCH_MB(1, true, true); break;
                case 2: M2V_LAUNCH_MB(2, true, true); break;
                default: M2V_LAUNCH_MB(3, true, true); break;
            }
        } else {
            switch (e->VL) {
                case 1: M2V_LAUNCH_MB(1, true, false); break;
                case 2: M2V_LAUNCH_MB(2, true, false); break;
                default: M2V_LAUNCH_MB(3, true, false); break;
            }
        }
    } else {
        if (e->conformant) M2V_LAUNCH_MB(1, false, true);
        else M2V_LAUNCH_MB(1, false, false);
    }
#undef M2V_LAUNCH_MB
    HIPCHK(hipGetLastError());
    t.stop();
}

// ---------------------------------------------------------------------------------------------
// A chunk of `nf` consecutive frames of the current sequence is encoded in three parts:
//   plan_chunk   per-frame jobs, GOP segments, reconstruction slots, launch lists, device buffers
//   run_step(j)  macroblock kernel for the j-th frame of every GOP segment (frame f+1 needs recon(f))
//   finish_chunk scans, headers, stream assembly into `d_stream` (ctl carries base/total/cap)
// Everything is enqueued on `s`; nothing is synchronised here.
// ---------------------------------------------------------------------------------------------
void plan_chunk(m2v_enc *e, hipStream_t s, const uint8_t *d_frames, size_t nf, bool last, uint32_t last_valid_beats)
{
    e->chain_ev = nullptr;                  // copies are enqueued below: the next timer records its own start event
    const Geom &g = e->g;
    const size_t frame_bytes = (size_t)g.ysz * 3;
    const uint32_t bpf = g.ysz / 4;
    const uint32_t gop = e->pframes + 1u;

    // ---- per-frame jobs, GOP segments, reconstruction slots ----
    std::vector<FrameJob> jobs(nf);
    std::vector<int> seg_start;                   // chunk-frame index where each GOP segment starts
    for (size_t k = 0; k < nf; ++k) {
        const size_t n = e->frames_total + k;
        jobs[k].in = d_frames + k * frame_bytes;
        jobs[k].i_frame = (int32_t)(n % gop);
        jobs[k].n = (uint32_t)n;
        jobs[k].valid_beats = (last && k == nf - 1) ? last_valid_beats : bpf;
        jobs[k].ref = nullptr;
        jobs[k].rec = nullptr;
        jobs[k].fidx = 0;
        jobs[k].hidx = -1;
        jobs[k].rhidx = -1;
        if (k == 0 || jobs[k].i_frame == 0) seg_start.push_back((int)k);
    }
    const size_t nseg = seg_start.size();
    e->rec_bytes = (size_t)g.ysz + 2 * (size_t)g.csz;
    const bool need_any_rec = e->pframes > 0;
    std::vector<int> rec_slot(nf, -1);
    if (need_any_rec) {
        if (e->rec_pool_bytes < e->rec_bytes) {       // geometry grew since the pool was allocated (new sequence)
            if (e->persist_slot >= 0) throw HipError{hipErrorInvalidValue, "reference lost on geometry change"};
            for (auto p : e->rec_pool) (void)hipFree(p);
            e->rec_pool.clear();
            e->rec_pool_bytes = e->rec_bytes;
        }
        const size_t want = e->keep_recon ? nf + 1 : 2 * nseg + 1;
        while (e->rec_pool.size() < want) {
            uint8_t *p = nullptr;
            HIPCHK(hipMalloc((void **)&p, e->rec_pool_bytes));
            e->rec_pool.push_back(p);
        }
        std::vector<int> free_slots;
        for (int i = 0; i < (int)e->rec_pool.size(); ++i) if (i != e->persist_slot) free_slots.push_back(i);
        size_t fs = 0;
        for (size_t sg = 0; sg < nseg; ++sg) {
            const size_t a = seg_start[sg], b = sg + 1 < nseg ? (size_t)seg_start[sg + 1] : nf;
            int slots[2] = {-1, -1};
            for (size_t k = a; k < b; ++k) {
                // a frame's reconstruction is needed iff a P frame of the same GOP follows (ref(f+1) = recon(f))
                const bool known_last = last && k == nf - 1;
                const bool followed = (uint32_t)jobs[k].i_frame < e->pframes && !known_last;
                int prev = (k == a) ? (jobs[k].i_frame != 0 ? e->persist_slot : -1) : rec_slot[k - 1];
                if (jobs[k].i_frame != 0) {
                    if (prev < 0) throw HipError{hipErrorInvalidValue, "P frame without a reference"};
                    jobs[k].ref = e->rec_pool[prev];
                }
                if (followed) {
                    int sl;
                    if (e->keep_recon) sl = free_slots[fs++];
                    else {
                        const int which = (int)((k - a) & 1);
                        if (slots[which] < 0) slots[which] = free_slots[fs++];
                        sl = slots[which];
                    }
                    rec_slot[k] = sl;
                    jobs[k].rec = e->rec_pool[sl];
                }
            }
        }
        e->persist_slot = rec_slot[nf - 1];
    }

    // ---- launch lists: step j = j-th frame of every segment; I and P frames in separate launches;
    //      halo list = frames of the step whose reconstruction is referenced later (strip mode) ----
    size_t maxlen = 0;
    for (size_t sg = 0; sg < nseg; ++sg) {
        const size_t a = seg_start[sg], b = sg + 1 < nseg ? (size_t)seg_start[sg + 1] : nf;
        maxlen = std::max(maxlen, b - a);
    }
    std::vector<int> lists;
    e->plan_steps.assign(maxlen, m2v_enc::Step{});
    // segment group of a GOP segment (option "split_streams"): contiguous runs of segments, group g on stream g
    const int groups = (int)std::max<size_t>(1, std::min<size_t>({(size_t)e->split_streams, nseg, (size_t)m2v_enc::kMaxSplit}));
    e->plan_groups = groups;
    auto group_of = [&](size_t sg) { return (int)(sg * (size_t)groups / nseg); };
    for (int k = 0; k <= m2v_enc::kMaxSplit; ++k) e->plan_gf[k] = (int)nf;
    for (size_t sg = nseg; sg-- > 0;) e->plan_gf[group_of(sg)] = seg_start[sg];       // first segment of every group (descending: the first one wins)
    e->slice_scan_done = false;
    for (size_t j = 0; j < maxlen; ++j) {
        m2v_enc::Step st{};
        for (int pass = 0; pass < 3; ++pass) {
            const int off = (int)lists.size();
            int cut[m2v_enc::kMaxSplit + 1];
            int gnext = 0;
            for (size_t sg = 0; sg < nseg; ++sg) {
                const size_t a = seg_start[sg], b = sg + 1 < nseg ? (size_t)seg_start[sg + 1] : nf;
                while (gnext <= group_of(sg)) cut[gnext++] = (int)lists.size() - off;      // the lists are in segment order
                if (a + j >= b) continue;
                const FrameJob &fj = jobs[a + j];
                if ((pass == 0 && fj.i_frame == 0) || (pass == 1 && fj.i_frame != 0) || (pass == 2 && fj.rec != nullptr)) {
                    if (pass == 2) {                                   // its place in the step's halo buffers; the next frame's reference
                        jobs[a + j].hidx = (int32_t)((int)lists.size() - off);
                        if (a + j + 1 < b) jobs[a + j + 1].rhidx = jobs[a + j].hidx;
                    }
                    lists.push_back((int)(a + j));
                }
            }
            const int cnt = (int)lists.size() - off;
            while (gnext <= m2v_enc::kMaxSplit) cut[gnext++] = cnt;
            if (pass == 0) { st.off_i = off; st.n_i = cnt; memcpy(st.cut_i, cut, sizeof cut); }
            else if (pass == 1) { st.off_p = off; st.n_p = cnt; memcpy(st.cut_p, cut, sizeof cut); }
            else { st.off_h = off; st.n_h = cnt; }
        }
        e->plan_steps[j] = st;
    }

    // ---- device buffers ----
    const size_t nmb = nf * (size_t)g.mbs;
    e->d_jobs.ensure(nf);
    e->d_lists.ensure(lists.size());
    e->d_joblist.ensure(lists.size());
    if (e->keep_recon) e->d_coef.ensure(nmb * 384);
    e->d_mbaux.ensure(nmb);
    e->d_mbdep.ensure(nmb);
    e->d_slots.ensure(nmb * (size_t)kSlotWords + 8);
    e->d_slots_small.ensure(nmb * (size_t)(kSmallSlotWords + kTinySlotWords) + 8);      // the 128-byte class, then the 64-byte class
    e->g.s16_off = (uint32_t)(nmb * (size_t)kSmallSlotWords);
    e->d_mbinfo.ensure(nmb);
    e->d_mblen.ensure(nmb);
    e->d_mboff.ensure(nmb);
    e->d_slice_bytes.ensure(nf * g.mbh);
    e->d_slice_off.ensure(nf * g.mbh);
    e->d_frame_off.ensure(nf + 1);
    // pinned staging: the caller synchronises the stream before the next chunk reuses it
    if (e->st().h_jobs_cap < nf) {
        if (e->st().h_jobs) (void)hipHostFree(e->st().h_jobs);
        e->st().h_jobs = nullptr; e->st().h_jobs_cap = 0;
        HIPCHK(hipHostMalloc((void **)&e->st().h_jobs, nf * sizeof(FrameJob)));
        e->st().h_jobs_cap = nf;
    }
    if (e->st().h_lists_cap < lists.size()) {
        if (e->st().h_lists) (void)hipHostFree(e->st().h_lists);
        e->st().h_lists = nullptr; e->st().h_lists_cap = 0;
        HIPCHK(hipHostMalloc((void **)&e->st().h_lists, lists.size() * sizeof(int)));
        if (e->st().h_joblist) (void)hipHostFree(e->st().h_joblist);
        e->st().h_joblist = nullptr;
        HIPCHK(hipHostMalloc((void **)&e->st().h_joblist, lists.size() * sizeof(FrameJob)));
        e->st().h_lists_cap = lists.size();
    }
    const bool on_device = e->dev_jobs_p == e->d_jobs.p && e->dev_lists_p == e->d_lists.p && e->dev_joblist_p == e->d_joblist.p &&
                           e->dev_jobs.size() == nf && e->dev_lists.size() == lists.size() &&
                           !memcmp(e->dev_jobs.data(), jobs.data(), nf * sizeof(FrameJob)) &&
                           !memcmp(e->dev_lists.data(), lists.data(), lists.size() * sizeof(int));
    if (!on_device) {
        memcpy(e->st().h_jobs, jobs.data(), nf * sizeof(FrameJob));
        memcpy(e->st().h_lists, lists.data(), lists.size() * sizeof(int));
        HIPCHK(hipMemcpyAsync(e->d_jobs.p, e->st().h_jobs, nf * sizeof(FrameJob), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(e->d_lists.p, e->st().h_lists, lists.size() * sizeof(int), hipMemcpyHostToDevice, s));
        for (size_t i = 0; i < lists.size(); ++i) {
            e->st().h_joblist[i] = jobs[(size_t)lists[i]];
            e->st().h_joblist[i].fidx = (uint32_t)lists[i];
        }
        HIPCHK(hipMemcpyAsync(e->d_joblist.p, e->st().h_joblist, lists.size() * sizeof(FrameJob), hipMemcpyHostToDevice, s));
        e->dev_jobs = jobs;
        e->dev_lists = lists;
        e->dev_jobs_p = e->d_jobs.p; e->dev_lists_p = e->d_lists.p; e->dev_joblist_p = e->d_joblist.p;
    }
    e->plan_nf = nf;
    e->dbg_frames = nf;
    e->dbg_rec_slot = rec_slot;
}

void run_step(m2v_enc *e, hipStream_t s, size_t j)
{
    const m2v_enc::Step &st = e->plan_steps[j];
    launch_mb<false>(e, s, e->d_lists.p + st.off_i, st.n_i, e->g);
    launch_mb<true>(e, s, e->d_lists.p + st.off_p, st.n_p, e->g);
}

// macroblock rows [r0, r1) of GOP step j only (strip mode: edge rows first, so that their halo is on its way to the
// neighbours while the interior rows are encoded)
void run_step_rows(m2v_enc *e, hipStream_t s, size_t j, int r0, int r1)
{
    if (r0 >= r1) return;
    const m2v_enc::Step &st = e->plan_steps[j];
    Geom gg = e->g;
    gg.row0 = r0;
    gg.row1 = r1;
    geom_finish(gg);
    launch_mb<false>(e, s, e->d_lists.p + st.off_i, st.n_i, gg);
    launch_mb<true>(e, s, e->d_lists.p + st.off_p, st.n_p, gg);
}

// strip mode: the strip's first and last macroblock row of GOP step j in one launch each for the I and the P frames of the step,
// their halo rows written by the kernel itself (k_mb<.., EDGE>); up / down = the send buffers, null without a neighbour
void run_step_edges_fused(m2v_enc *e, hipStream_t s, size_t j, uint8_t *up, uint8_t *down, const uint8_t *nb_up, const uint8_t *nb_down)
{
    const m2v_enc::Step &st = e->plan_steps[j];
    Geom gg = e->g;
    const int r0 = e->g.row0, r1 = e->g.row1, nrows = r1 - r0 >= 2 ? 2 : 1;
    gg.row0 = r0;
    gg.row1 = r0 + nrows;
    gg.rstride = nrows == 2 ? r1 - 1 - r0 : 1;
    gg.edge_top = r0;
    gg.edge_bot = r1 - 1;
    geom_finish(gg);
    launch_mb_edges<false>(e, s, e->d_lists.p + st.off_i, st.n_i, gg, up, down, nullptr, nullptr);      // an I frame has no reference
    launch_mb_edges<true>(e, s, e->d_lists.p + st.off_p, st.n_p, gg, up, down, nb_up, nb_down);
}

void finish_chunk(m2v_enc *e, hipStream_t s, bool first, bool last, uint8_t *d_stream, bool advance = false)
{
    const Geom &g = e->g;
    const size_t nf = e->plan_nf;
    const size_t rows = (size_t)(g.row1 - g.row0);
    {
        Timer t(e, s, 4, (double)nf * g.ysz);
        if (!e->slice_scan_done)
            hipLaunchKernelGGL(k_slice_scan, dim3((unsigned)(nf * rows)), dim3(128), 0, s, e->d_jobs.p, g, e->d_mbinfo.p,
                               e->d_mbaux.p, e->d_mblen.p, e->d_mboff.p, e->d_slice_bytes.p, e->d_mbdep.p, 0);
        e->slice_scan_done = false;
        // offsets of every frame and slice, stream length, and the boundary dwords k_assemble ORs into cleared
        hipLaunchKernelGGL(k_frame_scan, dim3(1), dim3(1024), 0, s, e->d_jobs.p, g, (int)nf, first ? 1 : 0, last ? 1 : 0,
                           e->d_slice_bytes.p, e->d_slice_off.p, e->d_frame_off.p, e->d_ctl.p, advance ? 1 : 0,
                           (uint32_t *)d_stream);
        HIPCHK(hipGetLastError());
        t.stop();
    }
    {
        // slices, and with them the headers and the sequence end code
        Timer t(e, s, 3, (double)nf * g.ysz);
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)(nf * rows)), dim3(kAsmThreads), 0, s, e->d_jobs.p, g, (int)nf,
                           e->d_mbaux.p, e->d_mbdep.p, e->d_slots_small.p, e->d_slots.p, e->d_mblen.p, e->d_mboff.p, e->d_slice_off.p,
                           (uint32_t *)d_stream, e->d_ctl.p, first ? 1 : 0, last ? 1 : 0, e->d_frame_off.p, e->d_slice_bytes.p);
        HIPCHK(hipGetLastError());
        t.stop();
    }
    e->frames_total += nf;
}

void encode_chunk(m2v_enc *e, hipStream_t s, const uint8_t *d_frames, size_t nf, bool first, bool last,
                  uint32_t last_valid_beats, uint8_t *d_stream, bool advance = false)
{
    plan_chunk(e, s, d_frames, nf, last, last_valid_beats);
    if (e->plan_groups > 1 && !e->profile && e->plan_steps.size() > 1) {
        // The GOP segments of the chunk as `plan_groups` independent groups, one stream each: a launch of 86 400
        // wavefronts ends with a partially filled GPU (10.55 rounds of 8 192 wave slots) and the next step of the
        // same GOPs cannot start before it has drained; the other groups' launches fill those slots.  A segment stays
        // on its stream (its frames depend on each other), so plain stream order is all the synchronisation needed.
        // Off while option "profile" times the launches with in-band events (one stream: unambiguous durations).
        const int G = e->plan_groups;
        if (!e->ev_fork) HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventRecord(e->ev_fork, s));
        for (int k = 1; k < G; ++k) {
            if (!e->side[k - 1]) HIPCHK(hipStreamCreateWithFlags(&e->side[k - 1], hipStreamNonBlocking));
            if (!e->ev_join[k - 1]) HIPCHK(hipEventCreateWithFlags(&e->ev_join[k - 1], hipEventDisableTiming));
            HIPCHK(hipStreamWaitEvent(e->side[k - 1], e->ev_fork, 0));
        }
        for (size_t j = 0; j < e->plan_steps.size(); ++j) {
            const m2v_enc::Step &st = e->plan_steps[j];
            for (int k = 0; k < G; ++k) {
                hipStream_t sk = k == 0 ? s : e->side[k - 1];
                launch_mb<false>(e, sk, e->d_lists.p + st.off_i + st.cut_i[k], st.cut_i[k + 1] - st.cut_i[k], e->g);
                launch_mb<true>(e, sk, e->d_lists.p + st.off_p + st.cut_p[k], st.cut_p[k + 1] - st.cut_p[k], e->g);
            }
        }
        // every group scans its own slices right behind its last macroblock kernel (nothing in k_slice_scan looks beyond a
        // slice): the group that finishes first does it while the others still encode; only k_frame_scan and k_assemble need all
        const size_t rows = (size_t)(e->g.row1 - e->g.row0);
        for (int k = 0; k < G; ++k) {
            hipStream_t sk = k == 0 ? s : e->side[k - 1];
            const int fa = e->plan_gf[k], fb = e->plan_gf[k + 1];
            if (fb > fa)
                hipLaunchKernelGGL(k_slice_scan, dim3((unsigned)((size_t)(fb - fa) * rows)), dim3(128), 0, sk, e->d_jobs.p, e->g, e->d_mbinfo.p,
                                   e->d_mbaux.p, e->d_mblen.p, e->d_mboff.p, e->d_slice_bytes.p, e->d_mbdep.p, fa);
        }
        HIPCHK(hipGetLastError());
        e->slice_scan_done = true;
        for (int k = 1; k < G; ++k) {
            HIPCHK(hipEventRecord(e->ev_join[k - 1], e->side[k - 1]));
            HIPCHK(hipStreamWaitEvent(s, e->ev_join[k - 1], 0));
        }
    } else {
        for (size_t j = 0; j < e->plan_steps.size(); ++j) run_step(e, s, j);
    }
    finish_chunk(e, s, first, last, d_stream, advance);
}

void ctl_init(m2v_enc *e, hipStream_t s, unsigned long long cap, unsigned long long prior = 0)
{
    e->d_ctl.ensure(1);
    if (!e->st().h_ctl) HIPCHK(hipHostMalloc((void **)&e->st().h_ctl, 2 * sizeof(StreamCtl)));   // [0] read-back, [1] initial values
    StreamCtl *init = e->st().h_ctl + 1;
    init->base_bytes = 0;
    init->total_bytes = 0;
    init->cap_bytes = cap & ~3ull;
    init->prior_bytes = prior;
    init->overflow = 0;
    init->pad = 0;
    // no synchronisation: the upload slot is only rewritten by the next call, after the caller's end-of-call sync
    HIPCHK(hipMemcpyAsync(e->d_ctl.p, init, sizeof(StreamCtl), hipMemcpyHostToDevice, s));
}

// ---------------------------------------------------------------------------------------------
// host-input path: the buffered frames go through the GPU chunk by chunk; a chunk's bytes reach the FIFO
// when its read-back completes.  Two host stages alternate so that the caller's next beats are copied
// into pinned memory while the previous chunk is uploaded, encoded and read back.
// ---------------------------------------------------------------------------------------------
// start of a chunk on the port path: the bytes of the sequence that precede this chunk are the previous
// chunk's prior + total (they are still in *ctl: one stream, in order); only the padding rule needs them
__global__ void k_ctl_chain(StreamCtl *ctl, unsigned long long cap, int first)
{
    const unsigned long long prior = first ? 0ull : ctl->prior_bytes + ctl->total_bytes;
    ctl->base_bytes = 0;
    ctl->total_bytes = 0;
    ctl->cap_bytes = cap & ~3ull;
    ctl->prior_bytes = prior;
    ctl->overflow = 0;
    ctl->pad = 0;
}

// memcpy split over up to `threads` threads (the calling one included) for copies of 8 MB and more
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t bytes, int threads)
{
    const size_t kMin = 8u << 20;
    size_t n = std::min<size_t>((size_t)std::max(threads, 1), bytes / kMin);
    if (n <= 1) { memcpy(dst, src, bytes); return; }
    const size_t part = ((bytes / n) + 4095) & ~(size_t)4095;
    std::vector<std::thread> pool;
    pool.reserve(n - 1);
    for (size_t k = 1; k < n; ++k) {
        const size_t off = k * part;
        if (off >= bytes) break;
        const size_t len = std::min(part, bytes - off);
        pool.emplace_back([=] { memcpy(dst + off, src + off, len); });
    }
    memcpy(dst, src, std::min(part, bytes));
    for (auto &t : pool) t.join();
}

void ensure_staging(m2v_enc *e)
{
    m2v_enc::HostStage &h = e->st();
    const size_t want = e->batch_frames * (size_t)e->g.ysz * 3;
    if (!h.h_ctl) HIPCHK(hipHostMalloc((void **)&h.h_ctl, 2 * sizeof(StreamCtl)));
    if (!h.ev_ctl) HIPCHK(hipEventCreateWithFlags(&h.ev_ctl, hipEventDisableTiming));
    if (!h.ev_out) HIPCHK(hipEventCreateWithFlags(&h.ev_out, hipEventDisableTiming));
    if (!h.ev_up) HIPCHK(hipEventCreateWithFlags(&h.ev_up, hipEventDisableTiming));
    if (h.h_in && h.h_in_cap >= want) return;
    if (h.h_in) (void)hipHostFree(h.h_in);
    h.h_in = nullptr;
    h.h_in_cap = 0;
    HIPCHK(hipHostMalloc((void **)&h.h_in, want));
    h.h_in_cap = want;
}

// wait for (block) or poll an event; false = not reached yet
bool event_reached(hipEvent_t ev, bool block)
{
    if (block) { HIPCHK(hipEventSynchronize(ev)); return true; }
    const hipError_t r = hipEventQuery(ev);
    if (r == hipErrorNotReady) return false;
    HIPCHK(r);
    return true;
}

// Move submitted chunks forward, oldest first.  block = wait for every step; until >= 0 = return as soon
// as that stage is free again.
void progress(m2v_enc *e, bool block, int until = -1)
{
    while (!e->pending.empty()) {
        const int idx = e->pending.front();
        m2v_enc::HostStage &h = e->hs[idx];
        if (h.stage == 1) {
            if (!event_reached(h.ev_ctl, block)) return;
            if (h.h_ctl->overflow) throw HipError{hipErrorOutOfMemory, "stream larger than the worst-case bound"};
            h.bytes = (size_t)h.h_ctl->total_bytes;
            if (h.bytes > h.h_out_cap) {
                if (h.h_out) (void)hipHostFree(h.h_out);
                h.h_out = nullptr;
                h.h_out_cap = 0;
                HIPCHK(hipHostMalloc((void **)&h.h_out, h.bytes + 4096));
                h.h_out_cap = h.bytes + 4096;
            }
            // the kernels that wrote d_out are complete (ev_ctl follows them): no cross-stream wait needed
            HIPCHK(hipMemcpyAsync(h.h_out, h.d_out.p, h.bytes, hipMemcpyDeviceToHost, e->copy_stream));
            HIPCHK(hipEventRecord(h.ev_out, e->copy_stream));
            h.stage = 2;
        }
        if (!event_reached(h.ev_out, block)) return;
        e->fifo.insert(e->fifo.end(), h.h_out, h.h_out + h.bytes);
        e->stream_bytes += h.bytes;
        if (h.last) e->end_pending = true;
        h.stage = 0;
        e->pending.pop_front();
        if (idx == until) break;
    }
    if (e->pending.empty()) collect_timers(e);
}

void flush_buffered(m2v_enc *e, bool last)
{
    const size_t nf = e->buffered;
    if (nf == 0 && !last) return;
    const Geom &g = e->g;
    const size_t frame_bytes = (size_t)g.ysz * 3;
    hipStream_t s = e->stream;
    if (nf == 0) {
        // stop arrived exactly on a frame boundary after an earlier flush: only the end code is owed
        progress(e, true);
        static const uint8_t endc[4] = {0x00, 0x00, 0x01, 0xB7};          // RTL:2625-2628
        e->fifo.insert(e->fifo.end(), endc, endc + 4);
        e->stream_bytes += 4;
        const unsigned long long padded = (e->stream_bytes / 32ull + 1ull) * 32ull;   // RTL:2932-2937
        e->fifo.resize(e->fifo.size() + (size_t)(padded - e->stream_bytes), 0);
        e->stream_bytes = padded;
        e->end_pending = true;
        return;
    }
    m2v_enc::HostStage &h = e->st();
    // the stage's own device buffer, filled on the upload stream: the copy of chunk k+1 crosses PCIe while the kernels
    // of chunk k run (the stage is only refilled after its previous chunk has completed, see the end of this function)
    h.d_in.ensure(std::max(nf, h.uploaded ? e->batch_frames : (size_t)0) * frame_bytes);
    if (h.uploaded < nf)
        HIPCHK(hipMemcpyAsync(h.d_in.p + h.uploaded * frame_bytes, h.h_in + h.uploaded * frame_bytes, (nf - h.uploaded) * frame_bytes,
                              hipMemcpyHostToDevice, e->up_stream));
    h.uploaded = 0;
    HIPCHK(hipEventRecord(h.ev_up, e->up_stream));
    HIPCHK(hipStreamWaitEvent(s, h.ev_up, 0));
    // worst case ~1.2 KB per macroblock; typical streams are ~100x smaller
    const size_t cap = nf * ((size_t)g.mbs * 1216 + (size_t)g.mbh * 8 + 64) + 256;
    h.d_out.ensure(cap);
    e->d_ctl.ensure(1);
    hipLaunchKernelGGL(k_ctl_chain, dim3(1), dim3(1), 0, s, e->d_ctl.p, (unsigned long long)cap, e->first_chunk ? 1 : 0);
    encode_chunk(e, s, h.d_in.p, nf, e->first_chunk, last, e->last_frame_valid_beats, h.d_out.p);
    HIPCHK(hipMemcpyAsync(h.h_ctl, e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(h.ev_ctl, s));
    h.stage = 1;
    h.last = last;
    e->pending.push_back(e->cur);
    e->buffered = 0;
    e->first_chunk = false;
    if (!e->async || e->profile || last) {
        // profile: the HIP-event timers of a chunk are read before the next one is queued
        // last:    the caller pulls next; nothing is left to overlap with
        progress(e, true);
        return;
    }
    e->cur ^= 1;
    if (e->st().stage != 0) progress(e, true, e->cur);     // the other stage must be free before it is refilled
    ensure_staging(e);
    progress(e, false);
}

void start_sequence(m2v_enc *e, uint32_t xs, uint32_t ys, uint32_t pf)
{
    if (!e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!e->up_stream) HIPCHK(hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking));
    e->g = make_geom(e, xs, ys);            // latched on the first beat (RTL:1060-1065)
    e->pframes = pf & 0xFFu;
    e->state = m2v_enc::DURING;
    e->frames_total = 0;
    e->first_chunk = true;
    e->buffered = 0;
    e->beat_pos = 0;
    e->persist_slot = -1;
    e->end_pending = false;
    e->last_frame_valid_beats = e->g.ysz / 4;
    for (auto &h : e->hs) h.uploaded = 0;
    // the FIFO total counts stream bytes of THIS sequence (padding rule): must be empty
    e->fifo.clear();
    e->fifo_rd = 0;
    e->stream_bytes = 0;
    for (auto &st : e->stats) st = KStat{};
    ensure_staging(e);
}

void do_stop(m2v_enc *e)
{
    const Geom &g = e->g;
    const uint32_t bpf = g.ysz / 4;
    if (e->beat_pos != 0) {
        // black-fill the frame in progress (RTL:1036-1056)
        uint8_t *f = e->st().h_in + e->buffered * (size_t)g.ysz * 3;
        const size_t done = e->beat_pos * 4;
        memset(f + done, 0x00, g.ysz - done);
        memset(f + g.ysz + done, 0x80, g.ysz - done);
        memset(f + 2 * (size_t)g.ysz + done, 0x80, g.ysz - done);
        e->last_frame_valid_beats = bpf;    // the fill is materialised on the host
        e->buffered++;
        e->beat_pos = 0;
    }
    flush_buffered(e, true);
    e->state = m2v_enc::ENDED;
}

int guard(m2v_enc *e, int (*fn)(m2v_enc *, void *), void *arg)
{
    try {
        if (e->device >= 0) HIPCHK(hipSetDevice(e->device));
        return fn(e, arg);
    } catch (const HipError &h) {
        e->set_err("%s: %s", h.what, hipGetErrorString(h.e));
        return h.e == hipErrorOutOfMemory ? M2V_E_NOMEM : M2V_E_HIP;
    } catch (const std::bad_alloc &) {
        e->set_err("host allocation failed");
        return M2V_E_NOMEM;
    } catch (const std::exception &ex) {        // nothing may unwind through the C boundary (e.g. std::system_error from a copy thread)
        e->set_err("%s", ex.what());
        return M2V_E_HIP;
    } catch (...) {
        e->set_err("unknown failure");
        return M2V_E_HIP;
    }
}

}  // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

const char *m2v_version(void)
{
    return kDebug ? "m2v_mi355x 0.3-debug (gfx950, wave64, one wavefront per macroblock; M2V_DEBUG: level dump, keep_recon, ablate)"
                  : "m2v_mi355x 0.3 (gfx950, wave64, one wavefront per macroblock)";
}

m2v_enc *m2v_create(int XL, int YL, int VECTOR_LEVEL, int Q_LEVEL, int device, int *err)
{
    auto fail = [&](int code, const std::string &why) -> m2v_enc * { t_create_err = why; if (err) *err = code; return nullptr; };
    t_create_err.clear();
    if (XL < 4 || XL > 7 || YL < 4 || YL > 7 || VECTOR_LEVEL < 1 || VECTOR_LEVEL > 3 || Q_LEVEL < 1 || Q_LEVEL > 4)
        return fail(M2V_E_PARAM, "m2v_create: XL, YL must be 4..7, VECTOR_LEVEL 1..3, Q_LEVEL 1..4 (RTL:11-14)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(M2V_E_NODEVICE, "m2v_create: no HIP device (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(M2V_E_NODEVICE, "m2v_create: device ordinal out of range");
    m2v_enc *e = new (std::nothrow) m2v_enc();
    if (!e) return fail(M2V_E_NOMEM, "m2v_create: host allocation failed");
    e->XL = XL; e->YL = YL; e->VL = VECTOR_LEVEL; e->Q = Q_LEVEL; e->device = device;
    try {
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        // the port path's upload and read-back streams are created with its first sequence (start_sequence): HIP spreads a process's
        // streams over a handful of hardware queues in creation order, and a handle that only ever runs the resident entry should not
        // push its own two GOP-group streams - or another handle's - onto the same queue (three handles with three streams each did
        // exactly that: the two groups of one handle serialised, 1.08 -> 1.20 ms per step)
        upload_tables(device);
        HIPCHK(hipDeviceSynchronize());
    } catch (...) {                             // nothing may unwind through the C boundary
        std::string why = "m2v_create: ";
        int code = M2V_E_HIP;
        try { throw; }
        catch (const HipError &h) { why += std::string(h.what) + ": " + hipGetErrorString(h.e); if (h.e == hipErrorOutOfMemory) code = M2V_E_NOMEM; }
        catch (const std::bad_alloc &) { why += "host allocation failed"; code = M2V_E_NOMEM; }
        catch (const std::exception &ex) { why += ex.what(); }
        catch (...) { why += "unknown failure"; }
        if (e->stream) (void)hipStreamDestroy(e->stream);
        if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
        if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
        delete e;
        return fail(code, why);
    }
    if (err) *err = M2V_OK;
    return e;
}

void m2v_destroy(m2v_enc *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    for (auto sd : e->side) if (sd) (void)hipStreamSynchronize(sd);
    e->d_coef.release(); e->d_mbaux.release(); e->d_mbdep.release(); e->d_slots.release(); e->d_slots_small.release(); e->d_mbinfo.release(); e->d_mblen.release();
    e->d_mboff.release(); e->d_slice_bytes.release(); e->d_slice_off.release(); e->d_frame_off.release();
    e->d_jobs.release(); e->d_lists.release(); e->d_joblist.release(); e->d_ctl.release(); e->d_segs.release();
    for (auto p : e->rec_pool) (void)hipFree(p);
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto &h : e->hs) {
        h.d_out.release();
        h.d_in.release();
        if (h.ev_up) (void)hipEventDestroy(h.ev_up);
        if (h.h_in) (void)hipHostFree(h.h_in);
        if (h.h_out) (void)hipHostFree(h.h_out);
        if (h.h_ctl) (void)hipHostFree(h.h_ctl);
        if (h.h_jobs) (void)hipHostFree(h.h_jobs);
        if (h.h_lists) (void)hipHostFree(h.h_lists);
        if (h.h_joblist) (void)hipHostFree(h.h_joblist);
        if (h.ev_ctl) (void)hipEventDestroy(h.ev_ctl);
        if (h.ev_out) (void)hipEventDestroy(h.ev_out);
    }
    if (e->ev_asm) { (void)hipEventSynchronize(e->ev_asm); (void)hipEventDestroy(e->ev_asm); }
    if (e->h_asm) (void)hipHostFree(e->h_asm);
    if (e->ev_strip) { (void)hipEventSynchronize(e->ev_strip); (void)hipEventDestroy(e->ev_strip); }
    if (e->h_strip) (void)hipHostFree(e->h_strip);
    if (e->comm_stream) { (void)hipStreamSynchronize(e->comm_stream); (void)hipStreamDestroy(e->comm_stream); }
    if (e->ev_edges) (void)hipEventDestroy(e->ev_edges);
    if (e->ev_halo) (void)hipEventDestroy(e->ev_halo);
    if (e->ev_interior) (void)hipEventDestroy(e->ev_interior);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    e->d_frame_pos.release(); e->d_alloff.release(); e->d_halo.release(); e->d_strip_own.release(); e->d_gather.release();
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto ev : e->ev_join) if (ev) (void)hipEventDestroy(ev);
    for (auto sd : e->side) if (sd) (void)hipStreamDestroy(sd);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
    delete e;
}

int m2v_reset(m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    for (auto sd : e->side) if (sd) (void)hipStreamSynchronize(sd);
    if (e->strip_stream && e->strip_stream != e->stream) (void)hipStreamSynchronize(e->strip_stream);
    for (auto &h : e->hs) { h.stage = 0; h.uploaded = 0; }
    e->dev_jobs.clear(); e->dev_lists.clear(); e->dev_jobs_p = nullptr;
    e->resident_inflight = false; e->resident_empty = false;
    e->pending.clear();
    e->state = m2v_enc::IDLE;
    e->buffered = 0; e->beat_pos = 0; e->frames_total = 0; e->persist_slot = -1;
    e->first_chunk = true; e->stream_bytes = 0; e->cur = 0;
    e->fifo.clear(); e->fifo_rd = 0; e->end_pending = false;
    // a strip sequence abandoned between m2v_strip_begin and m2v_strip_finish: back to the full frame
    e->strip_active = false;
    e->strip_stream = nullptr;
    e->strip_nf = 0;
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
    e->plan_steps.clear();
    e->plan_nf = 0;
    e->g.row0 = 0; e->g.row1 = e->g.mbh; e->g.strip = 0;
    geom_finish(e->g);
    e->timed.clear(); e->ev_used = 0; e->chain_ev = nullptr;
    e->err.clear();
    return M2V_OK;
}

int m2v_geometry(const m2v_enc *e, uint32_t xsize16, uint32_t ysize16, int *width, int *height)
{
    if (!e) return M2V_E_PARAM;
    const Geom g = make_geom(e, xsize16, ysize16);
    if (width) *width = g.W;
    if (height) *height = g.H;
    return M2V_OK;
}

struct PushBeatsArgs { uint32_t xs, ys, pf; const uint8_t *y, *u, *v; size_t n; int stop; };

static int push_beats_impl(m2v_enc *e, void *argp)
{
    auto *a = (PushBeatsArgs *)argp;
    if (e->strip_active) { e->set_err("m2v_push_*: a strip sequence is open (m2v_strip_finish or m2v_reset first)"); return M2V_E_STATE; }
    if (e->resident_inflight) { e->set_err("m2v_push_*: a resident sequence is in flight (m2v_encode_resident_end first)"); return M2V_E_STATE; }
    if (e->state == m2v_enc::ENDED) return M2V_OK;              // dropped while the sequence ends (RTL:1045-1058)
    size_t i = 0;
    if (a->n == 0) {
        if (a->stop && e->state == m2v_enc::DURING) do_stop(e);
        return M2V_OK;
    }
    if (e->state == m2v_enc::IDLE) start_sequence(e, a->xs, a->ys, a->pf);
    const Geom &g = e->g;
    const size_t bpf = g.ysz / 4;
    while (i < a->n) {
        uint8_t *f = e->st().h_in + e->buffered * (size_t)g.ysz * 3;
        const size_t take = std::min(a->n - i, bpf - e->beat_pos);
        memcpy(f + e->beat_pos * 4, a->y + i * 4, take * 4);    // raster order: beat b = pixels 4b..4b+3
        memcpy(f + g.ysz + e->beat_pos * 4, a->u + i * 4, take * 4);
        memcpy(f + 2 * (size_t)g.ysz + e->beat_pos * 4, a->v + i * 4, take * 4);
        e->beat_pos += take;
        i += take;
        if (e->beat_pos == bpf) {
            e->beat_pos = 0;
            e->buffered++;
            if (e->buffered == e->batch_frames && !(a->stop && i == a->n)) flush_buffered(e, false);
        }
    }
    if (a->stop) do_stop(e);
    else progress(e, false);
    return M2V_OK;
}

int m2v_push_beats(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *y4,
                   const uint8_t *u4, const uint8_t *v4, size_t nbeats, int stop_with_last)
{
    if (!e || (nbeats && (!y4 || !u4 || !v4))) return M2V_E_PARAM;
    PushBeatsArgs a{xsize16, ysize16, pframes_count, y4, u4, v4, nbeats, stop_with_last};
    return guard(e, push_beats_impl, &a);
}

// Packed 4:4:4 sources (capture cards, SDI/HDMI receivers hand out interleaved samples): the same beats, the
// twelve port bytes of a beat simply arrive interleaved instead of on three arrays.
int m2v_push_packed(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *pixels,
                    size_t nbeats, int layout, int stop_with_last)
{
    if (!e || (nbeats && !pixels)) return M2V_E_PARAM;
    int stride, oy, ou, ov;
    switch (layout) {
    case M2V_PACKED_YUV24: stride = 3; oy = 0; ou = 1; ov = 2; break;
    case M2V_PACKED_UYV24: stride = 3; oy = 1; ou = 0; ov = 2; break;
    case M2V_PACKED_YUVX32: stride = 4; oy = 0; ou = 1; ov = 2; break;
    case M2V_PACKED_AYUV32: stride = 4; oy = 1; ou = 2; ov = 3; break;
    default: e->set_err("m2v_push_packed: unknown layout %d", layout); return M2V_E_PARAM;
    }
    constexpr size_t kBlock = 4096;                     // beats per de-interleave block (48 KB of planar data: stays in L1/L2)
    uint8_t y[kBlock * 4], u[kBlock * 4], v[kBlock * 4];
    if (nbeats == 0) {
        PushBeatsArgs a{xsize16, ysize16, pframes_count, y, u, v, 0, stop_with_last};
        return guard(e, push_beats_impl, &a);
    }
    for (size_t done = 0; done < nbeats;) {
        const size_t nb = std::min(kBlock, nbeats - done);
        const uint8_t *src = pixels + done * 4 * (size_t)stride;
        for (size_t i = 0; i < nb * 4; ++i) {
            y[i] = src[i * stride + oy];
            u[i] = src[i * stride + ou];
            v[i] = src[i * stride + ov];
        }
        done += nb;
        PushBeatsArgs a{xsize16, ysize16, pframes_count, y, u, v, nb, (stop_with_last && done == nbeats) ? 1 : 0};
        const int r = guard(e, push_beats_impl, &a);
        if (r < 0) return r;
    }
    return M2V_OK;
}

struct PushFramesArgs { uint32_t xs, ys, pf; const uint8_t *frames; size_t n; };

static int push_frames_impl(m2v_enc *e, void *argp)
{
    auto *a = (PushFramesArgs *)argp;
    if (e->strip_active) { e->set_err("m2v_push_*: a strip sequence is open (m2v_strip_finish or m2v_reset first)"); return M2V_E_STATE; }
    if (e->resident_inflight) { e->set_err("m2v_push_*: a resident sequence is in flight (m2v_encode_resident_end first)"); return M2V_E_STATE; }
    if (e->state == m2v_enc::ENDED || a->n == 0) return M2V_OK;
    if (e->state == m2v_enc::IDLE) start_sequence(e, a->xs, a->ys, a->pf);
    const Geom &g = e->g;
    const size_t fb = (size_t)g.ysz * 3;
    if (e->beat_pos != 0) {
        e->set_err("m2v_push_frames: a frame is partially filled by m2v_push_beats");
        return M2V_E_STATE;
    }
    // Frames that already sit in page-locked host memory (hipHostMalloc / hipHostRegister: capture buffers, pinned tensors)
    // cross PCIe straight from there; anything else is first copied into the stage's pinned buffer by a few threads (one core
    // moves ~25 GB/s, less than half of what the link takes).
    // The whole range must be page-locked, not just its first byte (a pointer near the end of a registered region): the
    // first and the last byte are queried, and a range that is not pinned at both ends takes the staging copy.
    auto page_locked = [](const void *p) {
        hipPointerAttribute_t attr;
        const bool yes = hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!yes) (void)hipGetLastError();          // an ordinary pointer is "invalid value" to the query: not an error here
        return yes;
    };
    const bool pinned = e->direct_upload && page_locked(a->frames) && page_locked(a->frames + a->n * fb - 1);
    bool direct_pending = false;
    for (size_t k = 0; k < a->n;) {
        m2v_enc::HostStage &h = e->st();
        const size_t take = std::min(a->n - k, e->batch_frames - e->buffered);
        if (pinned) {
            h.d_in.ensure(e->batch_frames * fb);
            if (h.uploaded < e->buffered)           // frames staged on the host earlier in this chunk go first
                HIPCHK(hipMemcpyAsync(h.d_in.p + h.uploaded * fb, h.h_in + h.uploaded * fb, (e->buffered - h.uploaded) * fb,
                                      hipMemcpyHostToDevice, e->up_stream));
            HIPCHK(hipMemcpyAsync(h.d_in.p + e->buffered * fb, a->frames + k * fb, take * fb, hipMemcpyHostToDevice, e->up_stream));
            h.uploaded = e->buffered + take;
            direct_pending = true;
        } else {
            parallel_copy(h.h_in + e->buffered * fb, a->frames + k * fb, take * fb, e->copy_threads);
        }
        e->buffered += take;
        k += take;
        if (e->buffered == e->batch_frames) flush_buffered(e, false);
    }
    if (direct_pending) HIPCHK(hipStreamSynchronize(e->up_stream));     // the caller may reuse its buffer when this returns
    progress(e, false);
    return M2V_OK;
}

int m2v_push_frames(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *frames444,
                    size_t nframes)
{
    if (!e || (nframes && !frames444)) return M2V_E_PARAM;
    PushFramesArgs a{xsize16, ysize16, pframes_count, frames444, nframes};
    return guard(e, push_frames_impl, &a);
}

static int stop_impl(m2v_enc *e, void *)
{
    if (e->state == m2v_enc::DURING) do_stop(e);       // no effect while idle / already ending (RTL:1090)
    return M2V_OK;
}

int m2v_sequence_stop(m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    return guard(e, stop_impl, nullptr);
}

int m2v_busy(const m2v_enc *e) { return e && e->state != m2v_enc::IDLE; }

static int pull_progress_impl(m2v_enc *e, void *)
{
    // chunks still in flight: take what is complete; once the sequence has been stopped wait for the rest
    progress(e, e->state == m2v_enc::ENDED);
    return M2V_OK;
}

long long m2v_pull(m2v_enc *e, uint8_t *dst, size_t cap, int *last)
{
    if (!e || (!dst && cap)) return M2V_E_PARAM;
    if (last) *last = 0;
    if (!e->pending.empty()) {
        const int r = guard(e, pull_progress_impl, nullptr);
        if (r < 0) return r;
    }
    const size_t avail = e->fifo.size() - e->fifo_rd;
    // only whole 32-byte words leave; the residue waits for more data or for the end of the sequence
    size_t n = std::min(avail, cap) & ~(size_t)31;
    if (n) memcpy(dst, e->fifo.data() + e->fifo_rd, n);
    e->fifo_rd += n;
    if (e->fifo_rd > (1u << 20) && e->fifo_rd * 2 > e->fifo.size()) {      // compact
        e->fifo.erase(e->fifo.begin(), e->fifo.begin() + (long)e->fifo_rd);
        e->fifo_rd = 0;
    }
    if (e->end_pending && e->fifo_rd == e->fifo.size()) {
        if (last) *last = 1;
        e->end_pending = false;
        e->state = m2v_enc::IDLE;                      // o_last => SEQ_IDLE (RTL:1045-1047)
        // keep fifo bookkeeping until the next sequence starts
    }
    return (long long)n;
}

struct ResidentArgs { uint32_t xs, ys, pf; const uint8_t *d_in; size_t n; uint8_t *d_out; size_t cap; size_t *bytes; hipStream_t s; bool async = false; };

// The resident entry in two halves: everything enqueued (m2v_encode_resident_begin), then the one wait and the byte count
// (m2v_encode_resident_end).  m2v_encode_resident is both, back to back.
static int resident_end_impl(m2v_enc *e, void *argp)
{
    auto *bytes = (size_t *)argp;
    if (!e->resident_inflight) { e->set_err("m2v_encode_resident_end: nothing in flight"); return M2V_E_STATE; }
    e->resident_inflight = false;
    HIPCHK(hipStreamSynchronize(e->resident_stream));
    collect_timers(e);
    if (e->st().h_ctl->overflow) { e->set_err("output buffer too small"); return M2V_E_OVERFLOW; }
    if (bytes) *bytes = (size_t)e->st().h_ctl->total_bytes;
    return M2V_OK;
}

static int resident_impl(m2v_enc *e, void *argp)
{
    auto *a = (ResidentArgs *)argp;
    if (e->state != m2v_enc::IDLE || e->strip_active || e->resident_inflight) { e->set_err("m2v_encode_resident: encoder busy"); return M2V_E_STATE; }
    if (a->n == 0) { if (a->bytes) *a->bytes = 0; e->resident_empty = true; return M2V_OK; }   // no beat: the sequence never starts
    e->resident_empty = false;
    hipStream_t s = a->s ? a->s : e->stream;
    e->g = make_geom(e, a->xs, a->ys);
    e->pframes = a->pf & 0xFFu;
    e->frames_total = 0;
    e->persist_slot = -1;
    for (auto &st : e->stats) st = KStat{};
    const Geom &g = e->g;
    const size_t fb = (size_t)g.ysz * 3;
    // the control word starts from a one-thread kernel, not from a host-to-device copy (a copy engine round trip in front of the first kernel)
    e->d_ctl.ensure(1);
    if (!e->st().h_ctl) HIPCHK(hipHostMalloc((void **)&e->st().h_ctl, 2 * sizeof(StreamCtl)));
    hipLaunchKernelGGL(k_ctl_chain, dim3(1), dim3(1), 0, s, e->d_ctl.p, (unsigned long long)a->cap, 1);
    const size_t chunk = std::max<size_t>(1, e->batch_frames);
    // align chunks to GOP boundaries so every chunk starts with an I frame where possible
    const size_t gop = e->pframes + 1u;
    size_t step = chunk >= gop ? chunk / gop * gop : chunk;
    for (size_t k = 0; k < a->n; k += step) {
        const size_t nf = std::min(step, a->n - k);
        const bool first = k == 0, last = k + nf == a->n;
        encode_chunk(e, s, a->d_in + k * fb, nf, first, last, g.ysz / 4, a->d_out, /*advance=*/k > 0);
        if (!last) HIPCHK(hipStreamSynchronize(s));    // the per-chunk work buffers are reused
    }
    HIPCHK(hipMemcpyAsync(e->st().h_ctl, e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    e->resident_inflight = true;
    e->resident_stream = s;
    if (a->async) return M2V_OK;
    return resident_end_impl(e, a->bytes);
}

int m2v_encode_resident(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const void *d_frames444,
                        size_t nframes, void *d_out, size_t cap, size_t *out_bytes, void *hip_stream)
{
    if (!e || (nframes && (!d_frames444 || !d_out))) return M2V_E_PARAM;
    ResidentArgs a{xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, (uint8_t *)d_out, cap, out_bytes,
                   (hipStream_t)hip_stream};
    return guard(e, resident_impl, &a);
}

int m2v_encode_resident_begin(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const void *d_frames444,
                              size_t nframes, void *d_out, size_t cap, void *hip_stream)
{
    if (!e || (nframes && (!d_frames444 || !d_out))) return M2V_E_PARAM;
    ResidentArgs a{xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, (uint8_t *)d_out, cap, nullptr,
                   (hipStream_t)hip_stream, true};
    return guard(e, resident_impl, &a);
}

int m2v_encode_resident_end(m2v_enc *e, size_t *out_bytes)
{
    if (!e) return M2V_E_PARAM;
    if (e->resident_empty && !e->resident_inflight) { e->resident_empty = false; if (out_bytes) *out_bytes = 0; return M2V_OK; }
    return guard(e, resident_end_impl, out_bytes);
}

// ---------------------------------------------------------------------------------------------
// strip mode (BASELINE config c5): this handle encodes macroblock rows [row0,row1) of every frame
// ---------------------------------------------------------------------------------------------
struct StripBeginArgs { uint32_t xs, ys, pf; const uint8_t *d_in; size_t n; int row0, row1; hipStream_t s; };

static int strip_begin_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripBeginArgs *)argp;
    if (e->state != m2v_enc::IDLE || e->strip_active || e->resident_inflight) { e->set_err("m2v_strip_begin: encoder busy"); return M2V_E_STATE; }
    Geom g = make_geom(e, a->xs, a->ys);
    if (a->n == 0 || a->row0 < 0 || a->row1 > g.mbh || a->row0 >= a->row1) { e->set_err("m2v_strip_begin: bad rows / no frames"); return M2V_E_PARAM; }
    g.row0 = a->row0; g.row1 = a->row1; g.strip = 1;
    geom_finish(g);
    e->g = g;
    e->pframes = a->pf & 0xFFu;
    e->frames_total = 0;
    e->persist_slot = -1;
    for (auto &st : e->stats) st = KStat{};
    e->strip_stream = a->s ? a->s : e->stream;
    plan_chunk(e, e->strip_stream, a->d_in, a->n, true, g.ysz / 4);
    e->strip_active = true;
    return M2V_OK;
}

int m2v_strip_begin(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const void *d_frames444,
                    size_t nframes, int row0, int row1, void *hip_stream)
{
    if (!e || !d_frames444) return M2V_E_PARAM;
    StripBeginArgs a{xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, row0, row1, (hipStream_t)hip_stream};
    return guard(e, strip_begin_impl, &a);
}

int m2v_strip_info(const m2v_enc *e, int *steps, size_t *halo_bytes_per_direction)
{
    if (!e || !e->strip_active) return M2V_E_STATE;
    int mh = 0;
    for (auto &st : e->plan_steps) mh = std::max(mh, st.n_h);
    if (steps) *steps = (int)e->plan_steps.size();
    if (halo_bytes_per_direction) *halo_bytes_per_direction = (size_t)mh * (size_t)(3 * e->VL) * (size_t)e->g.W;   // (YR + UR) * W per frame
    return M2V_OK;
}

struct StripStepArgs { int j; uint8_t *up, *down; const uint8_t *from_up, *from_down; int part = 0; };   // part: 0 whole strip, 1 edge rows + halo pack, 2 interior rows

static int strip_step_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripStepArgs *)argp;
    if (!e->strip_active || a->j < 0 || a->j >= (int)e->plan_steps.size()) return M2V_E_STATE;
    const m2v_enc::Step &st = e->plan_steps[a->j];
    const int r0 = e->g.row0, r1 = e->g.row1;
    if (a->part == 0) {
        run_step(e, e->strip_stream, (size_t)a->j);
    } else if (a->part == 1) {                              // the rows the neighbours need: first and last of the strip
        run_step_rows(e, e->strip_stream, (size_t)a->j, r0, r0 + 1);
        if (r1 - r0 >= 2) run_step_rows(e, e->strip_stream, (size_t)a->j, r1 - 1, r1);
    } else {                                                // everything in between; no halo is packed here
        run_step_rows(e, e->strip_stream, (size_t)a->j, r0 + 1, r1 - 1);
        return st.n_h;
    }
    if (st.n_h > 0 && (a->up || a->down)) {
        e->chain_ev = nullptr;
        hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)st.n_h, 2), dim3(256), 0, e->strip_stream, e->d_jobs.p,
                           e->d_lists.p + st.off_h, e->g, 2 * e->VL, e->VL, e->g.row0 > 0 ? a->up : nullptr,
                           e->g.row1 < e->g.mbh ? a->down : nullptr);
        HIPCHK(hipGetLastError());
    }
    return st.n_h;
}

int m2v_strip_step(m2v_enc *e, int step, void *d_send_up, void *d_send_down)
{
    if (!e) return M2V_E_PARAM;
    StripStepArgs a{step, (uint8_t *)d_send_up, (uint8_t *)d_send_down, nullptr, nullptr, 0};
    return guard(e, strip_step_impl, &a);
}

int m2v_strip_step_edges(m2v_enc *e, int step, void *d_send_up, void *d_send_down)
{
    if (!e) return M2V_E_PARAM;
    StripStepArgs a{step, (uint8_t *)d_send_up, (uint8_t *)d_send_down, nullptr, nullptr, 1};
    return guard(e, strip_step_impl, &a);
}

int m2v_strip_step_interior(m2v_enc *e, int step)
{
    if (!e) return M2V_E_PARAM;
    StripStepArgs a{step, nullptr, nullptr, nullptr, nullptr, 2};
    return guard(e, strip_step_impl, &a);
}

static int strip_halo_in_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripStepArgs *)argp;
    if (!e->strip_active || a->j < 0 || a->j >= (int)e->plan_steps.size()) return M2V_E_STATE;
    const m2v_enc::Step &st = e->plan_steps[a->j];
    if (st.n_h > 0 && (a->from_up || a->from_down)) {
        e->chain_ev = nullptr;
        hipLaunchKernelGGL(k_halo_unpack, dim3((unsigned)st.n_h, 2), dim3(256), 0, e->strip_stream, e->d_jobs.p,
                           e->d_lists.p + st.off_h, e->g, 2 * e->VL, e->VL, e->g.row0 > 0 ? a->from_up : nullptr,
                           e->g.row1 < e->g.mbh ? a->from_down : nullptr);
        HIPCHK(hipGetLastError());
    }
    return M2V_OK;
}

int m2v_strip_halo_in(m2v_enc *e, int step, const void *d_from_up, const void *d_from_down)
{
    if (!e) return M2V_E_PARAM;
    StripStepArgs a{step, nullptr, nullptr, (const uint8_t *)d_from_up, (const uint8_t *)d_from_down};
    return guard(e, strip_halo_in_impl, &a);
}

// end of a strip sequence: back to the full frame
static void strip_close(m2v_enc *e)
{
    e->strip_active = false;
    Geom full = e->g; full.row0 = 0; full.row1 = full.mbh; full.strip = 0;
    geom_finish(full);
    e->g = full;
}

// pinned host memory of at least `bytes`, kept with the handle
static void ensure_pinned(uint8_t *&p, size_t &cap, size_t bytes)
{
    if (cap >= bytes) return;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    HIPCHK(hipHostMalloc((void **)&p, bytes * 2));
    cap = bytes * 2;
}

// scans + slice assembly of this strip into d_strip; the frame offsets stay on the device (d_frame_off) and are also
// copied to pinned host memory behind ev_strip: NOTHING is synchronised here
static void strip_finish_enqueue(m2v_enc *e, uint8_t *d_strip, size_t cap)
{
    hipStream_t s = e->strip_stream;
    const size_t nf = e->plan_nf;
    e->chain_ev = nullptr;
    e->d_ctl.ensure(1);
    hipLaunchKernelGGL(k_ctl_chain, dim3(1), dim3(1), 0, s, e->d_ctl.p, (unsigned long long)cap, 1);
    finish_chunk(e, s, false, false, d_strip);
    ensure_pinned(e->h_strip, e->h_strip_cap, (nf + 1) * sizeof(unsigned long long) + sizeof(StreamCtl));
    HIPCHK(hipMemcpyAsync(e->h_strip, e->d_frame_off.p, (nf + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(e->h_strip + (nf + 1) * sizeof(unsigned long long), e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    if (!e->ev_strip) HIPCHK(hipEventCreateWithFlags(&e->ev_strip, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->ev_strip, s));
    e->strip_nf = nf;
    strip_close(e);
}

struct StripFinishArgs { uint8_t *d_strip; size_t cap; unsigned long long *frame_off; };

static int strip_finish_async_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripFinishArgs *)argp;
    if (!e->strip_active) return M2V_E_STATE;
    e->d_ctl.ensure(1);
    strip_finish_enqueue(e, a->d_strip, a->cap);
    return M2V_OK;
}

static int strip_offsets_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripFinishArgs *)argp;
    if (!e->ev_strip || e->strip_nf == 0) { e->set_err("m2v_strip_offsets: no finished strip"); return M2V_E_STATE; }
    HIPCHK(hipEventSynchronize(e->ev_strip));           // the one wait of a strip sequence: its sizes are needed on the host
    collect_timers(e);
    memcpy(a->frame_off, e->h_strip, (e->strip_nf + 1) * sizeof(unsigned long long));
    const StreamCtl *c = (const StreamCtl *)(e->h_strip + (e->strip_nf + 1) * sizeof(unsigned long long));
    if (c->overflow) { e->set_err("strip buffer too small"); return M2V_E_OVERFLOW; }
    return M2V_OK;
}

int m2v_strip_finish_async(m2v_enc *e, void *d_strip, size_t cap)
{
    if (!e || !d_strip) return M2V_E_PARAM;
    StripFinishArgs a{(uint8_t *)d_strip, cap, nullptr};
    return guard(e, strip_finish_async_impl, &a);
}

int m2v_strip_offsets(m2v_enc *e, unsigned long long *frame_off)
{
    if (!e || !frame_off) return M2V_E_PARAM;
    StripFinishArgs a{nullptr, 0, frame_off};
    return guard(e, strip_offsets_impl, &a);
}

int m2v_strip_finish(m2v_enc *e, void *d_strip, size_t cap, unsigned long long *frame_off)
{
    const int r = m2v_strip_finish_async(e, d_strip, cap);
    return r < 0 ? r : m2v_strip_offsets(e, frame_off);
}

// headers + strips of all ranks -> the final stream.  d_all_off: [nranks][nf + 1] frame offsets in DEVICE memory; the layout
// is computed there (k_strip_layout), the byte count comes back through the control word.
static void strip_assemble_enqueue(m2v_enc *e, hipStream_t s, const Geom &g, uint32_t pf, size_t nf, int nranks, const void *const *strips,
                                   const unsigned long long *d_all_off, uint8_t *d_out, size_t cap)
{
    const uint32_t gop = (pf & 0xFFu) + 1u;
    const size_t nsegs = nf * (size_t)nranks;
    e->d_segs.ensure(nsegs * sizeof(CopySeg) + 16);
    e->d_frame_pos.ensure(nf + 1);
    e->d_ctl.ensure(1);
    StripSrc src{};
    for (int r = 0; r < nranks; ++r) src.strip[r] = (const uint8_t *)strips[r];
    e->chain_ev = nullptr;
    hipLaunchKernelGGL(k_ctl_chain, dim3(1), dim3(1), 0, s, e->d_ctl.p, (unsigned long long)cap, 1);
    Timer t(e, s, 2, (double)nf * g.ysz);
    hipLaunchKernelGGL(k_strip_layout, dim3(1), dim3(kLayoutThreads), 0, s, d_all_off, nranks, (int)nf, gop, src, (CopySeg *)e->d_segs.p,
                       e->d_frame_pos.p, e->d_ctl.p);
    // every segment cut into `split` parts so that the launch has one to two thousand blocks whatever the number of ranks
    const int split = (int)std::max<size_t>(1, std::min<size_t>(32, 2048 / std::max<size_t>(nsegs, 1)));
    const unsigned blocks = (unsigned)(nsegs * (size_t)split + (nf + kCopyThreads - 1) / kCopyThreads + 1);
    hipLaunchKernelGGL(k_strip_assemble, dim3(blocks), dim3(kCopyThreads), 0, s, (const CopySeg *)e->d_segs.p, (int)nsegs, split, g, (int)nf, gop,
                       e->d_frame_pos.p, d_out, e->d_ctl.p);
    HIPCHK(hipGetLastError());
    t.stop();
}

struct StripAsmArgs { uint32_t xs, ys, pf; size_t n; int nranks; const void *const *strips; const unsigned long long *const *offs;
                      uint8_t *d_out; size_t cap; size_t *bytes; hipStream_t s; };

static int strip_assemble_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripAsmArgs *)argp;
    if (e->strip_active || e->state != m2v_enc::IDLE) return M2V_E_STATE;
    if (a->nranks > kMaxStripRanks) { e->set_err("m2v_strip_assemble: at most %d strips", kMaxStripRanks); return M2V_E_PARAM; }
    hipStream_t s = a->s ? a->s : e->stream;
    const Geom g = make_geom(e, a->xs, a->ys);
    const uint32_t gop = (a->pf & 0xFFu) + 1u;
    const size_t nf = a->n;
    // the byte count is known on the host (the caller holds the offsets): same arithmetic as k_strip_layout
    unsigned long long pos = kSeqHeaderBytes;
    for (size_t f = 0; f < nf; ++f) {
        pos += (f % gop) == 0 ? kGopHeaderBytes + 17u : 18u;
        for (int r = 0; r < a->nranks; ++r) pos += a->offs[r][f + 1] - a->offs[r][f];
    }
    const unsigned long long total = ((pos + 4) / 32ull + 1ull) * 32ull;       // end code + final word rule (RTL:2932-2937)
    if (total > a->cap) { e->set_err("output buffer too small"); return M2V_E_OVERFLOW; }
    // The offsets go up from ONE pinned staging block with an asynchronous copy on the caller's stream: the call neither blocks
    // on a pageable copy nor synchronises the stream.  The staging is rewritten by the next call only after this call's
    // copy has been consumed (ev_asm).
    const size_t b_off = (size_t)a->nranks * (nf + 1) * sizeof(unsigned long long);
    if (!e->ev_asm) HIPCHK(hipEventCreateWithFlags(&e->ev_asm, hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(e->ev_asm));
    ensure_pinned(e->h_asm, e->h_asm_cap, b_off);
    for (int r = 0; r < a->nranks; ++r) memcpy(e->h_asm + (size_t)r * (nf + 1) * sizeof(unsigned long long), a->offs[r], (nf + 1) * sizeof(unsigned long long));
    e->d_alloff.ensure((size_t)a->nranks * (nf + 1));
    HIPCHK(hipMemcpyAsync(e->d_alloff.p, e->h_asm, b_off, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(e->ev_asm, s));
    strip_assemble_enqueue(e, s, g, a->pf, nf, a->nranks, a->strips, e->d_alloff.p, a->d_out, a->cap);
    if (a->bytes) *a->bytes = (size_t)total;            // known on the host: the stream is NOT synchronised here
    return M2V_OK;
}

int m2v_strip_assemble(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, size_t nframes, int nranks,
                       const void *const *d_strips, const unsigned long long *const *frame_off, void *d_out, size_t cap,
                       size_t *out_bytes, void *hip_stream)
{
    if (!e || !d_strips || !frame_off || !d_out || nranks < 1 || nframes == 0) return M2V_E_PARAM;
    StripAsmArgs a{xsize16, ysize16, pframes_count, nframes, nranks, d_strips, frame_off, (uint8_t *)d_out, cap, out_bytes,
                   (hipStream_t)hip_stream};
    return guard(e, strip_assemble_impl, &a);
}

// ---------------------------------------------------------------------------------------------
// m2v_strip_encode: one call = one strip of one sequence, start to finish, with the exchange inside (no interpreter between
// the GOP steps).  Per step:   edge rows + halo pack  (main stream)  -> event
//                              send / recv with the two neighbours (comm stream, behind the event)     -> event
//                              interior rows          (main stream, runs while the halo crosses xGMI)
//                              neighbour rows into the reconstruction buffers (main stream, behind the comm event)
// then the strip's slices, one all-gather of the per-frame sizes, the strips to the output rank, the final assembly there.
// ---------------------------------------------------------------------------------------------
struct StripEncodeArgs { m2v_comm *comm; int rank, world, dst; uint32_t xs, ys, pf; const uint8_t *d_in; size_t n; uint8_t *d_out; size_t cap;
                         size_t *bytes; hipStream_t s; };

static int strip_encode_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripEncodeArgs *)argp;
    using clk = std::chrono::steady_clock;
    const int rank = a->rank, world = a->world;
    const Geom full = make_geom(e, a->xs, a->ys);
    if (world < 1 || world > kMaxStripRanks || world > full.mbh || rank < 0 || rank >= world || a->dst < 0 || a->dst >= world ||
        (world > 1 && (!a->comm || a->comm->world != world))) {
        e->set_err("m2v_strip_encode: bad rank / world / communicator");
        return M2V_E_PARAM;
    }
    if (rank == a->dst && !a->d_out) { e->set_err("m2v_strip_encode: the output rank needs d_out"); return M2V_E_PARAM; }     // before anything collective
    // contiguous strips, sizes differing by at most one row, the first mbh % world ranks get the extra row (parallel.partition_rows)
    const int base = full.mbh / world, rem = full.mbh % world;
    const int row0 = rank * base + std::min(rank, rem), row1 = row0 + base + (rank < rem ? 1 : 0);
    StripBeginArgs b{a->xs, a->ys, a->pf, a->d_in, a->n, row0, row1, a->s};
    int r = strip_begin_impl(e, &b);
    if (r < 0) return r;
    hipStream_t s = e->strip_stream;
    const Geom &g = e->g;
    const size_t nf = e->plan_nf;
    int mh = 0;
    for (auto &st : e->plan_steps) mh = std::max(mh, st.n_h);
    const size_t halo_cap = (size_t)mh * (size_t)(3 * e->VL) * (size_t)g.W;
    e->d_halo.ensure(4 * halo_cap + 64);
    uint8_t *send_up = e->d_halo.p, *send_down = send_up + halo_cap, *recv_up = send_down + halo_cap, *recv_down = recv_up + halo_cap;
    const bool fused = !e->conformant && e->dct_mfma && !e->keep_recon;
    if (world > 1) {
        // (the general form's exchange stream only when that form runs: every stream a process creates moves the others around the
        // handful of hardware queues)
        if (!fused && !e->comm_stream) HIPCHK(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
        if (!e->ev_edges) HIPCHK(hipEventCreateWithFlags(&e->ev_edges, hipEventDisableTiming));
        if (!e->ev_halo) HIPCHK(hipEventCreateWithFlags(&e->ev_halo, hipEventDisableTiming));
    }
    // profile: GPU events around the exchange of every step: halo_total = edge rows (and their halo) written .. neighbour rows and
    // interior rows both there; halo_exposed = how much of that came after the interior rows were done
    std::vector<hipEvent_t> marks;
    auto mark = [&](hipStream_t on) {
        if (!e->profile) return;
        hipEvent_t ev = pool_event(e);
        e->chain_ev = nullptr;
        HIPCHK(hipEventRecord(ev, on));
        marks.push_back(ev);
    };
    const bool up = row0 > 0, down = row1 < g.mbh;
    // The edge rows run as ONE launch of the instantiation that also fills the halo buffers (no pack kernel), the interior rows at
    // the same time on a second stream: a strip of an 8-GPU job is ~20 000 wavefronts per step, 2.5 rounds of the wave slots -
    // edge rows first and alone would hold the whole GPU for one macroblock lifetime at a third of its slots.
    hipStream_t side = nullptr;
    if (world > 1) {
        if (!e->side[0]) HIPCHK(hipStreamCreateWithFlags(&e->side[0], hipStreamNonBlocking));
        side = e->side[0];
        if (!e->ev_done) HIPCHK(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
        if (!e->ev_interior) HIPCHK(hipEventCreateWithFlags(&e->ev_interior, hipEventDisableTiming));
        HIPCHK(hipEventRecord(e->ev_done, s));                  // the plan's uploads
    }
    double us_in_comm = 0;                 // host time inside the communicator (a local communicator blocks there until the neighbour thread has posted)
    const auto t_loop = clk::now();
    for (int j = 0; j < (int)e->plan_steps.size(); ++j) {
        const int n_h = e->plan_steps[(size_t)j].n_h;
        const bool xchg = world > 1 && n_h > 0 && (up || down);
        const size_t nbytes = (size_t)n_h * (size_t)(3 * e->VL) * (size_t)g.W;
        if (world > 1 && fused) {
            // main stream:  EDGE(j) [reads the rows received in step j-1, writes the rows to send] -> send / recv(j)
            // side stream:  interior(j)
            // EDGE(j) and interior(j) both need ALL of step j-1 on this strip; the neighbours' rows only EDGE(j) - and it follows
            // the receive in stream order.  Two launches, two event records, two waits and one exchange per step.
            HIPCHK(hipStreamWaitEvent(side, j == 0 ? e->ev_done : e->ev_edges, 0));
            if (j > 0) HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));
            run_step_edges_fused(e, s, (size_t)j, xchg && up ? send_up : nullptr, xchg && down ? send_down : nullptr,
                                 up ? recv_up : nullptr, down ? recv_down : nullptr);
            mark(s);
            HIPCHK(hipEventRecord(e->ev_edges, s));
            run_step_rows(e, side, (size_t)j, row0 + 1, row1 - 1);
            mark(side);
            HIPCHK(hipEventRecord(e->ev_interior, side));
            if (xchg) {
                const auto t_c = clk::now();
                a->comm->halo(rank, up ? send_up : nullptr, up ? recv_up : nullptr, down ? send_down : nullptr, down ? recv_down : nullptr, nbytes, s);
                us_in_comm += std::chrono::duration<double, std::micro>(clk::now() - t_c).count();
            }
            mark(s);
            if (j + 1 == (int)e->plan_steps.size()) HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));      // the scans follow on the main stream
            continue;
        }
        if (xchg) {
            // the general form (option conformant / dct_mfma = 0 / the debug library's keep_recon): edge rows, pack kernel, exchange
            // on a stream of its own beside the interior rows, unpack kernel
            HIPCHK(hipStreamWaitEvent(side, e->ev_done, 0));    // the previous step, neighbour rows included
            StripStepArgs sa{j, send_up, send_down, nullptr, nullptr, 1};
            if ((r = strip_step_impl(e, &sa)) < 0) return r;
            mark(s);
            HIPCHK(hipEventRecord(e->ev_edges, s));
            run_step_rows(e, side, (size_t)j, row0 + 1, row1 - 1);
            mark(side);
            HIPCHK(hipEventRecord(e->ev_interior, side));
            HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_edges, 0));
            const auto t_c = clk::now();
            a->comm->halo(rank, up ? send_up : nullptr, up ? recv_up : nullptr, down ? send_down : nullptr, down ? recv_down : nullptr, nbytes,
                          e->comm_stream);
            us_in_comm += std::chrono::duration<double, std::micro>(clk::now() - t_c).count();
            HIPCHK(hipEventRecord(e->ev_halo, e->comm_stream));
            HIPCHK(hipStreamWaitEvent(s, e->ev_halo, 0));
            HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));
            mark(s);
            StripStepArgs sh{j, nullptr, nullptr, up ? recv_up : nullptr, down ? recv_down : nullptr};
            if ((r = strip_halo_in_impl(e, &sh)) < 0) return r;
        } else {
            StripStepArgs sa{j, nullptr, nullptr, nullptr, nullptr, 0};
            if ((r = strip_step_impl(e, &sa)) < 0) return r;
        }
        if (world > 1) HIPCHK(hipEventRecord(e->ev_done, s));
    }
    e->strip_stats.steps = (int)e->plan_steps.size();
    e->strip_stats.host_us_per_step = std::chrono::duration<double, std::micro>(clk::now() - t_loop).count() / std::max<size_t>(1, e->plan_steps.size());
    e->strip_stats.comm_us_per_step = us_in_comm / std::max<size_t>(1, e->plan_steps.size());
    hipEvent_t g0 = nullptr, g1 = nullptr;

    // ---- this strip's slices; sizes; strips to the output rank; final assembly ----
    const size_t strip_cap = nf * ((size_t)(row1 - row0) * g.mbw * 1216 + (size_t)(row1 - row0) * 8 + 64) + 256;     // worst case
    e->d_strip_own.ensure(strip_cap);
    const Geom gfull = full;
    strip_finish_enqueue(e, e->d_strip_own.p, strip_cap);       // also closes the strip sequence
    if (e->profile) { g0 = pool_event(e); e->chain_ev = nullptr; HIPCHK(hipEventRecord(g0, s)); }     // from here: sizes, gather, final assembly
    const void *strips[kMaxStripRanks] = {};
    const unsigned long long *d_all = e->d_frame_off.p;
    if (world > 1) {
        e->d_alloff.ensure((size_t)world * (nf + 1));
        a->comm->allgather_u64(rank, e->d_frame_off.p, e->d_alloff.p, nf + 1, s);
        ensure_pinned(e->h_asm, e->h_asm_cap, (size_t)world * (nf + 1) * sizeof(unsigned long long));
        HIPCHK(hipMemcpyAsync(e->h_asm, e->d_alloff.p, (size_t)world * (nf + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));                        // the sizes decide the receive counts: the one host wait
        // (an overflow of this strip's buffer - impossible with the worst-case size above - is reported at the end: the other ranks
        // are waiting in the gather, and a rank that left now would leave them there)
        size_t sizes[kMaxStripRanks] = {}, total_in = 0;
        for (int k = 0; k < world; ++k) {
            sizes[k] = (size_t)((const unsigned long long *)e->h_asm)[(size_t)k * (nf + 1) + nf];
            if (k != a->dst) total_in += (sizes[k] + 255) & ~(size_t)255;
        }
        void *bufs[kMaxStripRanks] = {};
        if (rank == a->dst) {
            e->d_gather.ensure(total_in + 256);
            size_t off = 0;
            for (int k = 0; k < world; ++k) {
                if (k == a->dst) { strips[k] = e->d_strip_own.p; continue; }
                bufs[k] = e->d_gather.p + off;
                strips[k] = bufs[k];
                off += (sizes[k] + 255) & ~(size_t)255;
            }
        }
        a->comm->gather(rank, a->dst, e->d_strip_own.p, sizes, bufs, s);
        d_all = e->d_alloff.p;
    } else {
        strips[0] = e->d_strip_own.p;
    }
    size_t out_bytes = 0;
    if (rank == a->dst) {
        strip_assemble_enqueue(e, s, gfull, a->pf, nf, world, strips, d_all, a->d_out, a->cap);
        if (e->profile) { g1 = pool_event(e); e->chain_ev = nullptr; HIPCHK(hipEventRecord(g1, s)); }
        if (!e->st().h_ctl) HIPCHK(hipHostMalloc((void **)&e->st().h_ctl, 2 * sizeof(StreamCtl)));
        HIPCHK(hipMemcpyAsync(e->st().h_ctl, e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    } else if (e->profile) { g1 = pool_event(e); e->chain_ev = nullptr; HIPCHK(hipEventRecord(g1, s)); }
    HIPCHK(hipStreamSynchronize(s));
    {
        const StreamCtl *c = (const StreamCtl *)(e->h_strip + (nf + 1) * sizeof(unsigned long long));
        if (c->overflow) { e->set_err("strip buffer too small"); return M2V_E_OVERFLOW; }
    }
    if (rank == a->dst) {
        if (e->st().h_ctl->overflow) { e->set_err("output buffer too small"); return M2V_E_OVERFLOW; }
        out_bytes = (size_t)e->st().h_ctl->total_bytes;
    }
    if (e->profile) {
        double tot = 0, exp = 0;
        for (size_t k = 0; k + 3 <= marks.size(); k += 3) {       // per step: edges done (main), interior done (side), both + halo there (main)
            float m1 = 0, m2 = 0;
            if (hipEventElapsedTime(&m1, marks[k], marks[k + 2]) == hipSuccess) tot += m1;
            if (hipEventElapsedTime(&m2, marks[k + 1], marks[k + 2]) == hipSuccess && m2 > 0) exp += m2;
        }
        float gm = 0;
        if (g0 && g1 && hipEventElapsedTime(&gm, g0, g1) == hipSuccess) e->strip_stats.gather_ms = gm;
        e->strip_stats.halo_total_ms = tot;
        e->strip_stats.halo_exposed_ms = exp;
    }
    collect_timers(e);
    if (a->bytes) *a->bytes = out_bytes;
    return M2V_OK;
}

int m2v_strip_encode(m2v_enc *e, m2v_comm *comm, int rank, int world, int dst_rank, uint32_t xsize16, uint32_t ysize16,
                     uint32_t pframes_count, const void *d_frames444, size_t nframes, void *d_out, size_t cap, size_t *out_bytes,
                     void *hip_stream)
{
    if (!e || !d_frames444 || nframes == 0) return M2V_E_PARAM;
    StripEncodeArgs a{comm, rank, world, dst_rank, xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, (uint8_t *)d_out, cap,
                      out_bytes, (hipStream_t)hip_stream};
    const int r = guard(e, strip_encode_impl, &a);
    if (r < 0 && e->strip_active) strip_close(e);            // a failed sequence does not leave the handle in strip mode
    if (r < 0 && comm) comm->abort();                         // ... and the other ranks of an in-process communicator do not wait for it for ever
    return r;
}

int m2v_strip_stats(const m2v_enc *e, double *halo_total_ms, double *halo_exposed_ms, double *gather_ms, double *host_us_per_step,
                    double *comm_us_per_step)
{
    if (!e) return M2V_E_PARAM;
    if (comm_us_per_step) *comm_us_per_step = e->strip_stats.comm_us_per_step;
    if (halo_total_ms) *halo_total_ms = e->strip_stats.halo_total_ms;
    if (halo_exposed_ms) *halo_exposed_ms = e->strip_stats.halo_exposed_ms;
    if (gather_ms) *gather_ms = e->strip_stats.gather_ms;
    if (host_us_per_step) *host_us_per_step = e->strip_stats.host_us_per_step;
    return e->strip_stats.steps;
}

// ---- communicators (m2v_comm.hpp) ----

int m2v_comm_unique_id(void *id, size_t cap)
{
    if (!id || cap < sizeof(ncclUniqueId)) return M2V_E_PARAM;
    RcclApi &api = RcclApi::get();
    if (!api.ok()) { t_comm_err = api.err; return M2V_E_NODEVICE; }
    ncclUniqueId u;
    const ncclResult_t r = api.GetUniqueId(&u);
    if (r != ncclSuccess) { t_comm_err = std::string("ncclGetUniqueId: ") + api.GetErrorString(r); return M2V_E_HIP; }
    memcpy(id, &u, sizeof u);
    return (int)sizeof u;
}

m2v_comm *m2v_comm_init_rccl(const void *id, int rank, int world, int device, int *err)
{
    auto fail = [&](int code, const std::string &why) -> m2v_comm * { t_comm_err = why; if (err) *err = code; return nullptr; };
    if (!id || world < 1 || world > kMaxStripRanks || rank < 0 || rank >= world) return fail(M2V_E_PARAM, "m2v_comm_init_rccl: bad rank / world");
    if (hipSetDevice(device) != hipSuccess) return fail(M2V_E_NODEVICE, "m2v_comm_init_rccl: device ordinal out of range");
    try {
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        m2v_comm *c = new RcclComm(u, rank, world);
        if (err) *err = M2V_OK;
        return c;
    } catch (const std::exception &ex) {
        return fail(M2V_E_HIP, ex.what());
    }
}

m2v_comm *m2v_comm_init_solo(int world, int *err)
{
    if (world < 1 || world > kMaxStripRanks) { t_comm_err = "m2v_comm_init_solo: 1..16 ranks"; if (err) *err = M2V_E_PARAM; return nullptr; }
    m2v_comm *c = new (std::nothrow) SoloComm(world);
    if (err) *err = c ? M2V_OK : M2V_E_NOMEM;
    return c;
}

m2v_comm *m2v_comm_init_local(int world, int *err)
{
    if (world < 1 || world > LocalComm::kMax) { t_comm_err = "m2v_comm_init_local: 1..16 ranks"; if (err) *err = M2V_E_PARAM; return nullptr; }
    m2v_comm *c = new (std::nothrow) LocalComm(world);
    if (err) *err = c ? M2V_OK : M2V_E_NOMEM;
    return c;
}

void m2v_comm_destroy(m2v_comm *c) { delete c; }

int m2v_comm_selftest(m2v_comm *c, int rank, const void *d_send, void *d_recv, size_t nbytes, void *hip_stream)
{
    if (!c || !d_send || !d_recv) return M2V_E_PARAM;
    try {
        c->loopback(rank, d_send, d_recv, nbytes, (hipStream_t)hip_stream);
        return M2V_OK;
    } catch (const std::exception &ex) {
        t_comm_err = ex.what();
        return M2V_E_HIP;
    }
}

const char *m2v_comm_last_error(void) { return t_comm_err.c_str(); }

int m2v_set_option(m2v_enc *e, const char *name, long long value)
{
    if (!e || !name) return M2V_E_PARAM;
    if (!strcmp(name, "batch_frames")) {
        if (value < 1 || e->state != m2v_enc::IDLE) return M2V_E_PARAM;
        // any chunk length: the byte offsets inside a chunk are scanned in 64 bits (k_frame_scan).  65536 is a sanity bound only
        // (a chunk's frames are buffered on the device: 65536 frames of the largest geometry would be 824 GB)
        if (value > 65536) { e->set_err("m2v_set_option: batch_frames is at most 65536"); return M2V_E_PARAM; }
        e->batch_frames = (size_t)value;
        return M2V_OK;
    }
    if (!strcmp(name, "profile")) { e->profile = value != 0; return M2V_OK; }
    if (!strcmp(name, "async")) { e->async = value != 0; return M2V_OK; }
    if (!strcmp(name, "split_streams")) {
        if (value < 0 || value > m2v_enc::kMaxSplit) return M2V_E_PARAM;
        e->split_streams = value < 1 ? 1 : (int)value;      // 0 and 1 both mean one stream
        return M2V_OK;
    }
    if (!strcmp(name, "conformant")) {
        if (e->state != m2v_enc::IDLE) return M2V_E_PARAM;
        e->conformant = value != 0;
        return M2V_OK;
    }
    if (!strcmp(name, "dct_mfma")) { e->dct_mfma = value != 0; return M2V_OK; }
    if (!strcmp(name, "direct_upload")) { e->direct_upload = value != 0; return M2V_OK; }
    if (!strcmp(name, "copy_threads")) { if (value < 1 || value > 64) return M2V_E_PARAM; e->copy_threads = (int)value; return M2V_OK; }
    if (kDebug) {       // libm2v_mi355x_dbg.so only (-DM2V_DEBUG): the shipped library does not know these names
        if (!strcmp(name, "keep_recon")) { e->keep_recon = value != 0; return M2V_OK; }
        if (!strcmp(name, "ablate")) { e->ablate = (int)value; return M2V_OK; }   // profiling aid: output is invalid when != 0
    }
    e->set_err("m2v_set_option: unknown option '%s'", name);
    return M2V_E_PARAM;
}

int m2v_kernel_stats(const m2v_enc *e, int kernel, double *ms, double *units)
{
    if (!e || kernel < 0 || kernel > 4) return M2V_E_PARAM;
    if (ms) *ms = e->stats[kernel].ms;
    if (units) *units = e->stats[kernel].units;
    return e->stats[kernel].launches;
}

struct DebugArgs { int what; void *dst; size_t cap; long long ret; };

static int debug_impl(m2v_enc *e, void *argp)
{
    auto *a = (DebugArgs *)argp;
    const Geom &g = e->g;
    const size_t nmb = e->dbg_frames * (size_t)g.mbs;
    const void *src = nullptr;
    size_t bytes = 0;
    switch (a->what) {
        case 0: src = e->d_mbinfo.p; bytes = nmb * 4; break;
        case 1: if (!e->keep_recon) return M2V_E_STATE; src = e->d_coef.p; bytes = nmb * 768; break;
        case 2: src = e->d_mblen.p; bytes = nmb * 4; break;
        case 3: {
            const size_t rb = e->rec_bytes;
            bytes = e->dbg_frames * rb;
            if (bytes > a->cap) return M2V_E_OVERFLOW;
            memset(a->dst, 0, bytes);
            for (size_t k = 0; k < e->dbg_frames; ++k)
                if (k < e->dbg_rec_slot.size() && e->dbg_rec_slot[k] >= 0)
                    HIPCHK(hipMemcpy((uint8_t *)a->dst + k * rb, e->rec_pool[e->dbg_rec_slot[k]], rb, hipMemcpyDeviceToHost));
            a->ret = (long long)bytes;
            return M2V_OK;
        }
        default: return M2V_E_PARAM;
    }
    if (bytes > a->cap) return M2V_E_OVERFLOW;
    HIPCHK(hipMemcpy(a->dst, src, bytes, hipMemcpyDeviceToHost));
    a->ret = (long long)bytes;
    return M2V_OK;
}

long long m2v_debug_read(m2v_enc *e, int what, void *dst, size_t cap)
{
    if (!e || !dst) return M2V_E_PARAM;
    DebugArgs a{what, dst, cap, 0};
    const int r = guard(e, debug_impl, &a);
    return r < 0 ? r : a.ret;
}

const char *m2v_last_error(const m2v_enc *e) { return e ? e->err.c_str() : t_create_err.c_str(); }

/* table accessors (no GPU needed): tests/test_abi.py checks the product's tables against the oracle's */
int m2v_debug_table(int which, int i, int j)
{
    switch (which) {
        case 0: return kDctBasis[(i & 7) * 8 + (j & 7)];
        case 1: return kIntraW[(i & 7) * 8 + (j & 7)];
        case 2: return kZigzagPos[(i & 7) * 8 + (j & 7)];
        case 3: return i >= 0 && i < 17 ? kMotionCode[i] : -1;
        case 4: return i >= 0 && i < 64 ? kCbpCode[i] : -1;
        case 5: return i >= 0 && i < 2 && j >= 0 && j < 12 ? (kDcSizeLen[i][j] << 16) | kDcSizeCode[i][j] : -1;
        case 6: return i >= 0 && i < 32 && j >= 1 && j <= 40 ? kAcCode[i * 40 + j - 1] : 0;
        default: return -1;
    }
}

}  // extern "C"
