// m2v_strips.hip — strip mode (BASELINE config c5): a handle encodes the macroblock rows [row0, row1) of every
// frame; slices are independent (RTL:2704-2715) and strips meet only in the +-2 VECTOR_LEVEL luma / +-VECTOR_LEVEL chroma rows of the
// previous reconstruction (window geometry RTL:1446-1448), which the communicators of m2v_comm.hpp move between the GPUs.
#include <chrono>
#include <functional>
#include <new>

#include "m2v_host.hpp"
#include "m2v_comm.hpp"

static thread_local std::string t_comm_err;          // why the last m2v_comm_* constructor on this thread failed

extern "C" {

// ---------------------------------------------------------------------------------------------
// strip mode (BASELINE config c5): this handle encodes macroblock rows [row0,row1) of every frame
// ---------------------------------------------------------------------------------------------
struct StripBeginArgs { uint32_t xs, ys, pf; const uint8_t *d_in; size_t n; int row0, row1; hipStream_t s; };

static int strip_begin_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripBeginArgs *)argp;
    if (e->state != m2v_enc::IDLE || e->strip_active || e->resident_inflight || e->strip_inflight) { e->set_err("m2v_strip_begin: encoder busy"); return M2V_E_STATE; }
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
    e->scan_peer_gaveup = nullptr;                          // (left behind by a sequence that failed before its scans: not this one's)
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
        timer_break(e);
        launch_halo_pack(e, e->strip_stream, e->d_lists.p + st.off_h, st.n_h, e->g.row0 > 0 ? a->up : nullptr, e->g.row1 < e->g.mbh ? a->down : nullptr);
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
        timer_break(e);
        launch_halo_unpack(e, e->strip_stream, e->d_lists.p + st.off_h, st.n_h, e->g.row0 > 0 ? a->from_up : nullptr,
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


// scans + slice assembly of this strip into d_strip; the frame offsets stay on the device (d_frame_off) and are also
// copied to pinned host memory behind ev_strip: NOTHING is synchronised here
static void strip_finish_enqueue(m2v_enc *e, uint8_t *d_strip, size_t cap)
{
    hipStream_t s = e->strip_stream;
    const size_t nf = e->plan_nf;
    timer_break(e);
    e->d_ctl.ensure(1);
    ctl_begin(e, (unsigned long long)cap, true);
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
    e->strip_stream = nullptr;                          // nothing of the sequence is left on the (possibly caller-owned) stream
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
    timer_break(e);
    Timer t(e, s, 2, (double)nf * g.ysz);
    launch_strip_assemble(e, s, g, gop, nf, nranks, src, d_all_off, d_out, (unsigned long long)cap);
    HIPCHK(hipGetLastError());
    t.stop();
}

struct StripAsmArgs { uint32_t xs, ys, pf; size_t n; int nranks; const void *const *strips; const unsigned long long *const *offs;
                      uint8_t *d_out; size_t cap; size_t *bytes; hipStream_t s; };

static int strip_assemble_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripAsmArgs *)argp;
    if (e->strip_active || e->resident_inflight || e->strip_inflight || e->state != m2v_enc::IDLE) { e->set_err("m2v_strip_assemble: encoder busy"); return M2V_E_STATE; }
    if (((uintptr_t)a->d_out & 15u) != 0) { e->set_err("m2v_strip_assemble: d_out must be 16-byte aligned"); return M2V_E_PARAM; }
    if (a->nranks > kMaxStripRanks) { e->set_err("m2v_strip_assemble: at most %d strips", kMaxStripRanks); return M2V_E_PARAM; }
    for (int r = 0; r < a->nranks; ++r)
        if (((uintptr_t)a->strips[r] & 3u) != 0) { e->set_err("m2v_strip_assemble: d_strips[%d] must be 4-byte aligned", r); return M2V_E_PARAM; }
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
// the GOP steps).  Per step (the usual form; the general form of options conformant / dct_mfma = 0 uses pack / unpack kernels
// and a stream of its own for the exchange):
//      main stream:  EDGE(j) [first and last macroblock row: reads the rows received in step j-1, writes the rows to send]
//                    -> send / recv(j) with the two neighbours
//      side stream:  interior(j), beside both
// then the strip's slices and one all-gather of the per-frame sizes.  All of that is the SEQUENCE: it depends on nothing the
// host has to look at, so it is enqueued in one go - or, from the second call of the same shape on, launched as ONE recorded
// hipGraph (option "strip_graph").  The recording does NOT shorten the sequence: hipGraphLaunch takes the host as long as the
// individual calls did on this runtime, and the step is bound by the GPU-side chain edge rows -> exchange -> next edge rows, not
// by the host (profiles/r04_experiments.txt item 1: output rank -4 %, inner rank +9 %).  It is kept for world == 1 and as an
// opt-in; the peer transport (below) is what takes the exchange out of the chain.  Then the one host wait (the sizes decide the
// receive counts), the strips to the output rank, the final assembly there.
//
// Failures: a rank whose local work fails (a launch or a copy refused) keeps the collective call order - it goes on exchanging,
// with whatever is in its buffers - and marks its row of the all-gathered sizes; every rank then sees the mark after the same
// call, skips the gather and returns an error.  Nobody is left waiting inside RCCL for a rank that has gone home.
// ---------------------------------------------------------------------------------------------
struct StripEncodeArgs { m2v_comm *comm; int rank, world, dst; uint32_t xs, ys, pf; const uint8_t *d_in; size_t n; uint8_t *d_out; size_t cap;
                         size_t *bytes; hipStream_t s; };

// frames of GOP step j whose reconstruction a later frame references (= the frames of the step's halo list): GOPs longer than j + 1
static int halo_frames_of_step(size_t nf, uint32_t gop, int j)
{
    int n = 0;
    for (size_t a = 0; a < nf; a += gop)
        if ((size_t)j + 1 < std::min<size_t>(gop, nf - a)) ++n;
    return n;
}

constexpr unsigned long long kStripPoison = ~0ull;         // a failed rank's "size" in the all-gathered table
constexpr unsigned long long kStripRetry = ~0ull - 1ull;   // peer transport: a wait on this rank ran out of budget - the sequence has to be encoded again

struct StripSeq {
    m2v_comm *comm; int rank, world, row0, row1; bool up, down, fused;
    uint8_t *send_up, *send_down, *recv_up, *recv_down;
    size_t strip_cap;
    uint32_t gop;
    size_t nf;
    int steps;
    int W;              // luma width of the sequence (the exchange sizes must not depend on whether this rank's plan succeeded)
    int mbw;
    PeerState *peer;    // non-null: the peer form of the step (one launch, rows stored into the neighbours' landing blocks)
};

// One strip sequence between its launch and its collection: what m2v_strip_encode_end (or the second half of m2v_strip_encode) needs to
// know about what m2v_strip_encode_begin enqueued.  Lives with the handle (m2v_enc::flight, allocated on first use).
struct m2v::StripFlight {
    StripEncodeArgs a{};
    StripSeq q{};
    Geom full{};
    bool defer_sizes = false;       // the sizes all-gather has not been issued yet (it is the first thing the collection does)
    bool halves = false;            // launched by m2v_strip_encode_begin (a rank that does not own the output then leaves _end without the final wait)
    bool use_peer = false;
    PeerState *pst = nullptr;
    int fail = 0;
    std::string fail_text;
    hipStream_t s = nullptr;
    int graph_used = 0;
    std::vector<hipEvent_t> marks;
};

static StripFlight &flight_of(m2v_enc *e)
{
    if (!e->flight) e->flight = new StripFlight();
    return *e->flight;
}

// Enqueues the sequence on s (and the handle's side / comm streams, forked from and joined back into s by events): no allocation,
// no synchronisation, no host state of the handle changed - this is what is recorded into the graph.  `fail` (direct mode only):
// see above; local work is skipped once it is set, the exchanges are not.
// everybody's per-frame sizes (and the marks of a failed / retrying rank): the one collective behind a strip's slices
static void strip_enqueue_sizes(m2v_enc *e, hipStream_t s, const StripSeq &q, int fail)
{
    if (q.world <= 1) return;
    if (fail) (void)hipMemsetAsync(e->d_frame_off.p, 0xFF, (q.nf + 1) * sizeof(unsigned long long), s);        // the mark
    q.comm->allgather_u64(q.rank, e->d_frame_off.p, e->d_alloff.p, q.nf + 1, s);
    const hipError_t ce = hipMemcpyAsync(e->h_asm, e->d_alloff.p, (size_t)q.world * (q.nf + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
    if (ce != hipSuccess && !fail) throw HipError{ce, "hipMemcpyAsync(all-gathered sizes)"};
}

static void strip_enqueue_sequence(m2v_enc *e, hipStream_t s, const StripSeq &q, bool recording, int &fail, std::string &fail_text,
                                   std::vector<hipEvent_t> *marks, double *us_in_comm, bool with_sizes = true)
{
    using clk = std::chrono::steady_clock;
    auto local = [&](auto &&fn) {
        if (fail) return;
        if (recording) { fn(); return; }                    // a failure while recording abandons the recording, nothing has run yet
        try { fn(); }
        catch (const HipError &h) {
            fail = h.e == hipErrorOutOfMemory ? M2V_E_NOMEM : M2V_E_HIP;
            fail_text = std::string(h.what) + ": " + hipGetErrorString(h.e);
            (void)hipGetLastError();
        }
    };
    auto mark = [&](hipStream_t on) {
        if (!marks || fail) return;
        hipEvent_t ev = pool_event(e);
        timer_break(e);
        HIPCHK(hipEventRecord(ev, on));
        marks->push_back(ev);
    };
    auto exchange = [&](size_t nbytes, hipStream_t on) {
        const auto t_c = clk::now();
        q.comm->halo(q.rank, q.up ? q.send_up : nullptr, q.up ? q.recv_up : nullptr, q.down ? q.send_down : nullptr, q.down ? q.recv_down : nullptr,
                     nbytes, on);
        if (us_in_comm) *us_in_comm += std::chrono::duration<double, std::micro>(clk::now() - t_c).count();
    };
    hipStream_t side = q.world > 1 ? e->side[0] : nullptr;
    if (q.world > 1 && !q.peer) local([&] { HIPCHK(hipEventRecord(e->ev_done, s)); });          // the plan's uploads
    for (int j = 0; j < q.steps; ++j) {
        const int n_h = halo_frames_of_step(q.nf, q.gop, j);
        const bool xchg = q.world > 1 && n_h > 0 && (q.up || q.down);
        const size_t nbytes = (size_t)n_h * (size_t)(3 * e->VL) * (size_t)q.W;
        if (q.peer) {
            // ONE launch per frame type and GOP group for the whole strip; the rows of step j land in the neighbours' buffers of parity j & 1, the
            // neighbours' rows of step j - 1 are read from this rank's buffers of the other parity once the GOP's counter says that
            // every block of the neighbour's row has delivered.  Why two parities are enough: a block stores into a neighbour's
            // buffer (at its GOP's place) only after it has seen that neighbour's count for the GOP's frame of step j - 1 complete,
            // i.e. after every read the neighbour made of that place in step j - 1.
            PeerState &ps = *q.peer;
            const unsigned set = (unsigned)(ps.seq & 1), par = (unsigned)(j & 1);
            PeerStep k{};
            k.cnt_up = q.up ? ps.cnt(0, set) : nullptr;
            k.cnt_down = q.down ? ps.cnt(1, set) : nullptr;
            k.seen_up = ps.seen(0, set);
            k.seen_down = ps.seen(1, set);
            k.gaveup = ps.gaveup();
            k.budget = ps.budget;
            // The GOPs of the sequence as `plan_groups` independent chains on a stream each (option split_streams, as the whole-frame
            // entry does): a strip's launch is two and a half rounds of the GPU's wave slots, its start and its drain are a quarter of
            // its time, and the other chain's launch fills them.  Nothing orders the chains against each other: a GOP's frames depend
            // on each other only, and so do its arrival counters and its place in the landing buffers.
            local([&] {
                const int G = e->profile ? 1 : e->plan_groups;
                if (j == 0 && G > 1) {
                    if (!e->ev_fork) HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
                    HIPCHK(hipEventRecord(e->ev_fork, s));
                    for (int gk = 1; gk < G; ++gk) {
                        if (!e->side[gk - 1]) HIPCHK(hipStreamCreateWithFlags(&e->side[gk - 1], hipStreamNonBlocking));
                        if (!e->ev_join[gk - 1]) HIPCHK(hipEventCreateWithFlags(&e->ev_join[gk - 1], hipEventDisableTiming));
                        HIPCHK(hipStreamWaitEvent(e->side[gk - 1], e->ev_fork, 0));
                    }
                }
                for (int gk = 0; gk < G; ++gk)
                    run_step_peer(e, gk == 0 ? s : e->side[gk - 1], (size_t)j, G > 1 ? gk : -1, xchg && q.up ? ps.put(0, par) : nullptr,
                                  xchg && q.down ? ps.put(1, par) : nullptr, q.up ? ps.got(0, par ^ 1u) : nullptr, q.down ? ps.got(1, par ^ 1u) : nullptr, k);
                if (j + 1 == q.steps && G > 1)
                    for (int gk = 1; gk < G; ++gk) {
                        HIPCHK(hipEventRecord(e->ev_join[gk - 1], e->side[gk - 1]));
                        HIPCHK(hipStreamWaitEvent(s, e->ev_join[gk - 1], 0));
                    }
            });
            continue;
        }
        if (q.world > 1 && q.fused) {
            // EDGE(j) and interior(j) both need ALL of step j-1 on this strip; the neighbours' rows only EDGE(j) - and it follows
            // the receive in stream order.  Two launches, two event records, two waits and one exchange per step.
            local([&] {
                HIPCHK(hipStreamWaitEvent(side, j == 0 ? e->ev_done : e->ev_edges, 0));
                if (j > 0) HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));
                run_step_edges_fused(e, s, (size_t)j, xchg && q.up ? q.send_up : nullptr, xchg && q.down ? q.send_down : nullptr,
                                     q.up ? q.recv_up : nullptr, q.down ? q.recv_down : nullptr);
                mark(s);
                HIPCHK(hipEventRecord(e->ev_edges, s));
                run_step_rows(e, side, (size_t)j, q.row0 + 1, q.row1 - 1);
                mark(side);
                HIPCHK(hipEventRecord(e->ev_interior, side));
            });
            if (xchg) exchange(nbytes, s);
            local([&] {
                mark(s);
                if (j + 1 == q.steps) HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));      // the scans follow on the main stream
            });
            continue;
        }
        if (xchg) {
            // the general form (option conformant / dct_mfma = 0 / the debug library's keep_recon): edge rows, pack kernel, exchange
            // on a stream of its own beside the interior rows, unpack kernel
            local([&] {
                HIPCHK(hipStreamWaitEvent(side, e->ev_done, 0));    // the previous step, neighbour rows included
                run_step_rows(e, s, (size_t)j, q.row0, q.row0 + 1);
                if (q.row1 - q.row0 >= 2) run_step_rows(e, s, (size_t)j, q.row1 - 1, q.row1);
                timer_break(e);
                launch_halo_pack(e, s, e->d_lists.p + e->plan_steps[(size_t)j].off_h, n_h, q.up ? q.send_up : nullptr, q.down ? q.send_down : nullptr);
                HIPCHK(hipGetLastError());
                mark(s);
                HIPCHK(hipEventRecord(e->ev_edges, s));
                run_step_rows(e, side, (size_t)j, q.row0 + 1, q.row1 - 1);
                mark(side);
                HIPCHK(hipEventRecord(e->ev_interior, side));
                HIPCHK(hipStreamWaitEvent(e->comm_stream, e->ev_edges, 0));
            });
            exchange(nbytes, fail ? s : e->comm_stream);
            local([&] {
                HIPCHK(hipEventRecord(e->ev_halo, e->comm_stream));
                HIPCHK(hipStreamWaitEvent(s, e->ev_halo, 0));
                HIPCHK(hipStreamWaitEvent(s, e->ev_interior, 0));
                mark(s);
                timer_break(e);
                launch_halo_unpack(e, s, e->d_lists.p + e->plan_steps[(size_t)j].off_h, n_h, q.up ? q.recv_up : nullptr, q.down ? q.recv_down : nullptr);
                HIPCHK(hipGetLastError());
            });
        } else {
            local([&] { run_step(e, s, (size_t)j); });
        }
        if (q.world > 1) local([&] { HIPCHK(hipEventRecord(e->ev_done, s)); });
    }
    // ---- this strip's slices, their sizes; everybody's sizes ----
    local([&] {
        timer_break(e);
        ctl_begin(e, (unsigned long long)q.strip_cap, true);
        if (q.peer) {
            // this strip's k_frame_scan also settles the peer form's accounts (PeerScan): the retry mark, the give-up word, the NEXT
            // sequence's arrival counters
            PeerState &ps = *q.peer;
            e->scan_peer_gaveup = ps.gaveup();
            e->scan_peer_clear = (unsigned int *)(ps.block + PeerState::off_cnt((unsigned)((ps.seq + 1) & 1), 0));
            // every line ANY earlier sequence of this communicator counted on: the set was last used two sequences ago, possibly by a
            // sequence of more GOPs than this one (8 GOPs, 1 GOP, 8 GOPs: slots 1..7 of the first must not greet the third with old counts)
            ps.lines_used = std::max(ps.lines_used, 2 * std::max(1, halo_frames_of_step(q.nf, q.gop, 0)));
            e->scan_peer_lines = ps.lines_used;
            e->scan_peer_mark = kStripRetry;
        }
        finish_chunk(e, s, false, false, e->d_strip_own.p);
        e->frames_total -= q.nf;                            // (host state is the caller's business: a recorded sequence is replayed without this code)
        HIPCHK(hipMemcpyAsync(e->h_strip, e->d_frame_off.p, (q.nf + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(e->h_strip + (q.nf + 1) * sizeof(unsigned long long), e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    });
    if (with_sizes) strip_enqueue_sizes(e, s, q, fail);
}

// One strip sequence between its launch and its collection (m2v_enc::flight).  m2v_strip_encode = launch + collect in one call;
// m2v_strip_encode_begin = launch with the sizes all-gather DEFERRED, m2v_strip_encode_end = that all-gather + collect.
//
// Why the all-gather waits for _end: two handles taking turns from one thread on ONE base communicator issue their collectives in one
// deterministic order on every rank - and RCCL runs the operations of a communicator in the order they were issued, whatever stream
// each is on.  begin(A) begin(B) end(A) must therefore not put B's all-gather (which sits behind ALL of B's kernels) in front of A's
// gather: issued in _end the order is sizes(A), strips(A), sizes(B), strips(B) ... - the order they become ready in.
static int strip_launch_attempt(m2v_enc *e, StripFlight &F)
{
    using clk = std::chrono::steady_clock;
    StripEncodeArgs *a = &F.a;
    StripSeq &q = F.q;
    const int rank = a->rank, world = a->world;
    int &fail = F.fail;
    std::string &fail_text = F.fail_text;
    auto local = [&](auto &&fn) {
        if (fail) return;
        try { fn(); }
        catch (const HipError &h) {
            fail = h.e == hipErrorOutOfMemory ? M2V_E_NOMEM : M2V_E_HIP;
            fail_text = std::string(h.what) + ": " + hipGetErrorString(h.e);
            (void)hipGetLastError();
        }
    };
    const Geom &full = F.full;
    const size_t nf = a->n;
    const uint32_t gop = q.gop;
    PeerState *const pst = F.pst;
    q.peer = F.use_peer ? pst : nullptr;
    int r = M2V_OK;
    local([&] {
        StripBeginArgs b{a->xs, a->ys, a->pf, a->d_in, a->n, q.row0, q.row1, a->s};
        r = strip_begin_impl(e, &b);
    });
    if (r < 0) return r;                                    // (parameters: the same answer on every rank)
    hipStream_t &s = F.s;
    s = e->strip_active ? e->strip_stream : (a->s ? a->s : e->stream);
    local([&] {
        if ((int)e->plan_steps.size() != q.steps) throw HipError{hipErrorInvalidValue, "strip plan and step count disagree"};
        for (int j = 0; j < q.steps; ++j)
            if (e->plan_steps[(size_t)j].n_h != halo_frames_of_step(nf, gop, j)) throw HipError{hipErrorInvalidValue, "strip plan and halo list disagree"};
        e->d_strip_own.ensure(q.strip_cap);
        ensure_pinned(e->h_strip, e->h_strip_cap, (nf + 1) * sizeof(unsigned long long) + sizeof(StreamCtl));
        e->d_ctl.ensure(1);
        if (world > 1 && !q.peer) {
            // (the general form's exchange stream only when that form runs: every stream a process creates moves the others around the
            // handful of hardware queues)
            if (!q.fused && !e->comm_stream) HIPCHK(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
            if (!e->ev_edges) HIPCHK(hipEventCreateWithFlags(&e->ev_edges, hipEventDisableTiming));
            if (!e->ev_halo) HIPCHK(hipEventCreateWithFlags(&e->ev_halo, hipEventDisableTiming));
            // The edge rows run as ONE launch of the instantiation that also fills the halo buffers (no pack kernel), the interior rows at
            // the same time on a second stream: a strip of an 8-GPU job is ~20 000 wavefronts per step, 2.5 rounds of the wave slots -
            // edge rows first and alone would hold the whole GPU for one macroblock lifetime at a third of its slots.
            if (!e->side[0]) HIPCHK(hipStreamCreateWithFlags(&e->side[0], hipStreamNonBlocking));
            if (!e->ev_done) HIPCHK(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
            if (!e->ev_interior) HIPCHK(hipEventCreateWithFlags(&e->ev_interior, hipEventDisableTiming));
        }
    });

    // ---- the sequence: recorded graph, or call by call ----
    // profile: GPU events around the exchange of every step: halo_total = edge rows (and their halo) written .. neighbour rows and
    // interior rows both there; halo_exposed = how much of that came after the interior rows were done
    F.marks.clear();
    double us_in_comm = 0;                 // host time inside the communicator (a local communicator blocks there until the neighbour thread has posted)
    const auto t_loop = clk::now();
    m2v_enc::StripGraph &sg = e->strip_graph;
    int &graph_used = F.graph_used;
    graph_used = 0;
    // (the general form - options conformant / dct_mfma = 0 - is enqueued call by call: it issues the exchange on a stream of its own,
    // and RCCL 2.26 crashes when its send / recv group is recorded on a stream that joined the recording through an event)
    // Automatic (the default): world == 1 and the single-GPU timing communicators.  Between the ranks of a real RCCL job the sequence
    // is enqueued call by call unless the caller opts in with option strip_graph = 1: a recording with cross-rank ncclSend / ncclRecv
    // inside has never run on hardware, the ranks of a job do not necessarily record on the same call, and on one GPU the recorded
    // form is no faster for an inner rank (profiles/r04_experiments.txt item 1).  The peer form is never recorded: its counter set
    // and its launch arguments change from sequence to sequence, and it is five launches per step shorter to begin with.  Nor is a
    // sequence whose sizes all-gather is deferred (m2v_strip_encode_begin): the recording ends with that all-gather.
    const bool graph_wanted = e->strip_graph_opt > 0 || (e->strip_graph_opt < 0 && (!a->comm || a->comm->graph_by_default()));
    // (a recording of more than one rank's form has parallel branches - edge rows and interior rows on a stream each; with the runtime limited
    // to ONE hardware queue, GPU_MAX_HW_QUEUES=1, hipGraphLaunch of such a graph crashes inside the runtime, hip::Graph::UpdateStreams: found
    // by running the suite with that setting, profiles/r05_experiments.txt item 16 - so no recording there)
    static const bool one_queue = [] { const char *v = getenv("GPU_MAX_HW_QUEUES"); return v && atoi(v) == 1; }();
    const bool graph_ok = graph_wanted && !F.defer_sizes && !q.peer && !sg.broken && !fail && !e->profile && (world == 1 || (q.fused && !one_queue)) &&
                          (!a->comm || a->comm->capturable());
    if (graph_ok) {
        // everything a recording references exists before the key (which holds the allocation generation) is taken: the output rank's
        // assembly tables are allocated here, not after the host wait - a rank must not find its own recording stale on the next call
        if (rank == a->dst) local([&] { e->d_segs.ensure(nf * (size_t)world * sizeof(CopySeg) + 16); e->d_frame_pos.ensure(nf + 1); });
    }
    if (graph_ok && !fail) {
        const std::vector<unsigned long long> key = {alloc_generation().load(), (unsigned long long)full.W, (unsigned long long)full.H, (unsigned long long)full.Q,
            (unsigned long long)q.row0, (unsigned long long)q.row1, (unsigned long long)nf, (unsigned long long)gop, (unsigned long long)rank,
            (unsigned long long)world, (unsigned long long)(uintptr_t)a->comm, (unsigned long long)q.fused, (unsigned long long)e->VL,
            (unsigned long long)e->conformant, (unsigned long long)e->dct_mfma};
        if (sg.exec && sg.key == key) graph_used = 1;
        else if (sg.seen == key) {
            // the second call of this shape: record.  (Not the first: RCCL sets up its connections to a peer inside the first send /
            // recv that uses them, which is no business of a recording, and a one-off call would pay for a graph it never launches.)
            if (sg.exec) { (void)hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
            hipGraph_t graph = nullptr;
            bool began = false;
            try {
                HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
                began = true;
                int none = 0;
                std::string none_text;
                strip_enqueue_sequence(e, s, q, true, none, none_text, nullptr, nullptr);
                began = false;
                HIPCHK(hipStreamEndCapture(s, &graph));
                HIPCHK(hipGraphInstantiate(&sg.exec, graph, nullptr, nullptr, 0));
                (void)hipGraphDestroy(graph);
                sg.key = key;
                sg.captures++;
                graph_used = 1;
            } catch (const std::exception &ex) {            // a communicator that cannot be recorded after all
                if (began) { (void)hipStreamEndCapture(s, &graph); }
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                if (sg.exec) { (void)hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
                sg.broken = true;
                e->set_err("strip_graph: recording failed (%s); the sequence is enqueued call by call from now on", ex.what());
            } catch (const HipError &h) {
                if (began) { (void)hipStreamEndCapture(s, &graph); }
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipGetLastError();
                if (sg.exec) { (void)hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
                sg.broken = true;
                e->set_err("strip_graph: recording failed (%s: %s); the sequence is enqueued call by call from now on", h.what, hipGetErrorString(h.e));
            }
        }
        sg.seen = key;
    }
    if (graph_used) {
        HIPCHK(hipGraphLaunch(sg.exec, s));
        sg.launches++;
    } else {
        strip_enqueue_sequence(e, s, q, false, fail, fail_text, e->profile && !q.peer ? &F.marks : nullptr, &us_in_comm, !F.defer_sizes);
        // what this call allocated on the way (the launches' block tables on a shape's first call) belongs to the shape that was seen:
        // the next call of it finds everything in place and records
        if (graph_ok && !fail && !sg.seen.empty()) sg.seen[0] = alloc_generation().load();
    }
    // host state the sequence leaves behind, recorded or not
    if (e->strip_active) {
        e->frames_total += nf;
        e->strip_nf = nf;
        strip_close(e);
    }
    if (q.peer) { pst->seq++; pst->sequences++; }           // (the same on every rank, whatever became of this rank's local work)
    e->strip_stats.steps = q.steps;
    e->strip_stats.graph = graph_used;
    e->strip_stats.peer = q.peer ? 1 : 0;
    e->strip_stats.host_us_per_step = std::chrono::duration<double, std::micro>(clk::now() - t_loop).count() / std::max(1, q.steps);
    e->strip_stats.comm_us_per_step = us_in_comm / std::max(1, q.steps);
    return M2V_OK;
}

// parameters, the strip's rows, the buffers every exchange touches, the first attempt
static int strip_launch(m2v_enc *e, StripFlight &F, const StripEncodeArgs &args, bool defer_sizes)
{
    F = StripFlight{};
    e->strip_inflight = false;
    F.a = args;
    F.defer_sizes = defer_sizes;
    F.halves = defer_sizes;
    StripEncodeArgs *a = &F.a;
    const int rank = a->rank, world = a->world;
    F.full = make_geom(e, a->xs, a->ys);
    const Geom &full = F.full;
    if (world < 1 || world > kMaxStripRanks || world > full.mbh || rank < 0 || rank >= world || a->dst < 0 || a->dst >= world ||
        (world > 1 && (!a->comm || a->comm->world != world))) {
        e->set_err("m2v_strip_encode: bad rank / world / communicator");
        return M2V_E_PARAM;
    }
    if (e->state != m2v_enc::IDLE || e->strip_active || e->resident_inflight) { e->set_err("m2v_strip_encode: encoder busy"); return M2V_E_STATE; }
    // contiguous strips, sizes differing by at most one row, the first mbh % world ranks get the extra row (parallel.partition_rows)
    const int base = full.mbh / world, rem = full.mbh % world;
    const int row0 = rank * base + std::min(rank, rem), row1 = row0 + base + (rank < rem ? 1 : 0);
    const size_t nf = a->n;
    const uint32_t gop = (a->pf & 0xFFu) + 1u;

    // ---- plan and buffers.  Everything up to here was the same on every rank; from here on a failure is this rank's alone and
    //      must not break the collective call order (see the head of this section) ----
    if (kDebug && (e->ablate & (1 << 21))) { F.fail = M2V_E_HIP; F.fail_text = "injected failure (ablate bit 21)"; }    // -DM2V_DEBUG: the failure protocol under test
    // a bad output buffer is the output rank's alone to know: a local failure like any other (the other ranks are told through the
    // size table, nobody is left waiting in an exchange); with one rank it is simply the answer
    if (rank == a->dst && (!a->d_out || ((uintptr_t)a->d_out & 15u) != 0)) {
        const char *why = !a->d_out ? "the output rank needs d_out" : "d_out must be 16-byte aligned";
        if (world == 1) { e->set_err("m2v_strip_encode: %s", why); return M2V_E_PARAM; }
        F.fail = M2V_E_PARAM; F.fail_text = why;
    }
    StripSeq &q = F.q;
    q.comm = a->comm; q.rank = rank; q.world = world; q.row0 = row0; q.row1 = row1;
    q.up = row0 > 0; q.down = row1 < full.mbh;
    q.fused = !e->conformant && e->dct_mfma && !e->keep_recon;
    q.gop = gop; q.nf = nf; q.W = full.W;
    q.steps = (int)std::min<size_t>(gop, nf);
    q.strip_cap = nf * ((size_t)(row1 - row0) * full.mbw * 1216 + (size_t)(row1 - row0) * 8 + 64) + 256;     // worst case
    const size_t halo_cap = (size_t)halo_frames_of_step(nf, gop, 0) * (size_t)(3 * e->VL) * (size_t)full.W;
    // the buffers an exchange touches come first: with them a rank can keep the call order whatever else fails
    e->d_halo.ensure(4 * halo_cap + 64);
    e->d_frame_off.ensure(nf + 1);
    if (world > 1) {
        e->d_alloff.ensure((size_t)world * (nf + 1));
        ensure_pinned(e->h_asm, e->h_asm_cap, (size_t)world * (nf + 1) * sizeof(unsigned long long));
    }
    q.send_up = e->d_halo.p; q.send_down = q.send_up + halo_cap; q.recv_up = q.send_down + halo_cap; q.recv_down = q.recv_up + halo_cap;
    // peer transport (m2v_comm.hpp, PeerComm): the usual form of the step, a connected communicator that has not fallen back, and a
    // step's rows fitting its landing buffers - all of it the same on every rank
    F.pst = a->comm ? a->comm->peer() : nullptr;
    PeerState *const pst = F.pst;
    F.use_peer = pst && pst->connected && !pst->degraded && world > 1 && q.fused && halo_cap <= pst->cap && pst->world == world && pst->rank == rank &&
                 halo_frames_of_step(nf, gop, 0) <= kPeerSlots;
    q.mbw = full.mbw;
    const int r = strip_launch_attempt(e, F);
    if (r < 0) return r;
    e->strip_inflight = true;
    return M2V_OK;
}

// the one host wait; strips to the output rank; final assembly
static int strip_collect(m2v_enc *e, StripFlight &F, size_t *bytes)
{
    StripEncodeArgs *a = &F.a;
    StripSeq &q = F.q;
    const int rank = a->rank, world = a->world;
    const size_t nf = a->n;
    const Geom &full = F.full;
    PeerState *const pst = F.pst;
    int &fail = F.fail;
    auto local = [&](auto &&fn) {
        if (fail) return;
        try { fn(); }
        catch (const HipError &h) {
            fail = h.e == hipErrorOutOfMemory ? M2V_E_NOMEM : M2V_E_HIP;
            F.fail_text = std::string(h.what) + ": " + hipGetErrorString(h.e);
            (void)hipGetLastError();
        }
    };
    e->strip_inflight = false;
    hipStream_t s = F.s;
    hipEvent_t g0 = nullptr, g1 = nullptr;
    const void *strips[kMaxStripRanks] = {};
    const unsigned long long *d_all = nullptr;
    int failed_rank = -1;
    // Twice at most: a peer sequence in which some rank's wait ran out of budget (every rank reads that in the all-gathered sizes) is
    // encoded again, exchanged through the base communicator, and the communicator stays there.
    for (int attempt = 0;; ++attempt) {
        if (F.defer_sizes) { strip_enqueue_sizes(e, s, q, fail); F.defer_sizes = false; }      // (see the head of this section)
        g0 = g1 = nullptr;
        if (e->profile && !fail) { g0 = pool_event(e); timer_break(e); HIPCHK(hipEventRecord(g0, s)); }     // from here: gather, final assembly
        for (auto &sp : strips) sp = nullptr;
        d_all = e->d_frame_off.p;
        failed_rank = fail ? rank : -1;
        if (world > 1) {
            const hipError_t se = hipStreamSynchronize(s);      // the sizes decide the receive counts
            if (se != hipSuccess && !fail) throw HipError{se, "hipStreamSynchronize(strip sequence)"};
            const unsigned long long *all = (const unsigned long long *)e->h_asm;
            int retry_rank = -1;
            for (int k = 0; k < world; ++k) {
                const unsigned long long mark = all[(size_t)k * (nf + 1) + nf];
                if (mark == kStripPoison && failed_rank < 0) failed_rank = k;
                if (mark == kStripRetry && retry_rank < 0) retry_rank = k;
            }
            if (failed_rank < 0 && retry_rank >= 0) {
                if (!q.peer || attempt > 0) throw HipError{hipErrorUnknown, "a retry mark in the size table of a sequence that was not in the peer form"};
                // no error: the sequence again, the rows exchanged through the base communicator - on every rank, they all read the same table
                pst->degraded = true;
                pst->giveups++;
                F.use_peer = false;
                collect_timers(e);
                const int r = strip_launch_attempt(e, F);
                if (r < 0) return r;
                continue;
            }
            if (failed_rank < 0) {
                // (an overflow of this strip's buffer - impossible with the worst-case size above - is reported at the end: the other ranks
                // are waiting in the gather, and a rank that left now would leave them there)
                size_t sizes[kMaxStripRanks] = {}, total_in = 0;
                for (int k = 0; k < world; ++k) {
                    sizes[k] = (size_t)all[(size_t)k * (nf + 1) + nf];
                    if (k != a->dst) total_in += (sizes[k] + 255) & ~(size_t)255;
                }
                void *bufs[kMaxStripRanks] = {};
                if (rank == a->dst) {
                    local([&] { e->d_gather.ensure(total_in + 256); });
                    size_t off = 0;
                    for (int k = 0; k < world; ++k) {
                        if (k == a->dst) { strips[k] = e->d_strip_own.p; continue; }
                        bufs[k] = e->d_gather.p + off;
                        strips[k] = bufs[k];
                        off += (sizes[k] + 255) & ~(size_t)255;
                    }
                }
                if (fail) throw HipError{hipErrorOutOfMemory, "no room for the other ranks' strips on the output rank"};     // (they are already sending: nothing to keep in order any more)
                a->comm->gather(rank, a->dst, e->d_strip_own.p, sizes, bufs, s);
                d_all = e->d_alloff.p;
            }
        } else {
            strips[0] = e->d_strip_own.p;
        }
        break;
    }   // attempt
    if (failed_rank >= 0) {
        // (a rank that did not get as far as its scans has not cleared the next sequence's arrival counters either: the peer form is
        // off for this communicator from here on - on every rank, they all read the same table)
        if (pst) pst->degraded = true;
        (void)hipStreamSynchronize(s);
        collect_timers(e);
        if (fail) { e->set_err("m2v_strip_encode: %s", F.fail_text.c_str()); return fail; }
        e->set_err("m2v_strip_encode: rank %d of the job failed", failed_rank);
        return M2V_E_HIP;
    }
    size_t out_bytes = 0;
    if (rank == a->dst) {
        strip_assemble_enqueue(e, s, full, a->pf, nf, world, strips, d_all, a->d_out, a->cap);
        if (e->profile) { g1 = pool_event(e); timer_break(e); HIPCHK(hipEventRecord(g1, s)); }
        if (!e->st().h_ctl) HIPCHK(hipHostMalloc((void **)&e->st().h_ctl, 2 * sizeof(StreamCtl)));
        HIPCHK(hipMemcpyAsync(e->st().h_ctl, e->d_ctl.p, sizeof(StreamCtl), hipMemcpyDeviceToHost, s));
    } else if (e->profile) { g1 = pool_event(e); timer_break(e); HIPCHK(hipEventRecord(g1, s)); }
    // The final wait.  The output rank's caller reads d_out next; the blocking call promises a synchronised stream.  A rank that only SENT its
    // strip leaves m2v_strip_encode_end without it: everything the host reads (sizes, this strip's control word) was complete at the wait for
    // the sizes, the send reads d_strip_own on this handle's stream and the handle's next sequence is ordered behind it there - and a send
    // completes when the output rank has posted its receive, so waiting for it would tie every rank's host to the output rank's pace,
    // sequence by sequence, which is exactly what sequences in flight are there to undo.
    const bool final_wait = rank == a->dst || !F.halves || e->profile;
    if (final_wait) HIPCHK(hipStreamSynchronize(s));
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
        const std::vector<hipEvent_t> &marks = F.marks;
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
    if (final_wait) {
        collect_timers(e);
        e->strip_stream = nullptr;                           // synchronised above: nothing of the sequence is left on the caller's stream
    }                                                        // (else: m2v_reset / m2v_destroy / the next sequence's waits take care of it)
    if (bytes) *bytes = out_bytes;
    return M2V_OK;
}

static int strip_encode_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripEncodeArgs *)argp;
    if (e->strip_inflight) { e->set_err("m2v_strip_encode: a strip sequence is in flight on this handle (m2v_strip_encode_end first)"); return M2V_E_STATE; }
    StripFlight &F = flight_of(e);
    const int r = strip_launch(e, F, *a, false);
    if (r < 0) return r;
    return strip_collect(e, F, a->bytes);
}

static int strip_encode_begin_impl(m2v_enc *e, void *argp)
{
    auto *a = (StripEncodeArgs *)argp;
    if (e->strip_inflight) { e->set_err("m2v_strip_encode_begin: a strip sequence is in flight on this handle already"); return M2V_E_STATE; }
    return strip_launch(e, flight_of(e), *a, true);
}

static int strip_encode_end_impl(m2v_enc *e, void *argp)
{
    if (!e->strip_inflight) { e->set_err("m2v_strip_encode_end: nothing in flight"); return M2V_E_STATE; }
    return strip_collect(e, flight_of(e), (size_t *)argp);
}

int m2v_strip_encode(m2v_enc *e, m2v_comm *comm, int rank, int world, int dst_rank, uint32_t xsize16, uint32_t ysize16,
                     uint32_t pframes_count, const void *d_frames444, size_t nframes, void *d_out, size_t cap, size_t *out_bytes,
                     void *hip_stream)
{
    if (!e || !d_frames444 || nframes == 0) return M2V_E_PARAM;
    StripEncodeArgs a{comm, rank, world, dst_rank, xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, (uint8_t *)d_out, cap,
                      out_bytes, (hipStream_t)hip_stream};
    const bool was_inflight = e->strip_inflight;
    const int r = guard(e, strip_encode_impl, &a);
    if (r < 0 && !was_inflight) e->strip_inflight = false;
    if (r < 0 && e->strip_active) strip_close(e);            // a failed sequence does not leave the handle in strip mode
    if (r < 0 && comm && !was_inflight) comm->abort();        // ... and the other ranks of an in-process communicator do not wait for it for ever
    return r;
}

int m2v_strip_encode_begin(m2v_enc *e, m2v_comm *comm, int rank, int world, int dst_rank, uint32_t xsize16, uint32_t ysize16,
                           uint32_t pframes_count, const void *d_frames444, size_t nframes, void *d_out, size_t cap, void *hip_stream)
{
    if (!e || !d_frames444 || nframes == 0) return M2V_E_PARAM;
    StripEncodeArgs a{comm, rank, world, dst_rank, xsize16, ysize16, pframes_count, (const uint8_t *)d_frames444, nframes, (uint8_t *)d_out, cap,
                      nullptr, (hipStream_t)hip_stream};
    const bool was_inflight = e->strip_inflight;
    const int r = guard(e, strip_encode_begin_impl, &a);
    if (r < 0 && !was_inflight) {
        e->strip_inflight = false;
        if (e->strip_active) strip_close(e);
        if (comm) comm->abort();
    }
    return r;
}

int m2v_strip_encode_end(m2v_enc *e, size_t *out_bytes)
{
    if (!e) return M2V_E_PARAM;
    m2v_comm *comm = e->strip_inflight && e->flight ? e->flight->a.comm : nullptr;
    const int r = guard(e, strip_encode_end_impl, out_bytes);
    if (r < 0 && comm) {                                     // (comm: there was a sequence to collect)
        e->strip_inflight = false;
        if (e->strip_active) strip_close(e);
        comm->abort();
    }
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

int m2v_strip_graph_stats(const m2v_enc *e, int *last_call_was_graph, int *recordings, int *launches)
{
    if (!e) return M2V_E_PARAM;
    if (last_call_was_graph) *last_call_was_graph = e->strip_stats.graph;
    if (recordings) *recordings = e->strip_graph.captures;
    if (launches) *launches = e->strip_graph.launches;
    return e->strip_graph.broken ? 1 : 0;
}

int m2v_strip_last_form(const m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    return e->strip_stats.peer ? 2 : e->strip_stats.graph ? 1 : 0;
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

static m2v_comm *init_solo(int world, bool with_rccl, int *err)
{
    if (world < 1 || world > kMaxStripRanks) { t_comm_err = "m2v_comm_init_solo: 1..16 ranks"; if (err) *err = M2V_E_PARAM; return nullptr; }
    try {
        m2v_comm *c = new SoloComm(world, with_rccl);
        if (err) *err = M2V_OK;
        return c;
    } catch (const std::bad_alloc &) {
        t_comm_err = "host allocation failed";
        if (err) *err = M2V_E_NOMEM;
    } catch (const std::exception &ex) {
        t_comm_err = ex.what();
        if (err) *err = M2V_E_HIP;
    }
    return nullptr;
}

m2v_comm *m2v_comm_init_solo(int world, int *err) { return init_solo(world, false, err); }
m2v_comm *m2v_comm_init_solo_rccl(int world, int *err) { return init_solo(world, true, err); }

m2v_comm *m2v_comm_init_local(int world, int *err)
{
    if (world < 1 || world > LocalComm::kMax) { t_comm_err = "m2v_comm_init_local: 1..16 ranks"; if (err) *err = M2V_E_PARAM; return nullptr; }
    m2v_comm *c = new (std::nothrow) LocalComm(world);
    if (err) *err = c ? M2V_OK : M2V_E_NOMEM;
    return c;
}

m2v_comm *m2v_comm_init_callbacks(int world, const m2v_comm_callbacks *cb, int *err)
{
    if (world < 1 || world > kMaxStripRanks || !cb || !cb->halo || !cb->allgather_u64 || !cb->gather) {
        t_comm_err = "m2v_comm_init_callbacks: 1..16 ranks and all three functions";
        if (err) *err = M2V_E_PARAM;
        return nullptr;
    }
    m2v_comm *c = new (std::nothrow) CallbackComm(world, *cb);
    if (err) *err = c ? M2V_OK : M2V_E_NOMEM;
    return c;
}

// ---- the peer transport (m2v_comm.hpp, PeerComm) ----
static int comm_call(const char *what, const std::function<void()> &fn)
{
    try {
        fn();
        return M2V_OK;
    } catch (const std::bad_alloc &) {
        t_comm_err = std::string(what) + ": host allocation failed";
        return M2V_E_NOMEM;
    } catch (const std::exception &ex) {
        t_comm_err = std::string(what) + ": " + ex.what();
        (void)hipGetLastError();
        return M2V_E_HIP;
    }
}

m2v_comm *m2v_comm_init_peer(m2v_comm *base, int rank, int device, size_t halo_bytes, int *err)
{
    auto fail = [&](int code, const std::string &why) -> m2v_comm * { t_comm_err = why; if (err) *err = code; return nullptr; };
    if (!base || rank < 0 || rank >= base->world) return fail(M2V_E_PARAM, "m2v_comm_init_peer: a base communicator and a rank inside it");
    if (base->peer()) return fail(M2V_E_PARAM, "m2v_comm_init_peer: the base communicator is a peer communicator itself");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(M2V_E_NODEVICE, "m2v_comm_init_peer: device ordinal out of range");
    m2v_comm *c = nullptr;
    const int r = comm_call("m2v_comm_init_peer", [&] { c = new PeerComm(base, rank, device, halo_bytes ? halo_bytes : (size_t)4 << 20); });
    if (err) *err = r;
    return c;
}

static PeerComm *as_peer(m2v_comm *c)
{
    if (!c || !c->peer()) { t_comm_err = "not a peer communicator (m2v_comm_init_peer)"; return nullptr; }
    return static_cast<PeerComm *>(c);           // peer() is non-null for PeerComm only
}

int m2v_comm_peer_export(m2v_comm *c, void *desc, size_t cap)
{
    PeerComm *p = as_peer(c);
    if (!p || !desc || cap < sizeof(PeerDesc)) return M2V_E_PARAM;
    const int r = comm_call("m2v_comm_peer_export", [&] { PeerDesc d; p->export_desc(d); memcpy(desc, &d, sizeof d); });
    return r < 0 ? r : (int)sizeof(PeerDesc);
}

int m2v_comm_peer_connect(m2v_comm *c, const void *desc_up, const void *desc_down)
{
    PeerComm *p = as_peer(c);
    if (!p) return M2V_E_PARAM;
    return comm_call("m2v_comm_peer_connect", [&] {
        PeerDesc u, d;
        if (desc_up) memcpy(&u, desc_up, sizeof u);
        if (desc_down) memcpy(&d, desc_down, sizeof d);
        p->connect(desc_up ? &u : nullptr, desc_down ? &d : nullptr, false);
    });
}

int m2v_comm_peer_connect_all(m2v_comm *c)
{
    PeerComm *p = as_peer(c);
    if (!p) return M2V_E_PARAM;
    return comm_call("m2v_comm_peer_connect_all", [&] { p->connect_all(); });
}

int m2v_comm_peer_stats(m2v_comm *c, unsigned long long *peer_sequences, unsigned long long *giveups)
{
    if (!c || !c->peer()) return M2V_E_PARAM;
    const PeerState &st = *c->peer();
    if (peer_sequences) *peer_sequences = st.sequences;
    if (giveups) *giveups = st.giveups;
    return st.degraded ? 1 : 0;
}

const char *m2v_comm_kind(const m2v_comm *c) { return c ? c->kind() : ""; }

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

// The same pair recorded into a hipGraph and launched `launches` times: whether this transport can be part of the recorded strip
// sequence of m2v_strip_encode.  The stream is synchronised before returning.
int m2v_comm_selftest_captured(m2v_comm *c, int rank, const void *d_send, void *d_recv, size_t nbytes, void *hip_stream, int launches)
{
    if (!c || !d_send || !d_recv || launches < 1) return M2V_E_PARAM;
    if (!c->capturable()) { t_comm_err = std::string("a '") + c->kind() + "' communicator blocks on other threads: it cannot be recorded"; return M2V_E_STATE; }
    hipStream_t s = (hipStream_t)hip_stream, own = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int r = M2V_OK;
    bool began = false;
    try {
        if (!s) { M2V_COMM_HIP(hipStreamCreateWithFlags(&own, hipStreamNonBlocking)); s = own; }
        c->loopback(rank, d_send, d_recv, nbytes, s);           // connections are set up by the first use, outside any recording
        M2V_COMM_HIP(hipStreamSynchronize(s));
        M2V_COMM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        began = true;
        c->loopback(rank, d_send, d_recv, nbytes, s);
        began = false;
        M2V_COMM_HIP(hipStreamEndCapture(s, &graph));
        M2V_COMM_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int k = 0; k < launches; ++k) M2V_COMM_HIP(hipGraphLaunch(exec, s));
        M2V_COMM_HIP(hipStreamSynchronize(s));
    } catch (const std::exception &ex) {
        if (began) (void)hipStreamEndCapture(s, &graph);
        (void)hipGetLastError();
        t_comm_err = ex.what();
        r = M2V_E_HIP;
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (own) (void)hipStreamDestroy(own);
    return r;
}

const char *m2v_comm_last_error(void) { return t_comm_err.c_str(); }

}  // extern "C"

void m2v::strip_flight_release(m2v_enc *e)
{
    delete e->flight;
    e->flight = nullptr;
    e->strip_inflight = false;
}
