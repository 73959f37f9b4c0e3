// m2v_core.hip — handle life cycle and the chunk plan of libm2v_mi355x.so.
//
// Sequence control mirrors stage A of the RTL (RTL:1027-1095): the configuration is latched on the first beat, beats fill
// raster-order frames, i_sequence_stop black-fills the frame in progress, and the stream ends with sequence_end_code + one final
// zero-padded 32-byte word.  Unlike the 64-clock/macroblock RTL pipeline, frames are buffered and encoded in chunks: closed GOPs
// (closed_gop = 1, RTL:2656) are independent, so frame k of every GOP in a chunk runs in the same launch - that is what fills
// 256 CUs with one wavefront per macroblock.
#include <exception>
#include <new>
#include <stdexcept>

#include "m2v_host.hpp"

namespace m2v {

// why the last m2v_create on this thread failed: there is no handle yet to carry the text (m2v_last_error(NULL))
static thread_local std::string t_create_err;

// ---------------------------------------------------------------------------------------------
// geometry (RTL:985-1006)
// ---------------------------------------------------------------------------------------------
static int clamp_size16(uint32_t s, int L)
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
    g.rysz = (uint32_t)(g.mbw + 1) * (uint32_t)g.mbh * 256u;
    g.row0 = 0;
    g.row1 = g.mbh;
    g.strip = 0;
    g.ablate = e->ablate;
    g.cu_pack = e->cu_pack;
    geom_finish(g);
    return g;
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

void collect_timers(m2v_enc *e)
{
    e->open_t.on = false;       // (an interval nobody closed before the wait that led here would include that wait: dropped)
    for (auto &t : e->timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            e->stats[t.kernel].launches += t.count;
            e->stats[t.kernel].ms += ms;
            e->stats[t.kernel].units += t.units;
        }
    }
    e->timed.clear();
    e->ev_used = 0;
    e->chain_ev = nullptr;
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
    timer_break(e);                  // copies are enqueued below: the next timer records its own start event
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
    e->rec_bytes = (size_t)g.rysz * 3 / 2;           // tiled, with one extra tile column (rec_luma_off)
    const bool need_any_rec = e->pframes > 0;
    std::vector<int> rec_slot(nf, -1);
    if (need_any_rec) {
        if (e->rec_pool_bytes < e->rec_bytes) {       // geometry grew since the pool was allocated (new sequence)
            if (e->persist_slot >= 0) throw HipError{hipErrorInvalidValue, "reference lost on geometry change"};
            ++alloc_generation();
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
    e->d_slots.ensure(nmb * (size_t)kSlotWords + 8);
    e->d_slots_small.ensure(nmb * (size_t)(kSmallSlotWords + kTinySlotWords + kMicroSlotWords) + 8);      // the 128-byte class, then the 64- and the 32-byte class
    e->g.s16_off = (uint32_t)(nmb * (size_t)kSmallSlotWords);
    e->g.s8_off = (uint32_t)(nmb * (size_t)(kSmallSlotWords + kTinySlotWords));
    e->d_mbinfo.ensure(nmb);
    e->d_mblen.ensure(nmb);
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
        for (size_t i = 0; i < lists.size(); ++i) {
            e->st().h_joblist[i] = jobs[(size_t)lists[i]];
            e->st().h_joblist[i].fidx = (uint32_t)lists[i];
        }
        // The plan goes up through ONE small kernel that reads the pinned staging itself (host memory is mapped into the device's
        // address space), not through three copy-engine transfers: on the port path those queue on the engine behind the NEXT chunk's
        // frames - a millisecond of upload - before this chunk's first kernel can start.
        launch_plan_upload(e, s, e->st().h_jobs, nf, e->st().h_lists, e->st().h_joblist, lists.size());
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

// strip mode, peer transport: GOP step j of the WHOLE strip as one launch each for the I and the P frames of the step (edge rows first
// in dispatch order; k_mb<.., EDGE, PEER>).  ps: the counters and the budget; n_edge is filled in here.
void run_step_peer(m2v_enc *e, hipStream_t s, size_t j, int group, uint8_t *put_up, uint8_t *put_down, const uint8_t *got_up, const uint8_t *got_down, PeerStep ps)
{
    const m2v_enc::Step &st = e->plan_steps[j];
    // group >= 0: only the frames of that GOP group (plan_chunk's cut_i / cut_p); -1: the whole step
    const int i0 = group < 0 ? 0 : st.cut_i[group], i1 = group < 0 ? st.n_i : st.cut_i[group + 1];
    const int p0 = group < 0 ? 0 : st.cut_p[group], p1 = group < 0 ? st.n_p : st.cut_p[group + 1];
    Geom gg = e->g;
    const int r0 = e->g.row0, r1 = e->g.row1, nrows = r1 - r0 >= 2 ? 2 : 1;
    gg.rstride = nrows == 2 ? r1 - 1 - r0 : 1;
    gg.edge_top = r0;
    gg.edge_bot = r1 - 1;
    geom_finish(gg);
    ps.n_edge = (unsigned int)(nrows * gg.mbw);
    launch_mb_peer<false>(e, s, e->d_lists.p + st.off_i + i0, i1 - i0, gg, put_up, put_down, nullptr, nullptr, ps);      // an I frame has no reference
    launch_mb_peer<true>(e, s, e->d_lists.p + st.off_p + p0, p1 - p0, gg, put_up, put_down, got_up, got_down, ps);
}

void finish_chunk(m2v_enc *e, hipStream_t s, bool first, bool last, uint8_t *d_stream, bool advance)
{
    const Geom &g = e->g;
    const size_t nf = e->plan_nf;
    {
        Timer t(e, s, 4, (double)nf * g.ysz);
        if (!e->slice_scan_done) launch_slice_scan(e, s, g, 0, (int)nf);
        e->slice_scan_done = false;
        launch_frame_scan(e, s, g, nf, first, last, advance, d_stream);
        HIPCHK(hipGetLastError());
        t.stop();
    }
    {
        Timer t(e, s, 3, (double)nf * g.ysz);
        launch_assemble(e, s, g, nf, first, last, d_stream);
        HIPCHK(hipGetLastError());
        t.stop();
    }
    timer_break(e);             // what follows on this stream (control word read-back, events) is not the assembly's
    e->frames_total += nf;
}

void encode_chunk(m2v_enc *e, hipStream_t s, const uint8_t *d_frames, size_t nf, bool first, bool last,
                  uint32_t last_valid_beats, uint8_t *d_stream, bool advance)
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
        for (int k = 0; k < G; ++k) launch_slice_scan(e, k == 0 ? s : e->side[k - 1], e->g, e->plan_gf[k], e->plan_gf[k + 1]);
        HIPCHK(hipGetLastError());
        e->slice_scan_done = true;
        for (int k = 1; k < G; ++k) {
            HIPCHK(hipEventRecord(e->ev_join[k - 1], e->side[k - 1]));
            HIPCHK(hipStreamWaitEvent(s, e->ev_join[k - 1], 0));
        }
    } else {
        e->timer_merge = true;          // (option profile: the steps' launches of one kind as one timed interval)
        for (size_t j = 0; j < e->plan_steps.size(); ++j) run_step(e, s, j);
        e->timer_merge = false;
        timer_break(e);
    }
    finish_chunk(e, s, first, last, d_stream, advance);
}

// pinned host memory of at least `bytes`, kept with the handle
void ensure_pinned(uint8_t *&p, size_t &cap, size_t bytes)
{
    if (cap >= bytes) return;
    ++alloc_generation();
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    HIPCHK(hipHostMalloc((void **)&p, bytes * 2));
    cap = bytes * 2;
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

}  // namespace m2v

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
    e->d_gather.recorded = false;
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
    // a resident sequence enqueued on a CALLER's stream (m2v_encode_resident_begin with hip_stream != NULL) still reads and writes the
    // handle's work buffers: nothing is released under running kernels
    if (e->resident_inflight && e->resident_stream) (void)hipStreamSynchronize(e->resident_stream);
    if (e->strip_stream && e->strip_stream != e->stream) (void)hipStreamSynchronize(e->strip_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    if (e->up_stream2) (void)hipStreamSynchronize(e->up_stream2);
    for (auto sd : e->side) if (sd) (void)hipStreamSynchronize(sd);
    (void)hipGetLastError();            // (a caller's stream that no longer exists: tolerated above, and not left behind as HIP's last error)
    e->d_coef.release(); e->d_mbaux.release(); e->d_slots.release(); e->d_slots_small.release(); e->d_mbinfo.release(); e->d_mblen.release();
    e->d_slice_bytes.release(); e->d_slice_off.release(); e->d_frame_off.release();
    e->d_jobs.release(); e->d_lists.release(); e->d_joblist.release(); e->d_ctl.release(); e->d_segs.release();
    for (auto p : e->rec_pool) (void)hipFree(p);
    for (auto &c : e->mbmaps) { if (c.ev) (void)hipEventDestroy(c.ev); c.d.release(); }
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (auto &h : e->hs) {
        h.d_out.release();
        h.d_in.release();
        h.d_pk.release();
        if (h.h_pk) (void)hipHostFree(h.h_pk);
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
    if (e->strip_graph.exec) (void)hipGraphExecDestroy(e->strip_graph.exec);
    strip_flight_release(e);
    for (auto ev : e->ev_upl) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_up2) (void)hipEventDestroy(e->ev_up2);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto ev : e->ev_join) if (ev) (void)hipEventDestroy(ev);
    for (auto sd : e->side) if (sd) (void)hipStreamDestroy(sd);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
    if (e->up_stream2) (void)hipStreamDestroy(e->up_stream2);
    delete e;
}

int m2v_reset(m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    (void)hipSetDevice(e->device);
    // (a resident sequence on a caller's stream: its kernels use the work buffers the next call rewrites)
    if (e->resident_inflight && e->resident_stream) (void)hipStreamSynchronize(e->resident_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->up_stream) (void)hipStreamSynchronize(e->up_stream);
    if (e->up_stream2) (void)hipStreamSynchronize(e->up_stream2);
    for (auto sd : e->side) if (sd) (void)hipStreamSynchronize(sd);
    if (e->strip_stream && e->strip_stream != e->stream) (void)hipStreamSynchronize(e->strip_stream);
    (void)hipGetLastError();            // (a caller's stream that no longer exists: the next launch must not trip over this)
    for (auto &h : e->hs) { h.stage = 0; h.uploaded = 0; h.pk.clear(); h.pk_used = h.pk_valid = h.pk_up = 0; }
    e->dev_jobs.clear(); e->dev_lists.clear(); e->dev_jobs_p = nullptr;
    e->resident_inflight = false; e->resident_empty = false;
    e->pending.clear();
    e->state = m2v_enc::IDLE;
    e->buffered = 0; e->beat_pos = 0; e->frames_total = 0; e->persist_slot = -1;
    e->first_chunk = true; e->stream_bytes = 0; e->cur = 0;
    e->fifo.clear(); e->fifo_rd = 0; e->end_pending = false;
    // a strip sequence abandoned between m2v_strip_begin and m2v_strip_finish: back to the full frame
    e->strip_active = false;
    e->strip_inflight = false;
    e->strip_stream = nullptr;
    e->strip_nf = 0;
    if (e->comm_stream) (void)hipStreamSynchronize(e->comm_stream);
    e->plan_steps.clear();
    e->plan_nf = 0;
    e->ctl_init = 0;
    e->scan_peer_gaveup = nullptr;
    e->upl_pending[0] = e->upl_pending[1] = false;
    e->up_unsynced = false;
    e->g.row0 = 0; e->g.row1 = e->g.mbh; e->g.strip = 0;
    geom_finish(e->g);
    e->timed.clear(); e->ev_used = 0; e->chain_ev = nullptr; e->open_t.on = false;
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

int m2v_set_option(m2v_enc *e, const char *name, long long value)
{
    if (!e || !name) return M2V_E_PARAM;
    if (e->resident_inflight || e->strip_active || e->strip_inflight) {      // the sequence in flight was planned with the current options
        e->set_err("m2v_set_option: a %s sequence is in flight", e->resident_inflight ? "resident" : "strip");
        return M2V_E_STATE;
    }
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
    if (!strcmp(name, "strip_graph")) { e->strip_graph_opt = value < 0 ? -1 : value != 0; return M2V_OK; }
    if (!strcmp(name, "stream_priority")) {
        // the handle's own stream again, at another priority (-1 low, 0 normal, 1 high).  HIP keeps the hardware queues of different
        // priorities apart, so two handles with different priorities can never share one - which is what decides whether two
        // sequences in flight really overlap (bench.py; on some boxes two default-priority streams of a process land on ONE queue).
        if (value < -1 || value > 1 || e->state != m2v_enc::IDLE) return M2V_E_PARAM;
        int lo = 0, hi = 0;
        if (hipSetDevice(e->device) != hipSuccess || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return M2V_E_HIP;
        hipStream_t ns = nullptr;
        if (hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, value > 0 ? hi : value < 0 ? lo : (lo + hi) / 2) != hipSuccess) return M2V_E_HIP;
        if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
        // nothing may keep pointing at the stream that is gone (m2v_reset / m2v_destroy synchronise these two)
        if (e->strip_stream == e->stream) e->strip_stream = nullptr;
        if (e->resident_stream == e->stream) e->resident_stream = nullptr;
        e->stream = ns;
        ++alloc_generation();
        return M2V_OK;
    }
    if (!strcmp(name, "cu_pack")) { if (value < 0 || value > 8) return M2V_E_PARAM; e->cu_pack = (int)value; return M2V_OK; }
    if (!strcmp(name, "direct_upload")) {
        if (value < 0 || value > 2) return M2V_E_PARAM;
        if (e->up_stream && (e->upl_pending[0] || e->upl_pending[1])) { (void)hipStreamSynchronize(e->up_stream); if (e->up_stream2) (void)hipStreamSynchronize(e->up_stream2); }
        e->upl_pending[0] = e->upl_pending[1] = false;
        e->direct_upload = value != 0;
        e->direct_upload_deferred = value == 2;
        return M2V_OK;
    }
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
            const size_t rb = e->rec_bytes, pb = (size_t)g.ysz + 2 * (size_t)g.csz;
            bytes = e->dbg_frames * pb;
            if (bytes > a->cap) return M2V_E_OVERFLOW;
            memset(a->dst, 0, bytes);
            // the device keeps a reconstruction in shifted tiles (csrc/m2v_kernels.hpp, rec_luma_off): handed out as planar 4:2:0.
            // Luma tile tx of a tile row holds columns 16 tx - 8 .. 16 tx + 7, chroma tile tx columns 8 tx - 4 .. 8 tx + 3.
            std::vector<uint8_t> tiled(rb);
            const int tw = g.mbw + 1;
            for (size_t k = 0; k < e->dbg_frames; ++k)
                if (k < e->dbg_rec_slot.size() && e->dbg_rec_slot[k] >= 0) {
                    HIPCHK(hipMemcpy(tiled.data(), e->rec_pool[e->dbg_rec_slot[k]], rb, hipMemcpyDeviceToHost));
                    uint8_t *Y = (uint8_t *)a->dst + k * pb, *U = Y + g.ysz, *V = U + g.csz;
                    for (int ty = 0; ty < g.mbh; ++ty)
                        for (int tx = 0; tx < tw; ++tx) {
                            const uint8_t *lt = tiled.data() + ((size_t)ty * tw + tx) * 256, *ct = tiled.data() + g.rysz + ((size_t)ty * tw + tx) * 128;
                            const int x0 = 16 * tx - 8, c0 = 8 * tx - 4;
                            for (int r = 0; r < 16; ++r) {
                                if (tx > 0) memcpy(Y + (size_t)(16 * ty + r) * g.W + x0, lt + r * 16, 8);
                                if (tx < g.mbw) memcpy(Y + (size_t)(16 * ty + r) * g.W + x0 + 8, lt + r * 16 + 8, 8);
                            }
                            for (int r = 0; r < 8; ++r) {
                                if (tx > 0) { memcpy(U + (size_t)(8 * ty + r) * g.cw + c0, ct + r * 8, 4); memcpy(V + (size_t)(8 * ty + r) * g.cw + c0, ct + 64 + r * 8, 4); }
                                if (tx < g.mbw) { memcpy(U + (size_t)(8 * ty + r) * g.cw + c0 + 4, ct + r * 8 + 4, 4); memcpy(V + (size_t)(8 * ty + r) * g.cw + c0 + 4, ct + 64 + r * 8 + 4, 4); }
                            }
                        }
                }
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

int m2v_device_pci_bus_id(int device, char *buf, size_t cap)
{
    if (!buf || cap < 16) return M2V_E_PARAM;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) { (void)hipGetLastError(); return M2V_E_NODEVICE; }
    if (hipDeviceGetPCIBusId(buf, (int)std::min<size_t>(cap, 64), device) != hipSuccess) { (void)hipGetLastError(); return M2V_E_HIP; }
    buf[cap - 1] = 0;
    for (char *c = buf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');      // sysfs spells the address in lower case
    return (int)strlen(buf);
}

const char *m2v_last_error(const m2v_enc *e) { return e ? e->err.c_str() : t_create_err.c_str(); }

int m2v_debug_table(int which, int i, int j) { return debug_table(which, i, j); }

}  // extern "C"
