// m2v_resident.hip — whole sequences with input and output resident in HBM (what bench.py times): m2v_encode_resident, its two
// halves _begin / _end for callers that keep several sequences in flight.
#include "m2v_host.hpp"

extern "C" {

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
    if (e->state != m2v_enc::IDLE || e->strip_active || e->resident_inflight || e->strip_inflight) { e->set_err("m2v_encode_resident: encoder busy"); return M2V_E_STATE; }
    e->resident_empty = false;
    if (a->n == 0) {                                    // no beat: the sequence never starts
        if (a->bytes) *a->bytes = 0;
        e->resident_empty = a->async;                   // only _begin leaves an _end to answer
        return M2V_OK;
    }
    hipStream_t s = a->s ? a->s : e->stream;
    e->g = make_geom(e, a->xs, a->ys);
    e->pframes = a->pf & 0xFFu;
    e->frames_total = 0;
    e->persist_slot = -1;
    for (auto &st : e->stats) st = KStat{};
    const Geom &g = e->g;
    const size_t fb = (size_t)g.ysz * 3;
    // the control word is set up by the first chunk's k_frame_scan (no launch, no copy in front of the first kernel)
    if (!e->st().h_ctl) HIPCHK(hipHostMalloc((void **)&e->st().h_ctl, 2 * sizeof(StreamCtl)));
    ctl_begin(e, (unsigned long long)a->cap, true);
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

}  // extern "C"
