// m2v_port.hip — the port path of libm2v_mi355x.so: the module's pins as calls.  Beats in (RTL:25-28, 1027-1095), i_sequence_stop
// (RTL:1036-1056), o_sequence_busy (RTL:1095), 32-byte words out with o_last (RTL:2961-2994).
#include <thread>

#include "m2v_host.hpp"

namespace m2v {
namespace {

// ---------------------------------------------------------------------------------------------
// host-input path: the buffered frames go through the GPU chunk by chunk; a chunk's bytes reach the FIFO
// when its read-back completes.  Two host stages alternate so that the caller's next beats are copied
// into pinned memory while the previous chunk is uploaded, encoded and read back.
// ---------------------------------------------------------------------------------------------
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

void ensure_staging(m2v_enc *e, int idx = -1)
{
    m2v_enc::HostStage &h = idx < 0 ? e->st() : e->hs[idx];
    const size_t want = e->batch_frames * (size_t)e->g.ysz * 3;
    if (!h.h_ctl) HIPCHK(hipHostMalloc((void **)&h.h_ctl, 2 * sizeof(StreamCtl)));
    if (!h.ev_ctl) HIPCHK(hipEventCreateWithFlags(&h.ev_ctl, hipEventDisableTiming));
    if (!h.ev_out) HIPCHK(hipEventCreateWithFlags(&h.ev_out, hipEventDisableTiming));
    if (!h.ev_up) HIPCHK(hipEventCreateWithFlags(&h.ev_up, hipEventDisableTiming | hipEventDisableSystemFence));      // (nothing to release to the host behind an upload)
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

// 16 bytes per lane: a chunk's stream bytes into the pinned read-back buffer.  By a kernel, not by the copy engine: a read-back the copy
// engine is working on when the NEXT upload arrives delays that upload by most of its own duration (+45 us per 3.4 MB on this box, every
// second m2v_push_frames call; nothing when a kernel writes the bytes - profiles/r05_experiments.txt item 13)
__global__ __launch_bounds__(256) void k_readback(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// The blocking call's wait for its direct uploads.  An event behind the last transfer (flush_buffered records one for the chunk's kernels)
// is waited for ITSELF: the wait for the upload stream behind such an event takes ~30 us longer than the transfer, the wait for the event
// ~15 (tools/ubench/h2d_kernel.hip; the event carries no system fence - an upload leaves nothing to release to the host).  Without an event
// behind it the stream's last command is the transfer, and the wait for the stream costs nothing extra.
// (Round 5 also had a GATE here - the kernels queued behind a one-lane kernel polling a host word, released by the call after its wait for
// the bare transfer: 10 us better still, and unusable: a spinning kernel holds up its hardware queue, and in a process with several threads
// another thread's hipFree waits for the device inside the runtime while this thread needs the runtime to get to its release - ten-second
// stalls in tools/stress_threads.py.  profiles/r05_experiments.txt items 13, 16, 17.)
static void wait_uploads(m2v_enc *e)
{
    if (!e->up_unsynced) return;
    if (e->up_wait_ev) HIPCHK(hipEventSynchronize(e->up_wait_ev));
    else HIPCHK(hipStreamSynchronize(e->up_stream));
    e->up_unsynced = false;
}

// Move submitted chunks forward, oldest first.  block = wait for every step; until >= 0 = return as soon
// as that stage is free again.
// m2v_pull's destination while it moves chunks forward: a completed chunk's whole 32-byte words go straight from the pinned read-back
// buffer to the caller when nothing is waiting in the FIFO in front of them (one copy instead of two); the residue takes the FIFO
struct PullSink { uint8_t *dst; size_t cap, used; };

void progress(m2v_enc *e, bool block, int until = -1, PullSink *sink = nullptr)
{
    if (!sink) sink = (PullSink *)e->call_sink;
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
            if (h.bytes) {
                hipLaunchKernelGGL(k_readback, dim3(32), dim3(256), 0, e->copy_stream, (uint4 *)h.h_out, (const uint4 *)h.d_out.p, (h.bytes + 15) / 16);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(h.ev_out, e->copy_stream));
            h.stage = 2;
        }
        if (!event_reached(h.ev_out, block)) return;
        size_t direct = 0;
        const size_t waiting = e->fifo.size() - e->fifo_rd;            // less than a word: the residue of the chunk before
        if (sink && waiting < 32 && sink->cap - sink->used >= 32 && waiting + h.bytes >= 32) {
            // the residue first, then this chunk's bytes straight from the read-back buffer - whole words in all
            const size_t whole = std::min(waiting + h.bytes, sink->cap - sink->used) & ~(size_t)31;
            memcpy(sink->dst + sink->used, e->fifo.data() + e->fifo_rd, waiting);
            direct = whole - waiting;
            memcpy(sink->dst + sink->used + waiting, h.h_out, direct);
            sink->used += whole;
            e->fifo.clear();
            e->fifo_rd = 0;
        }
        e->fifo.insert(e->fifo.end(), h.h_out + direct, h.h_out + h.bytes);
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
    const bool staged = h.uploaded < nf;
    if (staged)
        HIPCHK(hipMemcpyAsync(h.d_in.p + h.uploaded * frame_bytes, h.h_in + h.uploaded * frame_bytes, (nf - h.uploaded) * frame_bytes,
                              hipMemcpyHostToDevice, e->up_stream));
    h.uploaded = 0;
    // the chunk's kernels behind its uploads (on both upload streams of option direct_upload = 2: the second one's are ordered into
    // the kernel stream by m2v_push_frames itself)
    HIPCHK(hipEventRecord(h.ev_up, e->up_stream));
    HIPCHK(hipStreamWaitEvent(s, h.ev_up, 0));
    e->up_wait_ev = h.ev_up;            // (what a blocking m2v_push_frames waits for: see wait_uploads)
    // worst case ~1.2 KB per macroblock; typical streams are ~100x smaller
    const size_t cap = nf * ((size_t)g.mbs * 1216 + (size_t)g.mbh * 8 + 64) + 256;
    h.d_out.ensure(cap);
    e->d_ctl.ensure(1);
    ctl_begin(e, (unsigned long long)cap, e->first_chunk);
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
    e->up_wait_ev = nullptr;
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

}  // namespace
}  // namespace m2v

extern "C" {

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

struct PushFramesArgs { uint32_t xs, ys, pf; const uint8_t *frames; size_t n; PullSink *sink; };

static int push_frames_impl(m2v_enc *e, void *argp)
{
    auto *a = (PushFramesArgs *)argp;
    if (e->strip_active) { e->set_err("m2v_push_*: a strip sequence is open (m2v_strip_finish or m2v_reset first)"); return M2V_E_STATE; }
    if (e->resident_inflight) { e->set_err("m2v_push_*: a resident sequence is in flight (m2v_encode_resident_end first)"); return M2V_E_STATE; }
    // option direct_upload = 2 promises that a pushed range is free once the NEXT m2v_push_frames has returned - whatever that next
    // call turns out to be: frames dropped while the sequence ends, no frames at all, or frames from ordinary memory (the staging path)
    auto settle_deferred = [&] {
        for (int k = 0; k < 2; ++k)
            if (e->upl_pending[k]) { HIPCHK(hipEventSynchronize(e->ev_upl[k])); e->upl_pending[k] = false; }
    };
    if (e->state == m2v_enc::ENDED || a->n == 0) { settle_deferred(); return M2V_OK; }
    if (e->state == m2v_enc::IDLE) start_sequence(e, a->xs, a->ys, a->pf);
    const Geom &g = e->g;
    const size_t fb = (size_t)g.ysz * 3;
    if (e->beat_pos != 0) {
        settle_deferred();
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
            // (deferred completion: the calls' transfers alternate between two upload streams - the copy engine sets the next one up while
            // the running one drains, which one in-order stream does not allow; measured: one stream 47.6 GB/s, no better than blocking)
            hipStream_t ups = e->up_stream;
            if (e->direct_upload_deferred && e->up_parity) {
                // (which copy engine a stream's transfers run on is the runtime's choice: the two streams overlap each other's set-up when
                // they are dealt different engines - 53 GB/s from one caller in a fresh process, 93 % of the plain copy - and behave like
                // one stream when they are not - 49 GB/s, seen in bench.py's process with its dozen streams; a stream of another
                // PRIORITY, which never shares a hardware queue, did not change that)
                if (!e->up_stream2) HIPCHK(hipStreamCreateWithFlags(&e->up_stream2, hipStreamNonBlocking));
                ups = e->up_stream2;
            }
            HIPCHK(hipMemcpyAsync(h.d_in.p + e->buffered * fb, a->frames + k * fb, take * fb, hipMemcpyHostToDevice, ups));
            if (ups != e->up_stream) {
                // the chunk's kernels (enqueued on the handle's stream by flush_buffered, possibly a few lines below) come behind these frames
                if (!e->ev_up2) HIPCHK(hipEventCreateWithFlags(&e->ev_up2, hipEventDisableTiming));
                HIPCHK(hipEventRecord(e->ev_up2, ups));
                HIPCHK(hipStreamWaitEvent(e->stream, e->ev_up2, 0));
            }
            h.uploaded = e->buffered + take;
            direct_pending = true;
            if (!e->direct_upload_deferred && ups == e->up_stream) { e->up_unsynced = true; e->up_wait_ev = nullptr; }     // (no event behind this transfer yet)
            // m2v_push_frames_pull: completed chunks leave for the caller's buffer HERE, while this call's frames cross the link
            if (a->sink) progress(e, false, -1, a->sink);
        } else {
            parallel_copy(h.h_in + e->buffered * fb, a->frames + k * fb, take * fb, e->copy_threads);
        }
        e->buffered += take;
        k += take;
        if (e->buffered == e->batch_frames) flush_buffered(e, false);
    }
    if (a->sink && !direct_pending) progress(e, false, -1, a->sink);
    if (direct_pending) {
        if (e->direct_upload_deferred) {
            // option direct_upload = 2: this call's frames are still being read when it returns; what is waited for here is the
            // PREVIOUS call's upload (its frames are free from now on).  The copy engine then always has the next transfer queued
            // behind the one it is working on: no idle link between two pushes of one caller.
            const int k = e->up_parity;
            if (!e->ev_upl[k]) HIPCHK(hipEventCreateWithFlags(&e->ev_upl[k], hipEventDisableTiming));
            HIPCHK(hipEventRecord(e->ev_upl[k], k && e->up_stream2 ? e->up_stream2 : e->up_stream));
            if (e->upl_pending[k ^ 1]) { HIPCHK(hipEventSynchronize(e->ev_upl[k ^ 1])); e->upl_pending[k ^ 1] = false; }
            e->upl_pending[k] = true;
            e->up_parity ^= 1;
        } else {
            wait_uploads(e);                                // the caller may reuse its buffer when this returns
            settle_deferred();                              // (the option was switched off between two calls)
        }
    } else {
        settle_deferred();                                  // a call through the staging copy: the previous call's frames are free all the same
    }
    progress(e, false, -1, a->sink);
    return M2V_OK;
}

int m2v_push_frames(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *frames444,
                    size_t nframes)
{
    if (!e || (nframes && !frames444)) return M2V_E_PARAM;
    PushFramesArgs a{xsize16, ysize16, pframes_count, frames444, nframes, nullptr};
    return guard(e, push_frames_impl, &a);
}

static long long pull_tail(m2v_enc *e, uint8_t *dst, size_t cap, const PullSink &sink, int *last);
static int pull_progress_impl(m2v_enc *e, void *argp);

long long m2v_push_frames_pull(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *frames444,
                               size_t nframes, uint8_t *dst, size_t cap, int *last)
{
    if (!e || (nframes && !frames444) || (!dst && cap)) return M2V_E_PARAM;
    if (last) *last = 0;
    PullSink sink{dst, cap, 0};
    PushFramesArgs a{xsize16, ysize16, pframes_count, frames444, nframes, &sink};
    e->call_sink = &sink;
    int r = guard(e, push_frames_impl, &a);
    e->call_sink = nullptr;
    if (r < 0) {
        // chunks that completed inside the failed call went to dst already: back to the front of the FIFO with them, a later m2v_pull
        // hands them out again (after an error only m2v_reset and m2v_pull are of any use, but nothing that was encoded is lost)
        if (sink.used) {
            try { e->fifo.insert(e->fifo.begin() + (long)e->fifo_rd, dst, dst + sink.used); } catch (...) { return M2V_E_NOMEM; }
        }
        return r;
    }
    if (e->state == m2v_enc::ENDED && !e->pending.empty()) {        // (frames are dropped while the sequence ends; the pull half waits like m2v_pull)
        r = guard(e, pull_progress_impl, &sink);
        if (r < 0) {
            if (sink.used) {
                try { e->fifo.insert(e->fifo.begin() + (long)e->fifo_rd, dst, dst + sink.used); } catch (...) { return M2V_E_NOMEM; }
            }
            return r;
        }
    }
    return pull_tail(e, dst, cap, sink, last);
}

static int upload_wait_impl(m2v_enc *e, void *)
{
    if (e->up_stream && (e->upl_pending[0] || e->upl_pending[1])) {
        HIPCHK(hipStreamSynchronize(e->up_stream));
        if (e->up_stream2) HIPCHK(hipStreamSynchronize(e->up_stream2));
    }
    e->upl_pending[0] = e->upl_pending[1] = false;
    return M2V_OK;
}

int m2v_upload_wait(m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    return guard(e, upload_wait_impl, nullptr);
}

static int stop_impl(m2v_enc *e, void *)
{
    const int r = upload_wait_impl(e, nullptr);         // every frame handed in has been read: the caller's buffers are free when this returns
    if (r < 0) return r;
    if (e->state == m2v_enc::DURING) do_stop(e);       // no effect while idle / already ending (RTL:1090)
    return M2V_OK;
}

int m2v_sequence_stop(m2v_enc *e)
{
    if (!e) return M2V_E_PARAM;
    return guard(e, stop_impl, nullptr);
}

int m2v_busy(const m2v_enc *e) { return e && e->state != m2v_enc::IDLE; }

static int pull_progress_impl(m2v_enc *e, void *argp)
{
    // chunks still in flight: take what is complete; once the sequence has been stopped wait for the rest
    progress(e, e->state == m2v_enc::ENDED, -1, (PullSink *)argp);
    return M2V_OK;
}

long long m2v_pull(m2v_enc *e, uint8_t *dst, size_t cap, int *last)
{
    if (!e || (!dst && cap)) return M2V_E_PARAM;
    if (last) *last = 0;
    PullSink sink{dst, cap, 0};
    if (!e->pending.empty()) {
        const int r = guard(e, pull_progress_impl, &sink);
        if (r < 0) return r;
    }
    return pull_tail(e, dst, cap, sink, last);
}

// what m2v_pull and m2v_push_frames_pull end with: the FIFO's whole words behind what went to the caller directly, the end of the sequence
static long long pull_tail(m2v_enc *e, uint8_t *dst, size_t cap, const PullSink &sink, int *last)
{
    const size_t avail = e->fifo.size() - e->fifo_rd;
    // only whole 32-byte words leave; the residue waits for more data or for the end of the sequence
    size_t n = std::min(avail, cap - sink.used) & ~(size_t)31;
    if (n) memcpy(dst + sink.used, e->fifo.data() + e->fifo_rd, n);
    e->fifo_rd += n;
    n += sink.used;
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

}  // extern "C"
