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

// fn(k) for k in [0, count), dealt to up to `threads` threads (the calling one included) when the items move 8 MB or more in all and
// at least 1 MB each thread (the planes of a run of whole frames: one core moves ~25 GB/s, the link takes twice that)
template <typename F>
void parallel_items(size_t count, size_t bytes_each, int threads, F fn)
{
    const size_t total = count * bytes_each;
    const size_t n = total < (8u << 20) ? 1 : std::min<size_t>((size_t)std::max(threads, 1), std::min(count, total >> 20));
    auto run = [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; ++k) fn(k); };
    if (n <= 1) { run(0, count); return; }
    std::vector<std::thread> pool;
    pool.reserve(n - 1);
    try {
        for (size_t t = 1; t < n; ++t) pool.emplace_back(run, count * t / n, count * (t + 1) / n);
    } catch (...) {                     // no more threads to be had: the ones that started finish, the rest is done here
        const size_t started = pool.size() + 1;
        run(0, count / n);
        run(count * started / n, count);
        for (auto &t : pool) t.join();
        return;
    }
    run(0, count / n);
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

// Packed 4:4:4 samples -> the three planes of the chunk's frames.  16 pixels per lane: STRIDE 16-byte loads (a wavefront's loads of one
// kind cover a contiguous run between them), byte selection in registers (v_perm_b32 / v_bfe), one 16-byte store per plane.
// src: frames of ysz * STRIDE bytes back to back; dst: frames of 3 * ysz bytes (Y plane, U plane, V plane).  HBM traffic 2 x the
// frame's bytes - against a link that delivers them at a hundredth of the rate.
template <int STRIDE, int OY, int OU, int OV>
__global__ __launch_bounds__(256) void k_unpack444(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t ysz, uint32_t nframes)
{
    const uint32_t per_frame = ysz >> 4;
    for (uint32_t f = blockIdx.y; f < nframes; f += gridDim.y) {
        const uint8_t *sf = src + (size_t)f * ysz * STRIDE;
        uint8_t *df = dst + (size_t)f * ysz * 3;
        for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < per_frame; t += gridDim.x * 256u) {
            const uint4 *p = (const uint4 *)(sf + (size_t)t * 16 * STRIDE);
            uint32_t w[4 * STRIDE];
#pragma unroll
            for (int k = 0; k < STRIDE; ++k) { const uint4 q = p[k]; w[4 * k] = q.x; w[4 * k + 1] = q.y; w[4 * k + 2] = q.z; w[4 * k + 3] = q.w; }
            auto byte = [&](int idx) -> uint32_t { return (w[idx >> 2] >> (8 * (idx & 3))) & 0xFFu; };
            auto plane = [&](int off) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[j] = byte((4 * j) * STRIDE + off) | (byte((4 * j + 1) * STRIDE + off) << 8) | (byte((4 * j + 2) * STRIDE + off) << 16) |
                           (byte((4 * j + 3) * STRIDE + off) << 24);
                return make_uint4(o[0], o[1], o[2], o[3]);
            };
            ((uint4 *)df)[t] = plane(OY);
            ((uint4 *)(df + ysz))[t] = plane(OU);
            ((uint4 *)(df + 2 * (size_t)ysz))[t] = plane(OV);
        }
    }
}

struct PackedFmt { int stride, oy, ou, ov; };
bool packed_fmt(int layout, PackedFmt &f)
{
    switch (layout) {
    case M2V_PACKED_YUV24: f = {3, 0, 1, 2}; return true;
    case M2V_PACKED_UYV24: f = {3, 1, 0, 2}; return true;
    case M2V_PACKED_YUVX32: f = {4, 0, 1, 2}; return true;
    case M2V_PACKED_AYUV32: f = {4, 1, 2, 3}; return true;
    }
    return false;
}

void launch_unpack(hipStream_t s, int layout, const uint8_t *src, uint8_t *dst, uint32_t ysz, uint32_t nframes)
{
    const dim3 grid(std::min<uint32_t>(((ysz >> 4) + 255u) / 256u, 1024u), std::min<uint32_t>(nframes, 32768u)), block(256);
    switch (layout) {
    case M2V_PACKED_YUV24: hipLaunchKernelGGL((k_unpack444<3, 0, 1, 2>), grid, block, 0, s, src, dst, ysz, nframes); break;
    case M2V_PACKED_UYV24: hipLaunchKernelGGL((k_unpack444<3, 1, 0, 2>), grid, block, 0, s, src, dst, ysz, nframes); break;
    case M2V_PACKED_YUVX32: hipLaunchKernelGGL((k_unpack444<4, 0, 1, 2>), grid, block, 0, s, src, dst, ysz, nframes); break;
    default: hipLaunchKernelGGL((k_unpack444<4, 1, 2, 3>), grid, block, 0, s, src, dst, ysz, nframes); break;
    }
    HIPCHK(hipGetLastError());
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
    if (h.pk.empty()) {
        if (h.uploaded < nf)
            HIPCHK(hipMemcpyAsync(h.d_in.p + h.uploaded * frame_bytes, h.h_in + h.uploaded * frame_bytes, (nf - h.uploaded) * frame_bytes,
                                  hipMemcpyHostToDevice, e->up_stream));
    } else {
        // some frames (usually all) came as packed samples: the planar ones in between go up run by run, the packed bytes that are
        // still on the host in one piece; k_unpack444 below turns them into the planes of their frames
        size_t k = h.uploaded, q = 0;
        while (k < nf) {
            while (q < h.pk.size() && h.pk[q].frame < k) ++q;
            if (q < h.pk.size() && h.pk[q].frame == k) { ++k; continue; }
            const size_t stop = q < h.pk.size() ? std::min<size_t>(h.pk[q].frame, nf) : nf;
            HIPCHK(hipMemcpyAsync(h.d_in.p + k * frame_bytes, h.h_in + k * frame_bytes, (stop - k) * frame_bytes, hipMemcpyHostToDevice, e->up_stream));
            k = stop;
        }
        if (h.pk_up < h.pk_valid)
            HIPCHK(hipMemcpyAsync(h.d_pk.p + h.pk_up, h.h_pk + h.pk_up, h.pk_valid - h.pk_up, hipMemcpyHostToDevice, e->up_stream));
    }
    h.uploaded = 0;
    // the chunk's kernels behind its uploads (on both upload streams of option direct_upload = 2: the second one's are ordered into
    // the kernel stream by m2v_push_frames itself)
    HIPCHK(hipEventRecord(h.ev_up, e->up_stream));
    HIPCHK(hipStreamWaitEvent(s, h.ev_up, 0));
    e->up_wait_ev = h.ev_up;            // (what a blocking m2v_push_frames waits for: see wait_uploads)
    if (!h.pk.empty()) {
        timer_break(e);
        for (size_t a = 0; a < h.pk.size();) {          // runs of consecutive frames of one layout: one launch each
            PackedFmt f{};
            packed_fmt(h.pk[a].layout, f);
            size_t b = a + 1;
            while (b < h.pk.size() && h.pk[b].layout == h.pk[a].layout && h.pk[b].frame == h.pk[a].frame + (b - a) &&
                   h.pk[b].off == h.pk[a].off + (b - a) * (size_t)g.ysz * (size_t)f.stride) ++b;
            launch_unpack(s, h.pk[a].layout, h.d_pk.p + h.pk[a].off, h.d_in.p + (size_t)h.pk[a].frame * frame_bytes, g.ysz, (uint32_t)(b - a));
            a = b;
        }
        h.pk.clear();
        h.pk_used = h.pk_valid = h.pk_up = 0;
    }
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
    for (auto &h : e->hs) { h.uploaded = 0; h.pk.clear(); h.pk_used = h.pk_valid = h.pk_up = 0; }
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
        if (e->cur_kind == 0) {
            uint8_t *f = e->st().h_in + e->buffered * (size_t)g.ysz * 3;
            const size_t done = e->beat_pos * 4;
            memset(f + done, 0x00, g.ysz - done);
            memset(f + g.ysz + done, 0x80, g.ysz - done);
            memset(f + 2 * (size_t)g.ysz + done, 0x80, g.ysz - done);
            e->last_frame_valid_beats = bpf;    // the fill is materialised on the host
        } else {
            e->last_frame_valid_beats = (uint32_t)e->beat_pos;      // a packed frame: the macroblock kernel fills (FrameJob::valid_beats)
        }
        e->buffered++;
        e->beat_pos = 0;
    }
    flush_buffered(e, true);
    e->state = m2v_enc::ENDED;
}

}  // namespace
}  // namespace m2v

extern "C" {

// Where the samples of a run of beats are: three arrays (stride 1; kind 0) or one interleaved array (kind 1 + M2V_PACKED_*)
struct BeatSrc { const uint8_t *y, *u, *v; int stride, kind; };
struct PushBeatsArgs { uint32_t xs, ys, pf; BeatSrc src; size_t n; int stop; };

static bool page_locked_range(const void *p, size_t bytes)
{
    // the whole range must be page-locked, not just its first byte (a pointer near the end of a registered region)
    auto one = [](const void *q) {
        hipPointerAttribute_t attr;
        const bool yes = hipPointerGetAttributes(&attr, q) == hipSuccess && attr.type == hipMemoryTypeHost;
        if (!yes) (void)hipGetLastError();          // an ordinary pointer is "invalid value" to the query: not an error here
        return yes;
    };
    return bytes && one(p) && one((const uint8_t *)p + bytes - 1);
}

// room for `bytes` more packed bytes in the stage being filled; false = the chunk has to leave first (a layout of more bytes per pixel
// than the one the buffers were sized for arrived in the middle of a chunk)
static bool pk_room(m2v_enc *e, m2v_enc::HostStage &h, size_t bytes, int stride)
{
    if (h.pk_used + bytes <= h.d_pk.n) return true;
    if (h.pk_used != 0) return false;
    h.d_pk.recorded = false;
    h.d_pk.ensure(std::max(e->batch_frames * (size_t)e->g.ysz * (size_t)stride, bytes));
    return true;
}

// the pinned staging of the packed bytes, as large as the device side (allocated with the first bytes that need it; when the device
// side has grown since, nothing of the chunk being filled is in it)
static void pk_staging(m2v_enc::HostStage &h)
{
    if (h.h_pk && h.h_pk_cap >= h.d_pk.n) return;
    if (h.h_pk) (void)hipHostFree(h.h_pk);
    h.h_pk = nullptr; h.h_pk_cap = 0;
    HIPCHK(hipHostMalloc((void **)&h.h_pk, h.d_pk.n));
    h.h_pk_cap = h.d_pk.n;
}

// `len` packed bytes of the caller at offset `off` of the stage's packed run: straight to the device from page-locked memory, else into
// the pinned staging (uploaded when the chunk leaves).  Bytes arrive in order: off is where the previous piece ended.
static void pk_put(m2v_enc *e, m2v_enc::HostStage &h, size_t off, const uint8_t *src, size_t len, bool pinned, bool &direct_pending)
{
    if (pinned) {
        if (h.pk_up < off)                      // bytes staged on the host earlier in this chunk go first
            HIPCHK(hipMemcpyAsync(h.d_pk.p + h.pk_up, h.h_pk + h.pk_up, off - h.pk_up, hipMemcpyHostToDevice, e->up_stream));
        HIPCHK(hipMemcpyAsync(h.d_pk.p + off, src, len, hipMemcpyHostToDevice, e->up_stream));
        h.pk_up = off + len;
        e->up_unsynced = true; e->up_wait_ev = nullptr;
        direct_pending = true;
    } else {
        pk_staging(h);
        parallel_copy(h.h_pk + off, src, len, e->copy_threads);
    }
    h.pk_valid = off + len;
}

static int push_beats_impl(m2v_enc *e, void *argp)
{
    auto *a = (PushBeatsArgs *)argp;
    if (e->strip_active) { e->set_err("m2v_push_*: a strip sequence is open (m2v_strip_finish or m2v_reset first)"); return M2V_E_STATE; }
    if (e->resident_inflight) { e->set_err("m2v_push_*: a resident sequence is in flight (m2v_encode_resident_end first)"); return M2V_E_STATE; }
    if (e->strip_inflight) { e->set_err("m2v_push_*: a strip sequence is in flight (m2v_strip_encode_end first)"); return M2V_E_STATE; }
    if (e->state == m2v_enc::ENDED) return M2V_OK;              // dropped while the sequence ends (RTL:1045-1058)
    size_t i = 0;
    if (a->n == 0) {
        if (a->stop && e->state == m2v_enc::DURING) do_stop(e);
        return M2V_OK;
    }
    if (e->state == m2v_enc::IDLE) start_sequence(e, a->xs, a->ys, a->pf);
    const Geom &g = e->g;
    const size_t bpf = g.ysz / 4;
    const BeatSrc &src = a->src;
    PackedFmt sf{1, 0, 0, 0};
    if (src.kind) packed_fmt(src.kind - 1, sf);
    // packed samples in page-locked memory (capture buffers) cross PCIe from where they are
    const bool pinned = src.kind != 0 && e->direct_upload && page_locked_range(src.y - sf.oy, a->n * 4 * (size_t)sf.stride);
    // ... and so do whole frames of beats on three page-locked arrays: one strided copy per plane (rows = frames)
    const bool pinned3 = src.kind == 0 && e->direct_upload && a->n >= bpf && page_locked_range(src.y, a->n * 4) && page_locked_range(src.u, a->n * 4) &&
                         page_locked_range(src.v, a->n * 4);
    bool direct_pending = false;
    while (i < a->n) {
        m2v_enc::HostStage &h = e->st();
        if (e->beat_pos == 0 && src.kind != 0) {
            // a frame begins, and it keeps the form its first beats come in: the caller's bytes as they are, de-interleaved on the
            // device.  Whole frames of the call leave in one piece.
            const size_t fbytes = (size_t)g.ysz * (size_t)sf.stride;
            const size_t whole = std::min((a->n - i) / bpf, e->batch_frames - e->buffered);
            const size_t nfr = std::max<size_t>(whole, 1);
            if (!pk_room(e, h, nfr * fbytes, sf.stride)) { flush_buffered(e, false); continue; }      // (pk_used != 0: complete frames are buffered)
            for (size_t k = 0; k < nfr; ++k) h.pk.push_back({(uint32_t)(e->buffered + k), src.kind - 1, h.pk_used + k * fbytes});
            e->cur_kind = src.kind;
            e->cur_pk_off = h.pk_used;
            h.pk_used += nfr * fbytes;
            if (whole >= 1) {
                pk_put(e, h, e->cur_pk_off, src.y - sf.oy + i * 4 * (size_t)sf.stride, whole * fbytes, pinned, direct_pending);
                i += whole * bpf;
                e->buffered += whole;
                if (e->buffered == e->batch_frames && !(a->stop && i == a->n)) flush_buffered(e, false);
                continue;
            }
        } else if (e->beat_pos == 0) {
            e->cur_kind = 0;
            const size_t whole = std::min((a->n - i) / bpf, e->batch_frames - e->buffered);
            if (pinned3 && whole >= 1) {
                const size_t fb = (size_t)g.ysz * 3;
                h.d_in.ensure(e->batch_frames * fb);
                if (h.uploaded < e->buffered)           // frames staged on the host earlier in this chunk go first (push_frames_impl does the same)
                    HIPCHK(hipMemcpyAsync(h.d_in.p + h.uploaded * fb, h.h_in + h.uploaded * fb, (e->buffered - h.uploaded) * fb, hipMemcpyHostToDevice, e->up_stream));
                const uint8_t *pl[3] = {src.y + i * 4, src.u + i * 4, src.v + i * 4};
                for (int c = 0; c < 3; ++c)
                    HIPCHK(hipMemcpy2DAsync(h.d_in.p + e->buffered * fb + (size_t)c * g.ysz, fb, pl[c], g.ysz, g.ysz, whole, hipMemcpyHostToDevice, e->up_stream));
                h.uploaded = e->buffered + whole;
                e->up_unsynced = true; e->up_wait_ev = nullptr;
                direct_pending = true;
                i += whole * bpf;
                e->buffered += whole;
                if (e->buffered == e->batch_frames && !(a->stop && i == a->n)) flush_buffered(e, false);
                continue;
            }
            if (whole >= 2) {
                // whole frames from ordinary memory: the three planes of each into its slot of the pinned staging, a few threads at work
                const size_t fb = (size_t)g.ysz * 3, ysz = g.ysz;
                const uint8_t *pl[3] = {src.y + i * 4, src.u + i * 4, src.v + i * 4};
                uint8_t *const d0 = h.h_in + e->buffered * fb;
                parallel_items(3 * whole, ysz, e->copy_threads, [=](size_t k) {
                    const size_t c = k / whole, f = k % whole;
                    memcpy(d0 + f * fb + c * ysz, pl[c] + f * ysz, ysz);
                });
                i += whole * bpf;
                e->buffered += whole;
                if (e->buffered == e->batch_frames && !(a->stop && i == a->n)) flush_buffered(e, false);
                continue;
            }
        }
        const size_t take = std::min(a->n - i, bpf - e->beat_pos);
        if (e->cur_kind == 0 && src.kind == 0) {
            uint8_t *f = h.h_in + e->buffered * (size_t)g.ysz * 3;
            memcpy(f + e->beat_pos * 4, src.y + i * 4, take * 4);    // raster order: beat b = pixels 4b..4b+3
            memcpy(f + g.ysz + e->beat_pos * 4, src.u + i * 4, take * 4);
            memcpy(f + 2 * (size_t)g.ysz + e->beat_pos * 4, src.v + i * 4, take * 4);
        } else if (e->cur_kind == src.kind) {
            pk_put(e, h, e->cur_pk_off + e->beat_pos * 4 * (size_t)sf.stride, src.y - sf.oy + i * 4 * (size_t)sf.stride, take * 4 * (size_t)sf.stride,
                   pinned, direct_pending);
        } else {
            // the frame in progress was begun in another form (a caller that mixes the entry points inside one frame): sample by
            // sample into the form the frame has
            PackedFmt df{1, 0, 0, 0};
            uint8_t *dy, *du, *dv;
            if (e->cur_kind == 0) {
                uint8_t *f = h.h_in + e->buffered * (size_t)g.ysz * 3 + e->beat_pos * 4;
                dy = f; du = f + g.ysz; dv = f + 2 * (size_t)g.ysz;
            } else {
                packed_fmt(e->cur_kind - 1, df);
                const size_t off = e->cur_pk_off + e->beat_pos * 4 * (size_t)df.stride, len = take * 4 * (size_t)df.stride;
                pk_staging(h);
                if (df.stride == 4) memset(h.h_pk + off, 0, len);
                dy = h.h_pk + off + df.oy; du = h.h_pk + off + df.ou; dv = h.h_pk + off + df.ov;
                h.pk_valid = off + len;
            }
            const uint8_t *sy = src.y + i * 4 * (size_t)sf.stride, *su = src.u + i * 4 * (size_t)sf.stride, *sv = src.v + i * 4 * (size_t)sf.stride;
            for (size_t k = 0; k < take * 4; ++k) {
                dy[k * df.stride] = sy[k * sf.stride];
                du[k * df.stride] = su[k * sf.stride];
                dv[k * df.stride] = sv[k * sf.stride];
            }
        }
        e->beat_pos += take;
        i += take;
        if (e->beat_pos == bpf) {
            e->beat_pos = 0;
            e->buffered++;
            if (e->buffered == e->batch_frames && !(a->stop && i == a->n)) flush_buffered(e, false);
        }
    }
    if (direct_pending) wait_uploads(e);            // the caller may reuse its buffer when this returns
    if (a->stop) do_stop(e);
    else progress(e, false);
    return M2V_OK;
}

int m2v_push_beats(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *y4,
                   const uint8_t *u4, const uint8_t *v4, size_t nbeats, int stop_with_last)
{
    if (!e || (nbeats && (!y4 || !u4 || !v4))) return M2V_E_PARAM;
    PushBeatsArgs a{xsize16, ysize16, pframes_count, BeatSrc{y4, u4, v4, 1, 0}, nbeats, stop_with_last};
    return guard(e, push_beats_impl, &a);
}

// Packed 4:4:4 sources (capture cards, SDI/HDMI receivers hand out interleaved samples): the same beats, the
// twelve port bytes of a beat simply arrive interleaved instead of on three arrays.  They stay interleaved until they are in HBM.
int m2v_push_packed(m2v_enc *e, uint32_t xsize16, uint32_t ysize16, uint32_t pframes_count, const uint8_t *pixels,
                    size_t nbeats, int layout, int stop_with_last)
{
    if (!e || (nbeats && !pixels)) return M2V_E_PARAM;
    PackedFmt f{};
    if (!packed_fmt(layout, f)) { e->set_err("m2v_push_packed: unknown layout %d", layout); return M2V_E_PARAM; }
    static const uint8_t none[4] = {0, 0, 0, 0};
    const uint8_t *p = pixels ? pixels : none;
    PushBeatsArgs a{xsize16, ysize16, pframes_count, BeatSrc{p + f.oy, p + f.ou, p + f.ov, f.stride, 1 + layout}, nbeats, stop_with_last};
    return guard(e, push_beats_impl, &a);
}

struct PushFramesArgs { uint32_t xs, ys, pf; const uint8_t *frames; size_t n; PullSink *sink; };

static int push_frames_impl(m2v_enc *e, void *argp)
{
    auto *a = (PushFramesArgs *)argp;
    if (e->strip_active) { e->set_err("m2v_push_*: a strip sequence is open (m2v_strip_finish or m2v_reset first)"); return M2V_E_STATE; }
    if (e->resident_inflight) { e->set_err("m2v_push_*: a resident sequence is in flight (m2v_encode_resident_end first)"); return M2V_E_STATE; }
    if (e->strip_inflight) { e->set_err("m2v_push_*: a strip sequence is in flight (m2v_strip_encode_end first)"); return M2V_E_STATE; }
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
