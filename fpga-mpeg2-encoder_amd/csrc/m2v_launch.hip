// m2v_launch.hip — the one translation unit of libm2v_mi355x.so that contains device code: it includes m2v_kernels.hpp, uploads the
// constant tables into this code object's device globals and offers one plain C++ launch function per kernel to the host units
// (m2v_host.hpp).  Kernel template arguments are chosen here from the handle's parameters (VECTOR_LEVEL, options).
#include <mutex>

#include "m2v_host.hpp"
#include "m2v_kernels.hpp"

namespace m2v {

namespace {

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
    // per-lane operands of the DCT-as-GEMM variant (k_mb<.., MFMA = true>, see MfmaLane)
    MfmaLane ml[64];
    SearchLane sl[64];
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, c = lane & 15;
        MfmaLane &m = ml[lane];
        memset(&m, 0, sizeof m);
        if ((c >> 3) == (g & 1))
            for (int b = 0; b < 8; ++b) {
                const int8_t w = (int8_t)(g < 2 ? kDctBasis[(c & 7) * 8 + b] : -kDctBasis[(c & 7) * 8 + b]);
                m.b1[b >> 2] |= (uint32_t)(uint8_t)w << (8 * (b & 3));
            }
        if ((c >> 3) == (g >> 1))
            for (int b = 0; b < 4; ++b) m.a2 |= (uint32_t)(uint8_t)kDctBasis[(c & 7) * 8 + 4 * (g & 1) + b] << (8 * b);
        const int tile = ((c >> 3) << 1) | (g >> 1);            // the second pass leaves Y transposed: lane (g, c) = row c, columns 4g .. 4g+3
        m.a1c = (uint32_t)(kMfmaCpOff + ((4 + (c >> 3)) * 8 + (c & 7)) * 16 + 8 * (g >> 1));        // &s_cp[4 + (c >> 3)][c & 7][8 (g >> 1)]
        uint32_t wq4 = 0;
        for (int v = 0; v < 4; ++v) {
            const int raster = (c & 7) * 8 + 4 * (g & 1) + v;
            m.zoff[v] = (uint32_t)(slot_of_tile(tile) * 128 + kZigzagPos[raster] * 2);
            wq4 |= (uint32_t)kIntraW[raster] << (8 * v);
            m.irecip[v] = recip[raster];
        }
        // the reference pairs (w0,w1) (w2,w3) start at dword gq of window row dy', the pairs (w1,w2) (w3,w4) at gq + 1: one of the
        // two starts is even in copy A, the other in copy B (which holds dword j + 1 at index j); VECTOR_LEVEL 3 geometry
        const int dyi = s3_dy(lane), gq = s3_group(lane), gap = win_b_gap(16 + 4 * 3);
        SearchLane &q = sl[lane];
        memset(&q, 0, sizeof q);
        // candidate j of the lane has dx = 4 gq - 8 + j
        const uint32_t cbase = 255u - (uint32_t)((dyi << 4) | (4 * gq));
        q.cb4 = cbase | ((cbase - 1u) << 8) | ((cbase - 2u) << 16) | ((cbase - 3u) << 24);
        q.iwq4 = wq4; q.iw = kIntraW[lane]; q.iwrecip = recip[lane];            // (the intra quantiser's words that share this quad)
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
            // basis row j as signed bytes and its negative, 16 bytes per row: ONE vector load per lane (a vector memory instruction costs
            // this kernel as much as ten arithmetic ones, profiles/r04_experiments.txt item 14)
            int8_t dct_pn[8][16];
            for (int j = 0; j < 8; ++j) { memcpy(dct_pn[j], &kDctBasis[j * 8], 8); memcpy(dct_pn[j] + 8, &dct_neg[j * 8], 8); }
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), dct_pn, sizeof dct_pn, cst + kConstDct));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), kCbpCode, sizeof kCbpCode, cst + kConstCbp));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), dcl, sizeof dcl, cst + kConstDcLuma));
            uint32_t hpd[16][kHpDeadStride / 4];
            for (int F = 0; F < 16; ++F)
                for (int pr = 0; pr < kHpDeadStride / 4; ++pr) hpd[F][pr] = hp_dead_word(pr, F);
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), hpd, sizeof hpd, cst + kConstHpDead));
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(d_lanetab), ac2, ac2_bytes, blk + (size_t)kQuadAc0 * 64 * 16));
        }
    HIPCHK(hipDeviceSynchronize());         // the copies read stack arrays: complete before they go out of scope
    // the lane tables d_lanek[VL - 1][P]: every instantiation of the macroblock kernel writes its own (LaneK, FILL = true); they
    // read the constant tables uploaded above
    {
        Geom g0{};
        const dim3 one(1), wave(64);
#define M2V_FILL(VLV, PV) hipLaunchKernelGGL((k_mb<VLV, PV, false, false, true>), one, wave, 0, 0, (const FrameJob *)nullptr, (const MbMap *)nullptr, g0, \
                                             (uint32_t *)nullptr, (MbAux *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int16_t *)nullptr)
        M2V_FILL(1, false); M2V_FILL(1, true); M2V_FILL(2, false); M2V_FILL(2, true); M2V_FILL(3, false); M2V_FILL(3, true);
#undef M2V_FILL
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
}

}  // namespace

void upload_tables(int device)
{
    if (device >= 0 && device < kMaxDevices) std::call_once(g_tables_once[device], upload_tables_now);
    else upload_tables_now();
}

// Block -> macroblock table of a k_mb launch (MbMap, m2v_types.hpp), filled once per launch shape by a small kernel on the launch's own
// stream (no host copy: nothing here waits for anything, and another rank's waiting kernels cannot hold it up) and kept with the handle.
// mode 0: the rows [row0, row1) through xcd_remap (option cu_pack); 1: the strip's two edge rows only (row0 and row0 + rstride; g.row1 -
// g.row0 = 1 or 2 local rows); 2: the peer form - the edge rows first (the launch's first n_edge blocks), then the rows in between.
__host__ __device__ inline MbMap mbmap_entry(uint32_t b, uint32_t n, const Geom &g, int mode, uint32_t n_edge)
{
    int by, bx, edge = mode == 1;
    if (mode == 2 && b < n_edge) edge = 1;
    if (edge) {
        const uint32_t local = xcd_remap(b, mode == 1 ? n : n_edge, 0u), lrow = local / (uint32_t)g.mbw;
        bx = (int)(local - lrow * (uint32_t)g.mbw);
        by = g.row0 + (int)lrow * g.rstride;
    } else {
        const uint32_t local = mode == 2 ? xcd_remap(b - n_edge, n - n_edge, (uint32_t)g.cu_pack) : xcd_remap(b, n, (uint32_t)g.cu_pack);
        const int mb = (g.row0 + (mode == 2 ? 1 : 0)) * g.mbw + (int)local;      // (the peer form's other blocks: the rows behind the strip's first)
        by = mb / g.mbw;
        bx = mb - by * g.mbw;
    }
    MbMap m;
    m.mb = (uint32_t)(by * g.mbw + bx) | (bx > 0 ? 1u << 24 : 0u) | (bx + 1 < g.mbw ? 1u << 25 : 0u) | (by > 0 ? 1u << 26 : 0u) |
           (by + 1 < g.mbh ? 1u << 27 : 0u) | (edge ? 1u << 28 : 0u);
    m.byx = ((uint32_t)by << 16) | (uint32_t)bx;
    return m;
}

__global__ void k_mbmap_fill(MbMap *out, uint32_t n, Geom g, int mode, uint32_t n_edge)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n) out[b] = mbmap_entry(b, n, g, mode, n_edge);
}

static const MbMap *mbmap_for(m2v_enc *e, hipStream_t s, const Geom &g, int mode, int n_edge = 0)
{
    const m2v_enc::MbMapKey key{g.row0, g.row1, g.mbw, g.mbh, mode == 1 ? 0 : g.cu_pack, mode, g.rstride, n_edge};
    for (auto &c : e->mbmaps)
        if (!memcmp(&c.key, &key, sizeof key)) {
            // a table filled on ANOTHER stream of the handle (the GOP groups run on a stream each): this stream waits for the fill once
            if (c.filled_on != s && std::find(c.waited.begin(), c.waited.end(), s) == c.waited.end()) {
                HIPCHK(hipStreamWaitEvent(s, c.ev, 0));
                c.waited.push_back(s);
            }
            return c.d.p;
        }
    const uint32_t n = (uint32_t)((g.row1 - g.row0) * g.mbw);
    if (e->mbmaps.size() >= 24) {                           // (a recorded graph that still points at it is invalidated by the release: alloc_generation)                           // (a recorded graph that still points at it is invalidated by the release: alloc_generation)
        (void)hipDeviceSynchronize();                       // launches on ANY of the handle's streams may still be reading the oldest table (rare: 24 shapes)
        if (e->mbmaps.front().ev) (void)hipEventDestroy(e->mbmaps.front().ev);
        e->mbmaps.front().d.release();
        e->mbmaps.pop_front();
    }
    e->mbmaps.emplace_back();
    auto &c = e->mbmaps.back();
    c.key = key;
    c.d.ensure(n);
    Geom gg = g;
    if (mode == 1) gg.cu_pack = 0;
    hipLaunchKernelGGL(k_mbmap_fill, dim3((n + 255) / 256), dim3(256), 0, s, c.d.p, n, gg, mode, (uint32_t)n_edge);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c.ev, s));
    c.filled_on = s;
    timer_break(e);
    return c.d.p;
}

// strip mode, m2v_strip_encode: the strip's first and last macroblock row in ONE launch of the EDGE instantiation, which also
// writes their outer rows of the reconstruction into the halo buffers (no pack kernel).  gg: row0 = first row, rstride = distance
// to the last one, row1 - row0 = 1 or 2 local rows.
template <bool P>
void launch_mb_edges(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g, uint8_t *up, uint8_t *down,
                     const uint8_t *nb_up, const uint8_t *nb_down)
{
    if (count <= 0) return;
    const dim3 grid((unsigned)((g.row1 - g.row0) * g.mbw), (unsigned)count), block(64);
    const MbMap *const mm = mbmap_for(e, s, g, 1);
    Timer t(e, s, P ? 0 : 1, (double)count * (g.row1 - g.row0) * g.mbw * 256.0);
    int16_t *dbg = e->keep_recon ? e->d_coef.p : nullptr;
    const FrameJob *const jl = e->d_joblist.p + (d_list - e->d_lists.p);
#define M2V_LAUNCH_EDGE(VLV) \
    hipLaunchKernelGGL((k_mb<VLV, P, false, true, false, true>), grid, block, 0, s, jl, mm, g, e->d_mbinfo.p, e->d_mbaux.p, \
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

// strip mode, peer transport: ALL rows of the strip in one launch of the EDGE + PEER instantiation - the first ps.n_edge blocks are the
// strip's first and last macroblock row, which store their outer rows into the neighbours' landing buffers (put_up / put_down) and
// read the neighbours' rows of the previous step from this rank's own (got_up / got_down); see PeerStep.  gg: row0 / row1 = the strip,
// rstride = distance from its first to its last row.
template <bool P>
void launch_mb_peer(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g, uint8_t *put_up, uint8_t *put_down,
                    const uint8_t *got_up, const uint8_t *got_down, const PeerStep &ps)
{
    if (count <= 0) return;
    const dim3 grid((unsigned)((g.row1 - g.row0) * g.mbw), (unsigned)count), block(64);
    const MbMap *const mm = mbmap_for(e, s, g, 2, (int)ps.n_edge);
    Timer t(e, s, P ? 0 : 1, (double)count * (g.row1 - g.row0) * g.mbw * 256.0);
    const FrameJob *const jl = e->d_joblist.p + (d_list - e->d_lists.p);
#define M2V_LAUNCH_PEER(VLV) \
    hipLaunchKernelGGL((k_mb<VLV, P, false, true, false, true, true>), grid, block, 0, s, jl, mm, g, e->d_mbinfo.p, e->d_mbaux.p, \
                       e->d_slots_small.p, e->d_slots.p, (int16_t *)nullptr, put_up, put_down, got_up, got_down, ps)
    switch (e->VL) {
        case 1: M2V_LAUNCH_PEER(1); break;
        case 2: M2V_LAUNCH_PEER(2); break;
        default: M2V_LAUNCH_PEER(3); break;
    }
#undef M2V_LAUNCH_PEER
    HIPCHK(hipGetLastError());
    t.stop();
}

template <bool P>
void launch_mb(m2v_enc *e, hipStream_t s, const int *d_list, int count, const Geom &g)
{
    if (count <= 0) return;
    const dim3 grid((unsigned)((g.row1 - g.row0) * g.mbw), (unsigned)count), block(64);      // one wavefront per macroblock; y = frame of the launch list
    const MbMap *const mm = mbmap_for(e, s, g, 0);
    Timer t(e, s, P ? 0 : 1, (double)count * g.ysz);
    int16_t *dbg = e->keep_recon ? e->d_coef.p : nullptr;
    const FrameJob *const jl = e->d_joblist.p + (d_list - e->d_lists.p);      // the same launch list, as jobs
#define M2V_LAUNCH_MB(VLV, PV, CV) \
    do { \
        if (e->dct_mfma && !(CV)) \
            hipLaunchKernelGGL((k_mb<VLV, PV, false, true>), grid, block, 0, s, jl, mm, g, e->d_mbinfo.p, e->d_mbaux.p, \
                               e->d_slots_small.p, e->d_slots.p, dbg); \
        else \
            hipLaunchKernelGGL((k_mb<VLV, PV, CV, false>), grid, block, 0, s, jl, mm, g, e->d_mbinfo.p, e->d_mbaux.p, \
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

template void launch_mb<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &);
template void launch_mb<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &);
template void launch_mb_edges<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *);
template void launch_mb_edges<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *);
template void launch_mb_peer<false>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *, const PeerStep &);
template void launch_mb_peer<true>(m2v_enc *, hipStream_t, const int *, int, const Geom &, uint8_t *, uint8_t *, const uint8_t *, const uint8_t *, const PeerStep &);


// the chunk's plan from pinned host memory (read by the kernel itself, over PCIe) into the device arrays the kernels index
__global__ __launch_bounds__(256) void k_plan_upload(uint32_t *__restrict__ d_jobs, const uint32_t *__restrict__ h_jobs, uint32_t n_jobs,
                                                     uint32_t *__restrict__ d_lists, const uint32_t *__restrict__ h_lists, uint32_t n_lists,
                                                     uint32_t *__restrict__ d_joblist, const uint32_t *__restrict__ h_joblist, uint32_t n_joblist)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint32_t k = i; k < n_jobs; k += stride) d_jobs[k] = h_jobs[k];
    for (uint32_t k = i; k < n_lists; k += stride) d_lists[k] = h_lists[k];
    for (uint32_t k = i; k < n_joblist; k += stride) d_joblist[k] = h_joblist[k];
}

void launch_plan_upload(m2v_enc *e, hipStream_t s, const FrameJob *h_jobs, size_t nf, const int *h_lists, const FrameJob *h_joblist, size_t nlist)
{
    static_assert(sizeof(FrameJob) % 4 == 0, "copied as dwords");
    const uint32_t n1 = (uint32_t)(nf * sizeof(FrameJob) / 4), n2 = (uint32_t)nlist, n3 = (uint32_t)(nlist * sizeof(FrameJob) / 4);
    const uint32_t blocks = std::min<uint32_t>(64u, (std::max(n1, n3) + 255u) / 256u);
    hipLaunchKernelGGL(k_plan_upload, dim3(std::max(1u, blocks)), dim3(256), 0, s, (uint32_t *)e->d_jobs.p, (const uint32_t *)h_jobs, n1,
                       (uint32_t *)e->d_lists.p, (const uint32_t *)h_lists, n2, (uint32_t *)e->d_joblist.p, (const uint32_t *)h_joblist, n3);
    HIPCHK(hipGetLastError());
}

// The control word of a chunk's stream starts inside k_frame_scan (ctl_init): what the next launch_frame_scan on this handle tells it.
void ctl_begin(m2v_enc *e, unsigned long long cap, bool first)
{
    e->d_ctl.ensure(1);
    e->ctl_init = first ? 1 : 2;
    e->ctl_cap = cap;
}

void launch_slice_scan(m2v_enc *e, hipStream_t s, const Geom &g, int f0, int f1)
{
    if (f1 <= f0) return;
    const size_t rows = (size_t)(g.row1 - g.row0);
    hipLaunchKernelGGL(k_slice_scan, dim3((unsigned)((size_t)(f1 - f0) * rows)), dim3(128), 0, s, e->d_jobs.p, g, e->d_mbinfo.p,
                       e->d_mbaux.p, e->d_mblen.p, e->d_slice_bytes.p, f0);
}

// offsets of every frame and slice, stream length, and the tail (end code + padding) cleared
void launch_frame_scan(m2v_enc *e, hipStream_t s, const Geom &g, size_t nf, bool first, bool last, bool advance, uint8_t *d_stream)
{
    PeerScan px{};
    if (e->scan_peer_gaveup) {
        px.gaveup = e->scan_peer_gaveup; px.clear = e->scan_peer_clear; px.clear_lines = e->scan_peer_lines; px.mark = e->scan_peer_mark;
        e->scan_peer_gaveup = nullptr;
    }
    hipLaunchKernelGGL(k_frame_scan, dim3(1), dim3(1024), 0, s, e->d_jobs.p, g, (int)nf, first ? 1 : 0, last ? 1 : 0,
                       e->d_slice_bytes.p, e->d_slice_off.p, e->d_frame_off.p, e->d_ctl.p, advance ? 1 : 0, (uint32_t *)d_stream,
                       e->ctl_init, e->ctl_cap, px);
    e->ctl_init = 0;
}

// slices, and with them the headers and the sequence end code
void launch_assemble(m2v_enc *e, hipStream_t s, const Geom &g, size_t nf, bool first, bool last, uint8_t *d_stream)
{
    const size_t rows = (size_t)(g.row1 - g.row0);
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)(nf * rows)), dim3(kAsmThreads), 0, s, e->d_jobs.p, g, (int)nf,
                       e->d_mbinfo.p, e->d_mbaux.p, e->d_slots_small.p, e->d_slots.p, e->d_slice_off.p,
                       (uint32_t *)d_stream, e->d_ctl.p, first ? 1 : 0, last ? 1 : 0, e->d_frame_off.p, e->d_slice_bytes.p);
}

void launch_halo_pack(m2v_enc *e, hipStream_t s, const int *d_list, int count, uint8_t *up, uint8_t *down)
{
    hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)count, 2), dim3(256), 0, s, e->d_jobs.p, d_list, e->g, 2 * e->VL, e->VL, up, down);
}

void launch_halo_unpack(m2v_enc *e, hipStream_t s, const int *d_list, int count, const uint8_t *from_up, const uint8_t *from_down)
{
    hipLaunchKernelGGL(k_halo_unpack, dim3((unsigned)count, 2), dim3(256), 0, s, e->d_jobs.p, d_list, e->g, 2 * e->VL, e->VL, from_up, from_down);
}

void launch_strip_assemble(m2v_enc *e, hipStream_t s, const Geom &g, uint32_t gop, size_t nf, int nranks, const StripSrc &src,
                           const unsigned long long *d_all_off, uint8_t *d_out, unsigned long long cap)
{
    const size_t nsegs = nf * (size_t)nranks;
    hipLaunchKernelGGL(k_strip_layout, dim3(1), dim3(kLayoutThreads), 0, s, d_all_off, nranks, (int)nf, gop, src, (CopySeg *)e->d_segs.p,
                       e->d_frame_pos.p, e->d_ctl.p, cap);
    // every segment cut into `split` parts so that the launch has one to two thousand blocks whatever the number of ranks
    const int split = (int)std::max<size_t>(1, std::min<size_t>(32, 2048 / std::max<size_t>(nsegs, 1)));
    const unsigned blocks = (unsigned)(nsegs * (size_t)split + (nf + kCopyThreads - 1) / kCopyThreads + 1);
    hipLaunchKernelGGL(k_strip_assemble, dim3(blocks), dim3(kCopyThreads), 0, s, (const CopySeg *)e->d_segs.p, (int)nsegs, split, g, (int)nf, gop,
                       e->d_frame_pos.p, d_out, e->d_ctl.p);
}

/* table accessors (no GPU needed): tests/test_abi.py checks the product's tables against the oracle's */
int debug_table(int which, int i, int j)
{
    switch (which) {
        case 0: return kDctBasis[(i & 7) * 8 + (j & 7)];
        case 1: return kIntraW[(i & 7) * 8 + (j & 7)];
        case 2: return kZigzagPos[(i & 7) * 8 + (j & 7)];
        case 3: return i >= 0 && i < 17 ? kMotionCode[i] : -1;
        case 4: return i >= 0 && i < 64 ? kCbpCode[i] : -1;
        case 5: return i >= 0 && i < 2 && j >= 0 && j < 12 ? (kDcSizeLen[i][j] << 16) | kDcSizeCode[i][j] : -1;
        case 6: return i >= 0 && i < 32 && j >= 1 && j <= 40 ? kAcCode[i * 40 + j - 1] : 0;
        default:
            // 16 + cu_pack: where k_mb sends block i of a launch of j blocks (the XCD / CU permutation of xcd_remap)
            if (which >= 16 && which <= 16 + 8 && i >= 0 && j > 0 && i < j) return (int)xcd_remap((uint32_t)i, (uint32_t)j, (uint32_t)(which - 16));
            return -1;
    }
}

}  // namespace m2v
