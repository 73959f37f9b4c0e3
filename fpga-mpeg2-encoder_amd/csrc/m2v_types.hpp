// m2v_types.hpp — structures and constants shared by the device code (m2v_kernels.hpp) and the host translation units
// (m2v_host.hpp): what crosses a kernel launch.  No device code here: host-only sources include this, not the kernels.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace m2v {

// -DM2V_DEBUG builds libm2v_mi355x_dbg.so, the library the stage-level parity tests and the profiling scripts load:
// it can dump the quantised levels, keep every frame's reconstruction (option "keep_recon") and skip kernel phases
// (option "ablate").  The shipped library has none of that code in its kernels.
#ifdef M2V_DEBUG
constexpr bool kDebug = true;
#else
constexpr bool kDebug = false;
#endif

// ----------------------------------------------------------------------------------------------
// shared host/device structures
// ----------------------------------------------------------------------------------------------
struct Geom {
    int W, H;        // clamped luma size (RTL:985-1006)
    int mbw, mbh;    // macroblocks per row / column
    int cw, ch;      // chroma plane size
    int Q;           // Q_LEVEL
    int mbs;         // mbw * mbh
    uint32_t ysz;    // W*H
    uint32_t csz;    // cw*ch
    uint32_t rysz;   // bytes of the luma tiles of a reconstruction: (mbw + 1) * mbh * 256 (the chroma tiles follow; m2v_kernels.hpp, rec_luma_off)
    int row0, row1;  // macroblock rows this GPU encodes: [0, mbh) normally, a strip in multi-GPU strip mode
    int strip;       // 1 = strip mode: the stream buffer holds only this strip's slices, no headers
    int ablate;      // M2V_DEBUG builds only: profiling aid (option "ablate", default 0 = everything on; results are INVALID otherwise):
                     // bit0 skip full-pel search, bit1 skip half-pel SADs, bit2 skip VLC, bit3 skip IDCT/recon, bit4 skip DCT/quant
    uint32_t strip_mbs;      // (row1 - row0) * mbw
    uint32_t magic_strip;    // floor(2^32 / strip_mbs), floor(2^32 / mbw): wave-uniform divisions on the scalar unit
    uint32_t magic_mbw;      // (geom_finish() fills the three after any change of the rows)
    uint32_t s16_off;        // word offset of the 64-byte slot class inside the compact-slot buffer (plan_chunk: macroblocks of the chunk * 32)
    uint32_t s8_off;         // ... and of the 32-byte class behind it
    int cu_pack;             // experiment (xcd_remap): option "cu_pack"
    int rstride;             // macroblock rows between the launch's local rows: 1 normally; the EDGE launch of strip mode runs two local
                             // rows, the strip's first and its last (row0 and row0 + rstride)
    int edge_top, edge_bot;  // strip mode: the strip's first and last macroblock row (k_mb<.., EDGE> copies their outer rows of the
                             // reconstruction into the halo buffers)
};

inline void geom_finish(Geom &g)
{
    if (g.rstride == 0) g.rstride = 1;
    g.strip_mbs = (uint32_t)((g.row1 - g.row0) * g.mbw);
    g.magic_strip = g.strip_mbs > 1 ? (uint32_t)(0x100000000ull / g.strip_mbs) : 0xFFFFFFFFu;
    g.magic_mbw = g.mbw > 1 ? (uint32_t)(0x100000000ull / (uint32_t)g.mbw) : 0xFFFFFFFFu;
}

// Which macroblock a block of a k_mb launch works on, worked out by the HOST once per launch shape (m2v_launch.hip, mbmap_for) and read
// with one scalar load: the XCD / CU permutation (xcd_remap), the row / column split and the neighbour flags were ~60 scalar
// instructions at the head of every macroblock, on a scalar unit that four SIMDs share and that is as busy as the vector ALUs
// (profiles/r04_experiments.txt item 19).
struct MbMap {
    uint32_t mb;            // bits 0-23: macroblock index by * mbw + bx; bit 24 / 25 / 26 / 27: it has a neighbour on the left / right / above / below
                            // (inside the FRAME); bit 28: the block is one of the strip's edge rows (k_mb<.., EDGE>)
    uint32_t byx;           // by << 16 | bx
};

struct FrameJob {           // one per frame of the chunk (device memory)
    const uint8_t *in;      // 4:4:4 planar frame: Y, U, V planes of W*H bytes
    const uint8_t *ref;     // reconstruction of the previous frame (4:2:0 planar) or nullptr
    uint8_t       *rec;     // where to store this frame's reconstruction, nullptr = not needed
    int32_t        i_frame; // index inside the GOP, 0 = I frame (RTL:1078)
    uint32_t       n;       // frame number inside the sequence (time code, RTL:2685-2698)
    uint32_t       valid_beats;  // beats of real input in this frame; the rest is black (RTL:1048-1056)
    uint32_t       fidx;    // k_mb's copies in launch-list order: the frame's index in the chunk (0 in the per-frame array)
    int32_t        hidx;    // strip mode: the frame's position in its GOP step's halo list (frames whose reconstruction is referenced later), -1 = none
    int32_t        rhidx;   // ... and that of its reference frame in the previous step's list (where the neighbours' rows of it were received)
};

struct StreamCtl {          // device-resident stream bookkeeping, carried across chunks
    unsigned long long base_bytes;   // bytes of the stream already produced before this chunk
    unsigned long long total_bytes;  // bytes after this chunk (incl. final padding when last)
    unsigned long long cap_bytes;    // capacity of the output buffer
    unsigned long long prior_bytes;  // stream bytes of this sequence that left in earlier buffers (final padding rule)
    uint32_t overflow;               // 1 = the chunk did not fit, nothing was written
    uint32_t pad;
};

constexpr int kSlotWords = 304;       // per-macroblock bit slot: 3 bit-contiguous segments, <= 9300 bits
constexpr int kSmallSlotWords = 32;   // macroblocks of <= 1024 stored bits (nearly all) use a compact 128-byte slot instead (64-byte slots were
                                      // tried: k_assemble touches half the lines for P frames, but most macroblocks of an I frame then sit in
                                      // the overflow slots, which it reads word by word - no net gain, profiles/archive/r02_v_bench.json)
constexpr int kSlotChunks = kSmallSlotWords / 4;
constexpr int kMicroSlotWords = 8;    // ... and those of <= 256 bits (the median P macroblock is 134 bits) a 32-byte slot in a third array
constexpr int kTinySlotWords = 16;    // ... and those of <= 512 bits (99 % of a P frame) a 64-byte slot in a second array behind the first:
                                      // k_assemble is bound by the cache lines it touches, two of these share one

struct MbAux {                        // 16 bytes per macroblock next to the uint32 info word
    uint32_t w0;                      // lenA | lenB << 16        (bits)
    uint32_t w1;                      // lenC | dcV  << 16
    uint32_t w2;                      // dcY00 | dcY11 << 16      (quantised DC levels, 16-bit two's complement)
    uint32_t w3;                      // dcU
};

// strip mode, peer transport (k_mb<.., EDGE, PEER>; m2v_comm.hpp PeerState): the strip's whole GOP step is ONE launch whose first
// n_edge blocks are the strip's first and last macroblock row.  They store their outer rows of the reconstruction straight into
// the NEIGHBOURS' landing buffers (the launch's halo_up / halo_down: peer or IPC-mapped memory) with write-through stores, then add
// one to the neighbour's arrival counter of their GOP; before they read the rows the neighbours delivered in the previous step (nb_up /
// nb_down: this rank's own landing buffers) they wait - bounded - until this rank's own counter of the GOP has reached
// (frame's index in the GOP) x (macroblocks per row): every block of the neighbour's edge row, of every earlier frame of this GOP.
// One counter per GOP of the sequence (= position in the step's halo list) and side, kPeerCntStride words apart: a thousand
// wavefronts asking for ONE word at the start of a launch are served one after the other (~12 ns each, measured: 36 us per step).
constexpr int kPeerSlots = 256;            // GOPs of one sequence the peer form can count (more: the ordinary exchange)
constexpr int kPeerCntStride = 64;         // uint32 words between the counters of consecutive GOPs: [slot][side], 128 bytes each
struct PeerStep {
    unsigned int *cnt_up, *cnt_down;            // the neighbours' arrival counters of GOP 0 (the one this rank's top row / bottom row adds to)
    const unsigned int *seen_up, *seen_down;    // own arrival counters of GOP 0: edge blocks of the rank above / below that have delivered
    unsigned int *gaveup;                       // own word: set when a wait ran out of budget (the sequence is then encoded again, exchanged the ordinary way)
    unsigned int budget;                        // bound of one wait, in 10 ns ticks of the 100 MHz wall clock
    unsigned int n_edge;                        // blocks [0, n_edge) of the launch are the edge rows, the others the rows in between
};

// strip mode: final assembly on the output rank (k_strip_layout / k_strip_assemble)
struct CopySeg { const uint8_t *src; unsigned long long dst_off; unsigned long long len; };
struct StripSrc { const uint8_t *strip[16]; };            // by value: the strips' device pointers (<= kMaxStripRanks)
constexpr int kMaxStripRanks = 16;

constexpr uint32_t kSeqHeaderBytes = 34;   // 269 bits + alignment (RTL:2598-2617)
constexpr uint32_t kGopHeaderBytes = 8;    // 59 bits + alignment  (RTL:2650-2656)

}  // namespace m2v
