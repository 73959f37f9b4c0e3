// m2v_kernels.hpp — HIP kernels of the MI355X MPEG-2 I/P encoder (gfx950, wave64).
//
// One wavefront per macroblock.  Kernels, in launch order for a chunk of frames:
//
//   k_mb<VL,P>      stages A..T of the RTL for one macroblock: 4:4:4->4:2:0, reference window
//                   into LDS, (2YR+1)^2 full-pel SADs with v_qsad_pk_u16_u8, half-pel refine,
//                   intra/inter decision, prediction, 6x 8x8 integer DCT, quantise, zig-zag,
//                   dequantise, Chen-Wang IDCT, reconstruction, and the run/level VLC of every
//                   coefficient: the macroblock's bits that do not depend on its left neighbour
//                   leave as up to three word-aligned bit segments            (RTL:1086-2847)
//   k_slice_scan    neighbour-dependent code lengths (motion vector deltas, DC differentials),
//                   bit offset of each macroblock in its slice, byte size of each slice
//   k_frame_scan    byte offset of every frame / slice in the stream (stage V alignment rules), stream
//                   length; clears the boundary dwords of the slices and the padded tail
//   k_assemble      one workgroup per slice, one thread per macroblock: slice header + macroblock header + DC codes +
//                   the stored segments ORed at their final bit positions into an LDS image of the slice, which
//                   leaves with coalesced dword stores (stages T,U,V); the sequence / GOP / picture headers and
//                   the sequence end code (RTL:2590-2716) travel with the first / last slice
//   k_strip_layout / k_strip_assemble   strip mode (config c5): where every (frame, rank) piece goes, computed on the device; the
//                   pieces moved with 16-byte stores, headers and trailer in the same launch
//
// All arithmetic is integer with the RTL's widths; see oracle/m2v_oracle.c for the plain-C
// statement of the same semantics that the parity tests compare against.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "m2v_tables.hpp"
#include "m2v_types.hpp"

namespace m2v {


// device copies of the tables
__constant__ int8_t   c_dct[64];
__constant__ int8_t   c_dct_neg[64];      // -c_dct: the prediction's half of the residual dot product (k_mb, stage G)
__constant__ int32_t  c_dct32[64];        // c_dct widened: a lane loads its basis row as two 16-byte reads, no unpacking
__constant__ uint8_t  c_zigzag[64];
__device__ uint16_t   d_motion_code[17];
__device__ uint16_t   d_cbp_code[64];
__device__ uint16_t   d_dc_code[2][12];
__device__ uint8_t    d_dc_len[2][12];
__device__ uint16_t   d_ac_code[32 * 40];
// The same table for the symbol pass of k_mb, padded so that clamped indices need no range test: [bank][min(run, 32)][min(|level|, 41) - 1],
// row 32 and column 40 hold 0 (= escape); bank 1 differs in one entry: run 0 / level 1 as the first coefficient of a non-intra
// block is '1s' instead of '11s' (RTL:2798-2802)
constexpr int kAcRuns = 33, kAcLevels = 41;
__device__ uint16_t   d_ac_code2[2 * kAcRuns * kAcLevels];
// DCT-as-GEMM variant of stage G (k_mb<.., MFMA = true>): per-lane operands of the matrix-core formulation, lane = (g = lane >> 4,
// c = lane & 15); register v of a 16x16 accumulator holds block row 4g + v, column c.  Filled by fill_mfma_tables().
// Per-lane tables are stored QUAD-MAJOR on the device ([16-byte quad of the row][lane]): the 64 lanes of a dwordx4 load then read
// 1 KB of consecutive bytes (8 cache lines).  Row-major, every lane of such a load touches its own cache line, and a dozen
// of those per macroblock kept the vector memory pipe busier than the pixels do (profiles/archive/r02_n_*).
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
// The six 8x8 tiles of a macroblock (Y00 Y01 Y10 Y11 U V = tiles 0 .. 5) sit in the coefficient buffers of k_mb (s_t, s_zig) in SLOTS:
// tile 0 in slot 0, U in slot 1, tile 1 in slot 2, tile 3 in slot 3, V in slot 4, tile 2 in slot 5.  The matrix-core transform leaves the
// coefficients of luma tile 2 (c >> 3) + (g >> 1) in lane (g, c) (the block TRANSPOSED: a lane's four are neighbours in a row); run on the
// chroma block [U 0; 0 V] it leaves U where tile 0's are and V where tile 3's are - with this order the chroma pass stores through the
// luma pass's lane addresses plus ONE constant (one slot).
__host__ __device__ constexpr int slot_of_tile(int t) { return t == 0 ? 0 : t == 1 ? 2 : t == 2 ? 5 : t == 3 ? 3 : t == 4 ? 1 : 4; }
__host__ __device__ constexpr int tile_of_slot(int s) { return s == 0 ? 0 : s == 1 ? 4 : s == 2 ? 1 : s == 3 ? 3 : s == 4 ? 5 : 2; }
static_assert(slot_of_tile(4) == slot_of_tile(0) + 1 && slot_of_tile(5) == slot_of_tile(3) + 1 && tile_of_slot(slot_of_tile(2)) == 2 && tile_of_slot(slot_of_tile(1)) == 1, "chroma one slot behind luma tiles 0 and 3");
constexpr int kMfmaCpOff = 704;      // k_mb's kOffCp (static_assert there): the host fills MfmaLane::a1c with absolute LDS addresses
struct MfmaLane {        // three quads: b1, a, a1c | zoff | the intra quantiser's reciprocals
    uint32_t b1[2];      // pass 1 B operand: +-basis row (c & 7) for the k group that matches c's tile column, else 0
    uint32_t a2;         // pass 2 basis operand: a = basis[c & 7][4 (g & 1) .. + 3] where the tile row of c matches g >> 1, else 0; the
                         // kernel forms the pair {a, 0} (multiplies the low dword of the limb operand) and the pair {0, a} (the high dword).
                         // It is the B operand (the limbs of pass 1 are A): the product is the coefficient block transposed
    uint32_t a1c;        // pass 1 A operand of the CHROMA block [U 0; 0 V]: LDS address of s_cp[4 + (c >> 3)][c & 7][8 (g >> 1)] (kMfmaCpOff = k_mb's kOffCp)
    uint32_t zoff[4];    // byte offset of coefficient v inside s_zig: slot * 128 + zigzag position * 2
    uint32_t irecip[4];  // intra macroblocks only: ceil(2^21 / W) of the lane's four coefficients (the intra words share two otherwise
                         // unused quads - this one and SearchLane's last - so that an I frame gets all seven in two loads)
};
// luma window in LDS (k_mb): row stride in dwords, and the distance in dwords from copy A to copy B (see the LDS map in k_mb)
constexpr int kWinStride = 12;
constexpr int win_b_gap(int wrows) { return ((wrows * kWinStride - 32 + 63) / 64) * 64 + 32; }

// VECTOR_LEVEL 3 full-pel search with all 64 lanes at work (k_mb stage B).  13 x 4 lanes own a candidate group (dy', 4 dx) and
// run macroblock rows 0..12 of it; the other 12 lanes ("helpers") look like the lanes of dy' = 13 + t:
// at step i they read window row 13 + t + i like everybody else, but pair it with macroblock row 13 + i % 3 (step 12: row
// 13 + t), which is row 13 / 14 / 15 of the group with dy' = t + 3 (i / 3) (step 12: dy' = 12).  So every lane issues 13 x 4
// v_qsad instead of 16 x 4 by 52 lanes.  A helper never clears its sums: after each triple it stores the running sum, the
// owner of dy' = t + 3 k adds (sum k) - (sum k - 1); the three step-12 parts of dy' = 12 meet in one ds_add_u64 cell.
// Lane -> (dy', group): dy' = (lane >> 1) & 15, group = (lane & 1) << 1 | lane >> 5.  A ds_read_b64 is served 32 lanes at a
// time; with the group's parity in lane bit 5 a half-wave reads one window copy only (A or B, see the LDS map in k_mb) and its
// 8 x 4 pairs of dwords fall on 64 different banks whatever the distance of the copies (with lane = dy' << 2 | group half of
// the reads were 2-way conflicted).  LDS byte offsets inside k_mb<3, true>'s block, shared with the host-side lane table
// (static_assert-ed in the kernel):
__host__ __device__ constexpr int s3_dy(int lane) { return (lane >> 1) & 15; }
__host__ __device__ constexpr int s3_group(int lane) { return ((lane & 1) << 1) | (lane >> 5); }
constexpr unsigned long long kS3Helpers = 0xFC000000FC000000ull;       // the lanes with s3_dy >= 13
constexpr bool s3_helpers_match(int lane = 0) { return lane == 64 || ((((kS3Helpers >> lane) & 1ull) != 0) == (s3_dy(lane) > 12) && s3_helpers_match(lane + 1)); }
static_assert(s3_helpers_match(), "kS3Helpers does not match s3_dy");
constexpr int kS3Cur = 2 * (8 + 2 * 3) * 16;             // current luma rows (behind the two chroma windows)
constexpr int kS3Win = kS3Cur + 256;                     // luma window, copy A
constexpr int kS3Scratch = 1600 + 384 + 1536;            // the end of the transform tiles and the level buffer s_zig: unused until the transform
// dwords between the six tiles of s_t: 64 of data + 8 of padding.  With 64 the tiles start on the same bank: the quantiser's stores of the four
// luma tiles (accumulator layout) were 4-way conflicted and the column pass of the IDCT 6-way; with 72 both are conflict-free (the row pass,
// 16 bytes per lane, costs the same either way).  Same box, ms per step: 64 1.149 / 1.162, 68 1.145, 72 1.141 / 1.154, 76 1.153, 80 1.153.
constexpr int kTileStride = 72;
constexpr int kS3Flush = kS3Scratch;                     // running sums [t][group][k], 8 bytes each
constexpr int kS3Rep = kS3Flush + 3 * 4 * 4 * 8;         // macroblock rows 13 14 15 13 14 15 ... (12 x 16 bytes) for the helpers
static_assert((kS3Rep - kS3Cur) % 256 == 128, "a step reads one row of each in the same instruction: 32 banks apart");
constexpr int kS3Sum12 = kS3Rep + 192;                   // rows 13..15 of dy' = 12, per group
constexpr int kS3Zero = kS3Sum12 + 4 * 8;                // 8 zero bytes: "sum -1"
struct SearchLane {      // absolute LDS byte offsets of one lane
    uint32_t even, odd;  // its two ds_read_b64 streams through the window (k_mb stage B)
    uint32_t cur, cur12; // macroblock rows of steps 0..11 (+ 16 per step) and of step 12
    uint32_t plus, minus;// owner: the two running sums it takes its rows 13..15 from; helper: where it stores them / the dy' = 12 cell
    // position byte of the lane's four candidates, 255 - (dy' << 4 | dx + 8), and the SAD bits that mark the dx slots outside +-6
    // (and everything in a helper lane) as dead
    uint32_t cb4, dead_lo, dead_hi;
    // intra macroblocks only (nothing to do with the search: the quad has room): quantiser weights of the lane's four matrix-core
    // coefficients as bytes; weight and ceil(2^21 / W) of the lane's raster position (chroma tiles)
    uint32_t iwq4, iw, iwrecip;
};




// ----------------------------------------------------------------------------------------------
// wave helpers (wave64).  DPP row_shr 1,2,4,8 = inclusive scan inside each row of 16 lanes;
// row_bcast:15 / row_bcast:31 carry the row totals across rows.  Lane 63 ends with the total.
// ----------------------------------------------------------------------------------------------
#define M2V_DPP(old, src, ctrl, rmask, bound) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xF, (bound))

// value of lane + 4 for the lanes of banks 0 and 2 (lanes 0-3, 8-11 of every row of 16; the others get 0): one DPP move
__device__ __forceinline__ uint32_t lane_plus4(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, true);    // row_shl:4
}

__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += M2V_DPP(0, v, 0x111, 0xF, true);      // row_shr:1
    v += M2V_DPP(0, v, 0x112, 0xF, true);      // row_shr:2
    v += M2V_DPP(0, v, 0x114, 0xF, true);      // row_shr:4
    v += M2V_DPP(0, v, 0x118, 0xF, true);      // row_shr:8
    v += M2V_DPP(0, v, 0x142, 0xA, false);     // row_bcast:15 into rows 1 and 3
    v += M2V_DPP(0, v, 0x143, 0xC, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_incl(v), 63); }

// Sums of FOUR values over the wave for the price of less than two (gfx950 lane swaps): v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so one add folds two registers' 64 partial sums into 32 + 32
// lanes of a single register; v_permlane16_swap does the same for odd / even rows of 16.  After both levels each row of
// 16 lanes holds one value's partial sums and a 4-step row scan finishes all four at once: 10 VALU instead of 24.
// Totals: a -> lane 15, c -> lane 31, b -> lane 47, d -> lane 63 of the result.
__device__ __forceinline__ int wave_sum4(int a, int b, int c, int d)
{
    const auto ab = __builtin_amdgcn_permlane32_swap(a, b, false, false);      // [a.lo b.lo], [a.hi b.hi]
    const auto cd = __builtin_amdgcn_permlane32_swap(c, d, false, false);
    const int x = (int)(ab[0] + ab[1]), y = (int)(cd[0] + cd[1]);              // rows: a a b b / c c d d
    const auto xy = __builtin_amdgcn_permlane16_swap(x, y, false, false);      // [x0 y0 x2 y2], [x1 y1 x3 y3]
    int v = (int)(xy[0] + xy[1]);                                              // rows: a c b d
    v += M2V_DPP(0, v, 0x111, 0xF, true);
    v += M2V_DPP(0, v, 0x112, 0xF, true);
    v += M2V_DPP(0, v, 0x114, 0xF, true);
    v += M2V_DPP(0, v, 0x118, 0xF, true);
    return v;
}

// A value that IS the same in every lane, stated to the compiler: everything computed from it stays on the scalar
// unit (SALU has slack, the vector ALU is the kernel's bottleneck).  Without it the compiler's divergence analysis
// gives up on some wave-uniform chains (the half-pel decision, the coded-block pattern) and runs them - and the
// exec-mask juggling of their "divergent" branches - on the vector unit.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// The same statement for a value the compiler already holds in a scalar register: the empty asm pins it to an SGPR and
// hides where it came from, so that it cannot be merged with an equal expression that some vector instruction needs as
// a lane mask or VGPR (which would drag this copy, and its users, onto the vector ALU as well).
__device__ __forceinline__ int sgpr(int v) { asm("" : "+s"(v)); return v; }
// EXEC-masked lane mask of a predicate as the compare instruction leaves it (HIP's __ballot goes through a 0/1 select
// and a second compare: two VALU instructions more per use)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// 1 if any lane's predicate holds, on the scalar unit
__device__ __forceinline__ uint32_t any_lane(bool p)
{
    const int n = sgpr(__builtin_popcountll(ballot(p)));
    // n > 0 as 0 / 1 without a boolean (see find_min_in_10_values); pinned to a scalar register: (x << 1) | (-n >> 31) is otherwise
    // matched as a funnel shift, which only the vector ALU has (v_mov + v_alignbit + v_readfirstlane per use)
    return (uint32_t)sgpr((int)((uint32_t)-n >> 31));
}

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x)
{
    // old = ~0 is the identity of min: lanes without a source (and the rows a row_bcast skips) keep their value, and the
    // compiler folds each move into ONE v_min_u32_dpp (with old = v it emits v_mov + v_mov_dpp + v_min: three)
    int v = (int)x;
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x111, 0xF, false));
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x112, 0xF, false));
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x114, 0xF, false));
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x118, 0xF, false));
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x142, 0xA, false));
    v = (int)umin32((uint32_t)v, (uint32_t)M2V_DPP(-1, v, 0x143, 0xC, false));
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int mean2(int a, int b) { return (a + b + 1) >> 1; }                    // RTL:750-757
__device__ __forceinline__ int mean4(int a, int b, int c, int d) { return (a + b + c + d + 1) >> 2; } // RTL:760-767
// a * b + c with 24-bit operands in ONE instruction.  Written out because the compiler, seeing the int8 basis table,
// prefers v_mul_i32_i24 with an SDWA byte select plus a separate add: two instructions per MAC, 48 extra per macroblock.
__device__ __forceinline__ int mad24(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// the same with the addend in a scalar register (the chain's rounding constant: no v_mov to start an accumulator)
__device__ __forceinline__ int mad24_s(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// the same with the multiplier in a scalar register
__device__ __forceinline__ int mad24_ms(int a, int b, int c)
{
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b), "v"(c));
    return d;
}
// sign(q) in {-1, 0, 1}: one v_med3_i32 with inline constants (the compiler builds it from two compares and two selects)
__device__ __forceinline__ int sign_of(int q)
{
    int d;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(d) : "v"(q));
    return d;
}
// Keeps every component of a loaded vector alive up to this point (no instruction).  A table quad with a component that some
// instantiation does not use would otherwise have that register reused right after the load is issued - and the compiler then
// waits for the load (write-after-write) long before its data is needed.
template <typename T>
__device__ __forceinline__ void keep_alive(const T &v)
{
    asm volatile("" : : "v"(v));
}
// a constant that stays in ONE vector register (the compiler otherwise re-materialises it with a v_mov at every use)
__device__ __forceinline__ int vgpr_const(int c)
{
    asm("" : "+v"(c));
    return c;
}
// clamp with both bounds in vector registers: one v_med3_i32 (two literal bounds cost the compiler a v_mov per use)
__device__ __forceinline__ int clamp_vv(int x, int lo, int hi)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(lo), "v"(hi));
    return d;
}
// first link of a v_dot4 chain: the VOP3P form takes the inline constant 0 as accumulator (v_dot4c needs a zeroed register)
__device__ __forceinline__ int dot4_first(uint32_t a, uint32_t b)
{
    int d;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ int iabs(int a) { return a < 0 ? -a : a; }
__device__ __forceinline__ int sext(int v, int bits) { return (int)((uint32_t)v << (32 - bits)) >> (32 - bits); }

// the same two means on four packed bytes with v_lerp_u8: D.b = (S0.b + S1.b + (S2.b & 1)) >> 1
__device__ __forceinline__ uint32_t avg2x4(uint32_t a, uint32_t b)           // (a+b+1)>>1 per byte
{
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);
}
// (a+b+c+d+1)>>2 per byte == (floor((a+b)/2) + floor((c+d)/2) + ((a^b)|(c^d))&1) >> 1, exhaustively
// checked in tests/test_host_logic.py::test_mean4_lerp_identity
__device__ __forceinline__ uint32_t avg4x4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return __builtin_amdgcn_lerp(__builtin_amdgcn_lerp(a, b, 0u), __builtin_amdgcn_lerp(c, d, 0u), (a ^ b) | (c ^ d));
}

// (a + b + c + d + 2) >> 2 per byte, the ISO/IEC 13818-2 7.6.4 rounding (option "conformant" only): with the floor
// means s = (a+b)>>1, t = (c+d)>>1 and their lost bits la, lc the sum is 2s + 2t + la + lc, so the result is
// (s + t + 1 + (la & lc)) >> 1 = lerp(s, t, 1) plus one where la & lc and s + t is even (checked exhaustively over
// (s, t, la, lc) in tests/test_host_logic.py); the per-byte result is <= 255, so the packed add cannot carry
__device__ __forceinline__ uint32_t avg4x4_iso(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const uint32_t s2 = __builtin_amdgcn_lerp(a, b, 0u), t2 = __builtin_amdgcn_lerp(c, d, 0u);
    const uint32_t both = (a ^ b) & (c ^ d);
    return __builtin_amdgcn_lerp(s2, t2, 0x01010101u) + (both & ~(s2 ^ t2) & 0x01010101u);
}

template <bool CONF>
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    if constexpr (CONF) return avg4x4_iso(a, b, c, d);
    else return avg4x4(a, b, c, d);
}

// The reconstruction (= the next frame's reference) is stored TILED: 256-byte luma tiles (16 rows of 16 bytes), and behind all of
// those (g.rysz bytes) 128-byte chroma tiles - 8 rows of 8 of U, then of V.  The kernel that writes a frame and the kernel that reads
// it as a reference are the same one, so the layout is free (profiles/r04_experiments.txt items 11 and 14).  The tiles are SHIFTED by half
// a macroblock: luma tile (ty, tx), tx = 0 .. mbw, holds frame columns 16 tx - 8 .. 16 tx + 7, chroma tile tx columns 8 tx - 4 .. 8 tx + 3
// (mbw + 1 tiles per tile row, the outer halves of the first and last unused).  The +-YR window of macroblock bx - columns
// 16 bx - 8 .. 16 bx + 23 - is then exactly the tile columns bx and bx + 1: every window row is two full 16-byte tile rows, the whole
// window ONE 16-byte load per lane (one 8-byte load for the chroma windows); a macroblock's reconstruction leaves as the right half
// of tile bx and the left half of tile bx + 1.  Byte offset of luma sample (x, y) / of sample (x, y) of chroma plane pl:
__device__ __forceinline__ uint32_t rec_luma_off(uint32_t x, uint32_t y, const Geom &g)
{
    return (__umul24(y >> 4, (uint32_t)g.mbw + 1u) + ((x + 8u) >> 4)) * 256u + ((y & 15u) << 4) + ((x + 8u) & 15u);
}
__device__ __forceinline__ uint32_t rec_chroma_off(uint32_t pl, uint32_t x, uint32_t y, const Geom &g)
{
    return g.rysz + (__umul24(y >> 3, (uint32_t)g.mbw + 1u) + ((x + 4u) >> 3)) * 128u + (pl << 6) + ((y & 7u) << 3) + ((x + 4u) & 7u);
}

// XCD-aware block remap: consecutive logical blocks land on the same XCD (shared L2 for the
// overlapping reference windows of neighbouring macroblocks).  Bijective for any grid size and any cu_pack
// (tests/test_abi.py walks it on the host through m2v_debug_table).
__host__ __device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n, uint32_t cu_pack = 0)
{
    const uint32_t xcd = b & 7u, q = n >> 3, r = n & 7u;
    const uint32_t start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    uint32_t i = b >> 3;                    // the XCD's i-th block
    // cu_pack (option "cu_pack", default 5 = 32 CUs per XCD): the dispatcher deals an XCD's workgroups to its CUs in turn, so blocks i,
    // i + 32, i + 64 ... of an XCD tend to land on ONE CU one after the other (observed, for speed only - the map is a bijection whatever
    // the hardware does).  Those eight get horizontally neighbouring macroblocks: they read the same 128-byte lines of the frame and of
    // the reference, and the CU's L1 serves some of the repeats instead of the L2 (-1.8 % per sequence, profiles/r04_experiments.txt item 7)
    if (cu_pack) {
        const uint32_t cus = 1u << cu_pack, span = cus << 3, qq = q + ((xcd - r) >> 31);      // q + 1 for xcd < r, by the sign bit (a 0 / 1 from a compare goes through the vector ALU)
        if (i < (qq / span) * span) {
            const uint32_t c = i & (cus - 1u), j = i >> cu_pack;
            i = (j >> 3) * span + (c << 3) + (j & 7u);
        }
    }
    return start + i;
}

// n / d for a wave-uniform n with M = floor(2^32 / d) (0xFFFFFFFF for d = 1): the estimate mulhi(n, M) is the
// quotient or one less, so a single correction makes it exact; s_mul_hi_u32 + a few SALU instead of the ~12 VALU
// instructions of the float-reciprocal sequence the compiler emits for a 32-bit division
__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t M)
{
    const uint32_t q = __umulhi(n, M);
    return n - q * d >= d ? q + 1u : q;
}

// ----------------------------------------------------------------------------------------------
// Chen-Wang IDCT passes (RTL:844-972).  32-bit wrapping arithmetic like the RTL's reg [31:0].
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void idct_row(const int a[8], int r[8])
{
    int x0 = a[0], x1 = a[4], x2 = a[6], x3 = a[2], x4 = a[1], x5 = a[7], x6 = a[5], x7 = a[3], x8;
    x0 = (int)((uint32_t)x0 << 11) | 128;      // RTL:857-859
    x1 = (int)((uint32_t)x1 << 11);
    // inputs are 13-bit coefficients: every product below fits the 24-bit multiplier
    x8 = __mul24(kW7, x4 + x5);
    x4 = x8 + __mul24(kW1 - kW7, x4);
    x5 = x8 - __mul24(kW1 + kW7, x5);
    x8 = __mul24(kW3, x6 + x7);
    x6 = x8 - __mul24(kW3 - kW5, x6);
    x7 = x8 - __mul24(kW3 + kW5, x7);
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = __mul24(kW6, x3 + x2);
    x2 = x1 - __mul24(kW2 + kW6, x2);
    x3 = x1 + __mul24(kW2 - kW6, x3);
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    x2 = (181 * (x4 + x5) + 128) >> 8;
    x4 = (181 * (x4 - x5) + 128) >> 8;
    r[0] = sext((x7 + x1) >> 8, 18);           // stored in 18 bits (RTL:886, 2170)
    r[1] = sext((x3 + x2) >> 8, 18);
    r[2] = sext((x0 + x4) >> 8, 18);
    r[3] = sext((x8 + x6) >> 8, 18);
    r[4] = sext((x8 - x6) >> 8, 18);
    r[5] = sext((x0 - x4) >> 8, 18);
    r[6] = sext((x3 - x2) >> 8, 18);
    r[7] = sext((x7 - x1) >> 8, 18);
}

__device__ __forceinline__ int clip255(int v) { return v < -255 ? -255 : v > 255 ? 255 : v; }   // RTL:778-783

__device__ __forceinline__ void idct_col(const int a[8], int r[8])
{
    int x0 = a[0], x1 = a[4], x2 = a[6], x3 = a[2], x4 = a[1], x5 = a[7], x6 = a[5], x7 = a[3], x8;
    x0 = (int)((uint32_t)x0 << 8) + 8192;      // RTL:924-926
    x1 = (int)((uint32_t)x1 << 8);
    // inputs are 18-bit row results: every product below fits the 24-bit multiplier
    x8 = __mul24(kW7, x4 + x5) + 4;
    x4 = (x8 + __mul24(kW1 - kW7, x4)) >> 3;
    x5 = (x8 - __mul24(kW1 + kW7, x5)) >> 3;
    x8 = __mul24(kW3, x6 + x7) + 4;
    x6 = (x8 - __mul24(kW3 - kW5, x6)) >> 3;
    x7 = (x8 - __mul24(kW3 + kW5, x7)) >> 3;
    x8 = x0 + x1;
    x0 = x0 - x1;
    x1 = __mul24(kW6, x3 + x2) + 4;
    x2 = (x1 - __mul24(kW2 + kW6, x2)) >> 3;
    x3 = (x1 + __mul24(kW2 - kW6, x3)) >> 3;
    x1 = x4 + x6;
    x4 = x4 - x6;
    x6 = x5 + x7;
    x5 = x5 - x7;
    x7 = x8 + x3;
    x8 = x8 - x3;
    x3 = x0 + x2;
    x0 = x0 - x2;
    x2 = (181 * (x4 + x5) + 128) >> 8;
    x4 = (181 * (x4 - x5) + 128) >> 8;
    // The RTL clips these to +-255 (RTL:778-783, 963-970) before add_clip_0_255 adds the prediction p in 0..255 and
    // clips to 0..255 (RTL:786-795).  The first clip cannot change the final pixel: v > 255 gives 255 + p >= 255 -> 255
    // either way, v < -255 gives -255 + p <= 0 -> 0 either way; so only the final clip is done (by the caller).
    r[0] = (x7 + x1) >> 14;
    r[1] = (x3 + x2) >> 14;
    r[2] = (x0 + x4) >> 14;
    r[3] = (x8 + x6) >> 14;
    r[4] = (x8 - x6) >> 14;
    r[5] = (x0 - x4) >> 14;
    r[6] = (x3 - x2) >> 14;
    r[7] = (x7 - x1) >> 14;
}


// ----------------------------------------------------------------------------------------------
// VLC helpers shared by k_mb (coefficients) and k_slice_scan / k_assemble (neighbour-dependent codes)
// ----------------------------------------------------------------------------------------------

struct BitCode { uint32_t code, len; };

__device__ __forceinline__ void lds_put(uint32_t *buf, uint32_t pos, uint32_t val, uint32_t len)
{
    if (!len) return;
    const uint32_t w = pos >> 5, b = pos & 31u;
    const unsigned long long v = (unsigned long long)val << (64u - len - b);
    atomicOr(&buf[w], (uint32_t)(v >> 32));
    const uint32_t lo = (uint32_t)v;
    if (lo) atomicOr(&buf[w + 1], lo);
}

// dct_dc_size + dct_dc_differential (RTL:2808-2821), at most 10 + 11 bits
__device__ __forceinline__ BitCode dc_code(int diff, int chroma)
{
    const int a = iabs(diff);
    const int size = a ? 32 - __clz(a) : 0;
    uint32_t bits = (uint32_t)diff & 0xFFFu;
    if (diff < 0) bits = (bits + ((1u << size) - 1u)) & 0xFFFu;
    const uint32_t sl = d_dc_len[chroma][size];
    return BitCode{((uint32_t)d_dc_code[chroma][size] << size) | bits, sl + (uint32_t)size};
}

// The same for a wave-uniform luma differential, on the scalar unit: the size code comes through a scalar load from `tab`
// (kConstDcLuma: code | length << 16), everything else is integer arithmetic on uniform values.  k_mb codes the DC of
// Y01 / Y10 / Y11 of every intra macroblock with it (the vector form costs 17 instructions in all 64 lanes for one value).
__device__ __forceinline__ BitCode dc_code_uniform(int diff, const uint8_t *tab)
{
    diff = sgpr(diff);
    const int a = diff < 0 ? -diff : diff;
    const int size = sgpr(a ? 32 - __builtin_clz((unsigned)a) : 0);
    uint32_t bits = (uint32_t)diff & 0xFFFu;
    bits = (bits + ((uint32_t)(diff >> 31) & ((1u << size) - 1u))) & 0xFFFu;
    const uint32_t e = *(const __attribute__((address_space(4))) uint32_t *)(tab + 4 * size);
    return BitCode{(uint32_t)sgpr((int)(((e & 0xFFFFu) << size) | bits)), (uint32_t)sgpr((int)((e >> 16) + (uint32_t)size))};
}

// motion_code + sign of the wrapped delta (RTL:2736-2748), at most 11 bits
__device__ __forceinline__ BitCode mv_code(int mv, int prev)
{
    int d = mv - prev;
    if (d > 15) d -= 32; else if (d < -16) d += 32;
    const uint32_t e = d_motion_code[iabs(d)];
    uint32_t code = e & 255u, len = e >> 8;
    if (d != 0) { code = (code << 1) | (d < 0 ? 1u : 0u); len += 1; }
    return BitCode{code, len};
}

// The parts of a macroblock that need the left neighbour (predictors reset at the start of a slice,
// RTL:2713-2715): p1 = type [+ mvx + mvy | + DC of Y00], p2 = DC of U, p3 = DC of V.
struct MbDep { BitCode p1, p2, p3; };

__device__ __forceinline__ MbDep mb_dependent(uint32_t info, const MbAux &aux, bool has_left, uint32_t linfo,
                                              const MbAux &laux, int i_frame)
{
    const int inter = info & 1, cbp = (info >> 1) & 63;
    MbDep d{};
    // macroblock_address_increment '1' + macroblock_type (RTL:2722-2731)
    if (!inter && i_frame != 0) d.p1 = BitCode{0x23, 6};
    else if (inter && cbp == 0) d.p1 = BitCode{0x09, 4};
    else                        d.p1 = BitCode{0x03, 2};
    if (inter) {
        int pmvx = 0, pmvy = 0;                 // vectors carry over from an inter neighbour only (RTL:2769-2773)
        if (has_left && (linfo & 1)) { pmvx = (int8_t)(linfo >> 8); pmvy = (int8_t)(linfo >> 16); }
        const BitCode cx = mv_code((int8_t)(info >> 8), pmvx), cy = mv_code((int8_t)(info >> 16), pmvy);
        d.p1.code = (((d.p1.code << cx.len) | cx.code) << cy.len) | cy.code;
        d.p1.len += cx.len + cy.len;
    } else {
        int pY = 0, pU = 0, pV = 0;             // DC predictors: last Y tile / U / V of an intra neighbour, else 0 (RTL:2786-2792)
        if (has_left && !(linfo & 1)) { pY = (int16_t)(laux.w2 >> 16); pU = (int16_t)laux.w3; pV = (int16_t)(laux.w1 >> 16); }
        const BitCode c0 = dc_code((int16_t)aux.w2 - pY, 0);
        d.p1.code = (d.p1.code << c0.len) | c0.code;
        d.p1.len += c0.len;
        d.p2 = dc_code((int16_t)aux.w3 - pU, 1);
        d.p3 = dc_code((int16_t)(aux.w1 >> 16) - pV, 1);
    }
    return d;
}

// Symbol list entry, 32 bits.  A level: [15:0] level, [25:20] zig-zag position, everything else 0.  A raw code: [19:0] code,
// [31:27] its length (never 0: that is how the two kinds are told apart), [26:20] the "position" a following level measures its
// run from, as a signed 7-bit number: 0 behind an intra DC code (position 0 is the DC), -1 at the start of a non-intra block -
// whose bit 26 also selects bank 1 of d_ac_code2 for the level right behind it.
__host__ __device__ constexpr uint32_t sym_raw(uint32_t len, uint32_t code, bool inter_start) { return (len << 27) | (inter_start ? 127u << 20 : 0u) | code; }

// Pass 1 of the coefficient VLC for one coded tile (lane = zig-zag index): rank the non-zero levels, append their
// {position, level} symbols and the end_of_block code to the macroblock's symbol list.  The run of a level is formed in pass 2
// from the position of the symbol before it (a raw symbol = block start).  Trimmed for VECTOR instruction count (a non-intra
// macroblock runs this five times): the list position is kept in BYTES (no shifts); only the non-zero lanes store, EXEC
// narrowed by scalar moves (vlc_store_symbols).  sym_base = LDS byte address of the list, eob = the end-of-block symbol in a
// register; returns the new byte length.  First the non-intra form: every position counts and there is no DC code.
// The stores of pass 1: the lanes of `mask` (the non-zero levels) put their symbol into their slot of the list, and the
// end_of_block symbol '10' (RTL:2835) ends up behind the last of them.  EXEC is narrowed by scalar moves around the stores: the
// vector ALU is the unit this kernel is bound by, and the round-2 form (zero lanes redirected to a dump word by a v_cndmask, the
// end code's wave-uniform address moved to a vector register, the dump address re-materialised) cost three vector instructions
// per tile.  Every storing lane first puts the end code BEHIND its own slot, then its symbol INTO its slot: LDS operations of a
// wavefront execute in order, so every end code but the last is overwritten by the next lane's symbol.  MAY_BE_EMPTY (intra
// tiles: the mask leaves out lane 0, the DC, and there may be no AC level at all, but the end code is unconditional): lane 0,
// whose slot is the first one, puts an end code INTO it beforehand, where the first level - if there is one - overwrites it.
template <bool MAY_BE_EMPTY>
__device__ __forceinline__ void vlc_store_symbols(uint32_t slot, uint32_t sym, uint32_t eob, unsigned long long mask)
{
    unsigned long long saved;
    if constexpr (MAY_BE_EMPTY)
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %3\n\t"
                     "s_mov_b64 exec, %4\n\tds_write_b32 %1, %3 offset:4\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(slot), "v"(sym), "v"(eob), "s"(mask) : "memory");
    else
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\tds_write_b32 %1, %3 offset:4\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                     : "=&s"(saved) : "v"(slot), "v"(sym), "v"(eob), "s"(mask) : "memory");
}

__device__ __forceinline__ uint32_t vlc_tile_symbols_inter(const int16_t *zig, uint32_t sym_base, int lane, uint32_t lane_pos, uint32_t nsym4, uint32_t eob)
{
    const int v = zig[lane];
    const unsigned long long mask = ballot(v != 0);
    const uint32_t end4 = (uint32_t)sgpr((int)(nsym4 + 4u * (uint32_t)__builtin_popcountll(mask)));
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t slot = (rank << 2) + (uint32_t)sgpr((int)(sym_base + nsym4));
    vlc_store_symbols<false>(slot, lane_pos | ((uint32_t)v & 0xFFFFu), eob, mask);     // a coded non-intra tile has at least one level
    return (uint32_t)sgpr((int)(end4 + 4u));
}

// and for a tile of an intra macroblock: lane 0 holds the DC level, which leaves through `dc` (for Y01 / Y10 / Y11 its differential
// against `dc_prev` is coded right here, RTL:2784-2786) and is cleared from the mask on the scalar side
__device__ __forceinline__ uint32_t vlc_tile_symbols_intra(const int16_t *zig, uint32_t sym_base, int lane, uint32_t lane_pos, uint32_t nsym4, uint32_t eob,
                                                           int &dc, int dc_prev, bool dc_chained, const uint8_t *dc_tab)
{
    typedef __attribute__((address_space(3))) uint32_t *LdsW;
    const int v = zig[lane];
    dc = __builtin_amdgcn_readlane(v, 0);
    nsym4 = (uint32_t)sgpr((int)nsym4);
    if (dc_chained) {
        const BitCode c = dc_code_uniform(dc - dc_prev, dc_tab);
        if (lane == 0) *(LdsW)(uintptr_t)(sym_base + nsym4) = sym_raw(c.len, c.code, false);
        nsym4 += 4u;
    }
    const unsigned long long mask = ballot(v != 0) & ~1ull;
    const uint32_t end4 = (uint32_t)sgpr((int)(nsym4 + 4u * (uint32_t)__builtin_popcountll(mask)));
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    const uint32_t slot = (rank << 2) + (uint32_t)sgpr((int)(sym_base + nsym4));
    vlc_store_symbols<true>(slot, lane_pos | ((uint32_t)v & 0xFFFFu), eob, mask);
    return (uint32_t)sgpr((int)(end4 + 4u));
}

// ----------------------------------------------------------------------------------------------
// k_mb: one wavefront = one macroblock, stages A..T
// ----------------------------------------------------------------------------------------------
// One wavefront per workgroup: __syncthreads() lowers to "s_waitcnt lgkmcnt(0); wave barrier" (no s_barrier), which is
// exactly the LDS write -> read ordering the phases need, and the static LDS base keeps every DS offset an immediate.
// (Measured alternatives: 4 wavefronts per workgroup with per-wave LDS regions -16 %, a seq_cst wavefront fence -14 %.)
#define M2V_WAVE_SYNC() __syncthreads()
// -DM2V_DEBUG profiling aid: option "ablate" = n << 8 ends the kernel at stop point n (tools/phase_valu.sh: the counters of the
// truncated kernels give the instruction count of every phase by difference; output invalid)
#define M2V_STOP(n) do { if (kDebug && (g.ablate >> 8) == (n)) return; } while (0)

typedef __attribute__((address_space(3))) uint32_t LdsU32;
struct QsadRow { unsigned long long w01, w12, w23, w34; };    // the four overlapping 8-byte reference operands of one window row

// full-pel search: the operands of a window row are issued one row ahead of their v_qsad (p / q alternate), by hand-issued ds_read_b64 - left to
// itself the compiler fuses them into ds_read2_b64, which runs at half the LDS rate (MI355X_MICROARCH.md, LDS table).  ae / ao: LDS byte addresses
// of the lane's even / odd dword pairs in its first window row.
// Measurement aid of experiment 19 (profiles/r06_experiments.txt; tools/variant_build.sh ub=-DM2V_EXP19_QSAD_PER_STEP=3): with 3 every
// step of the VECTOR_LEVEL 3 search drops its fourth v_qsad - a quarter of the search's vector cycles gone, results INVALID - which
// bounds from above what any re-packing of the search's 39 dead candidate slots (of 208) could buy.  4 = the kernel as shipped.
#ifndef M2V_EXP19_QSAD_PER_STEP
#define M2V_EXP19_QSAD_PER_STEP 4
#endif
constexpr int kExp19QsadPerStep = M2V_EXP19_QSAD_PER_STEP;

typedef const __attribute__((address_space(3))) u32x4_t *LdsU4;
typedef const __attribute__((address_space(3))) u32x2_t *LdsU2;
// VECTOR_LEVEL 1 / 2 (round 6): the (dy, dx group) pairs of the small ranges are few - 10 and 27 - so every pair is given to SEVERAL lanes, each
// running a part of the macroblock's sixteen rows: NR rows per lane instead of 16.  cur: LDS byte address of the part's first current row.
template <int RR, int NR, int WS>
__device__ __forceinline__ void search_rows_part(uint32_t cur, uint32_t ae, uint32_t ao, unsigned long long &acc, QsadRow &p, QsadRow &q)
{
    if constexpr (RR < NR) {
        if constexpr (RR + 1 < NR) {
            asm volatile("ds_read_b64 %0, %8 offset:%10\n\tds_read_b64 %1, %9 offset:%10\n\t"
                         "ds_read_b64 %2, %8 offset:%11\n\tds_read_b64 %3, %9 offset:%11\n\t"
                         "s_waitcnt lgkmcnt(4)"
                         : "=&v"(q.w01), "=&v"(q.w12), "=&v"(q.w23), "=&v"(q.w34),
                           "+v"(p.w01), "+v"(p.w12), "+v"(p.w23), "+v"(p.w34)
                         : "v"(ae), "v"(ao), "n"((RR + 1) * WS * 4), "n"((RR + 1) * WS * 4 + 8));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.w01), "+v"(p.w12), "+v"(p.w23), "+v"(p.w34));
        }
        const u32x4_t c = *(LdsU4)(uintptr_t)(cur + RR * 16);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w01, c.x, acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w12, c.y, acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w23, c.z, acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w34, c.w, acc);
        search_rows_part<RR + 1, NR, WS>(cur, ae, ao, acc, q, p);
    }
}

// The same for VECTOR_LEVEL 3 with the helper lanes (see kS3Cur above): 13 steps, the current row through a per-lane address,
// the helpers' running sums stored after steps 2, 5, 8, 11 with EXEC narrowed to them (hmask) inside the asm statement.
template <int RR>
__device__ __forceinline__ void search_rows13(uint32_t cur, uint32_t cur12, uint32_t ae, uint32_t ao, uint32_t flush, unsigned long long hmask,
                                              unsigned long long &acc, QsadRow &p, QsadRow &q)
{
    if constexpr (RR < 13) {
        if constexpr (RR + 1 < 13) {
            asm volatile("ds_read_b64 %0, %8 offset:%10\n\tds_read_b64 %1, %9 offset:%10\n\t"
                         "ds_read_b64 %2, %8 offset:%11\n\tds_read_b64 %3, %9 offset:%11\n\t"
                         "s_waitcnt lgkmcnt(4)"
                         : "=&v"(q.w01), "=&v"(q.w12), "=&v"(q.w23), "=&v"(q.w34),
                           "+v"(p.w01), "+v"(p.w12), "+v"(p.w23), "+v"(p.w34)
                         : "v"(ae), "v"(ao), "n"((RR + 1) * kWinStride * 4), "n"((RR + 1) * kWinStride * 4 + 8));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p.w01), "+v"(p.w12), "+v"(p.w23), "+v"(p.w34));
        }
        const u32x4_t c = RR < 12 ? *(LdsU4)(uintptr_t)(cur + RR * 16) : *(LdsU4)(uintptr_t)cur12;
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w01, c.x, acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w12, c.y, acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w23, c.z, acc);
        if constexpr (kExp19QsadPerStep >= 4) acc = __builtin_amdgcn_qsad_pk_u16_u8(p.w34, c.w, acc);
        if constexpr (RR == 2 || RR == 5 || RR == 8)
            asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %1, %0 offset:%3\n\ts_mov_b64 exec, -1"
                         : : "v"(acc), "v"(flush), "s"(hmask), "n"((RR / 3) * 8) : "memory");
        if constexpr (RR == 11)      // the last triple; step 12 starts from zero (its parts are summed by ds_add_u64)
            asm volatile("s_mov_b64 exec, %2\n\tds_write_b64 %1, %0 offset:24\n\ts_nop 0\n\tv_mov_b64 %0, 0\n\ts_mov_b64 exec, -1"
                         : "+v"(acc) : "v"(flush), "s"(hmask) : "memory");
        search_rows13<RR + 1>(cur, cur12, ae, ao, flush, hmask, acc, q, p);
    }
}

// Everything in k_mb that depends on the lane number alone - pixel / window coordinates, LDS addresses in the tile layouts,
// the lane's row and column of the transform basis - comes from this table instead of being recomputed by every wavefront
// (about 60 shifts, masks and adds per macroblock on a vector ALU that is the kernel's bottleneck; a load costs it nothing).
// Constants that many lanes share (the transform basis rows) stay in their compact tables: per lane they would only multiply
// the bytes the vector memory pipe has to move.
// The table of an instantiation k_mb<VL, P> is written by the SAME kernel template (FILL = true: the lambda lane_consts below is
// its only source), so the values cannot drift from the LDS map they describe.
struct LaneK {
    // quads 0, 1: requested with the pixels; quad 2 onwards: before the half-pel phase (registers)
    uint32_t win_st;                // LDS address of s_win[(lane >> 1) kWS + 4 (lane & 1)]: the lane's 16 bytes of the window (lanes < 2 WROWS)
    uint32_t hp;                    // LDS address of s_win[r kWS + c4]: the lane part of the half-pel neighbourhood
    uint32_t pred_st, cp_st;        // &s_pred[tile][ti], &s_cp[tile][r & 7][(c4 & 1) << 2]
    uint32_t cpc_st;                // &s_cp[4][r >> 1][2 c4]
    uint32_t cpred_st, cpc2_st;     // &s_pred[4 + pl][yc << 3 | xc], &s_cp[4 + pl][yc][8 + xc]
    uint32_t cwin_rd;               // &s_cwin[pl][(yc + UR) * 4]: the lane part of the chroma prediction fetch
    uint32_t row_st;                // row pass of the inverse transform: &s_t[slot][row * 8] of the lane's row (matrix-core transform: luma rows in lanes 0-15 / 32-47, chroma in 16-31; else unused)
    uint32_t cp_rd;                 // &s_cp[0][lane >> 3][0]
    uint32_t a1;                    // pass-1 A operand of the matrix-core transform
    uint32_t xrow;                  // &s_t[..] of the lane's first accumulator register (dequantised coefficients, raster order)
    uint32_t zz2;                   // 2 * zig-zag position of the lane
    uint32_t col_rd, col_pred;      // &s_t[slot][col], &s_pred[tile of that slot][col] of the column pass
    uint32_t crec_rd;               // &s_pred[4 + pl][yc << 3 | half << 2] of the chroma store
    uint32_t crec_r, crec_c, crec_p;// yc, 4 half, pl
    uint32_t pad;
};
static_assert(sizeof(LaneK) == 80 && sizeof(MfmaLane) == 48 && sizeof(SearchLane) == 48, "whole quads");
// One block of quads per instantiation k_mb<VL, P>: LaneK (written by the FILL instantiation), then SearchLane and MfmaLane
// (uploaded by the host, the same in every block).  One block = one scalar base register pair for all of the kernel's lane
// tables: the base points 4 KB into the block, so that the first eight quads are reached by the load's immediate offset.
constexpr int kQuadsLaneK = sizeof(LaneK) / 16, kQuadsSearch = sizeof(SearchLane) / 16, kQuadsMfma = sizeof(MfmaLane) / 16;
constexpr int kQuadSearch0 = kQuadsLaneK, kQuadMfma0 = kQuadSearch0 + kQuadsSearch;
// behind the lane quads, the small constant tables k_mb reads (copies: the originals stay where the other kernels and the FILL
// pass read them) - reached from the same pinned bases instead of one s_getpc_b64 / s_add_u32 / s_addc_u32 triple per access
constexpr int kQuadConst0 = kQuadMfma0 + kQuadsMfma;                 // 1 KB: c_dct32 | c_dct | c_dct_neg | d_cbp_code
constexpr int kConstDct32 = 0, kConstDct = 256, kConstCbp = 384;     // kConstDct: 8 rows of [8 basis bytes | their 8 negatives]
constexpr int kConstDcLuma = 512;            // 12 dwords: dct_dc_size_luminance code | length << 16 (read by the SCALAR unit, dc_code_uniform)
// the half-pel decision's dead candidates (RTL:1757-1760) as ready-made words, read by the SCALAR unit: entry F = no_l | no_r << 1 |
// no_u << 2 | no_d << 3, five dwords each - one per packed pair of half-pel SAD totals (candidates 0|1, 2|3, 4|5, 6|7, 8|-), bit 12
// set where the pair's low candidate is dead, bit 28 where its high one is (hp_dead_word below; the host fills the table with it)
constexpr int kConstHpDead = 576, kHpDeadStride = 20;
static_assert(kConstHpDead + 16 * kHpDeadStride <= 1024, "inside the constants' KB");
__host__ __device__ constexpr uint32_t hp_dead_bit(int k, int F)      // is half-pel candidate k = 3 (hy + 1) + (hx + 1) dead under F?
{
    return (uint32_t)(((k % 3 == 0) && (F & 1)) || ((k % 3 == 2) && (F & 2)) || ((k / 3 == 0) && (F & 4)) || ((k / 3 == 2) && (F & 8)));
}
__host__ __device__ constexpr uint32_t hp_dead_word(int pair, int F)
{
    return (hp_dead_bit(2 * pair, F) << 12) | (pair < 4 ? hp_dead_bit(2 * pair + 1, F) << 28 : 0u);
}
// The RTL's 10-way argmin (find_min_in_10_values, RTL:804-840) is the minimum of (value, rank): among equal values candidate 8 wins, then
// 9, then 4, 5, 6, 7, then 0, 1, 2, 3 (the tree compares the pairs with "<", the two halves 0-3 / 4-7 so that 4-7 win a tie, and 8-9 with "<="
// against both) - proved against the tree over every tie pattern in tests/test_host_logic.py.  kHpRank[k]: the rank of candidate k;
// kHpRankHy / kHpRankHx: (hy + 1, hx + 1) of the candidate of rank r, two bits each (rank 1 = intra: centre)
constexpr int kHpRank[10] = {6, 7, 8, 9, 2, 3, 4, 5, 0, 1};
constexpr uint32_t hp_rank_table(bool y, int r = 0)
{
    // candidate of rank r: 8, 9, 4, 5, 6, 7, 0, 1, 2, 3
    return r == 10 ? 0u : ((uint32_t)((r == 0 ? 8 : r == 1 ? 4 : r < 6 ? r + 2 : r - 6) / (y ? 3 : 1) % 3) << (2 * r)) | hp_rank_table(y, r + 1);
}
constexpr uint32_t kHpRankHy = hp_rank_table(true), kHpRankHx = hp_rank_table(false);
constexpr int kQuadAc0 = kQuadConst0 + 1;                            // d_ac_code2
constexpr int kQuadsPerBlock = kQuadAc0 + (2 * 2 * 33 * 41 + 1023) / 1024;
static_assert(kQuadConst0 == 11 && kQuadAc0 == 12, "k_mb's second table base points at quad 12");
__device__ u32x4_t d_lanetab[3][2][kQuadsPerBlock][64];

// CONF = option "conformant" (NOT the reference's behaviour): the reconstruction loop follows
// ISO/IEC 13818-2 where the RTL deviates from it, so that a standard decoder reproduces the encoder's reference frames
// exactly (no drift inside a GOP): four-sample average with +2, 4:2:0 chroma vector = mv / 2 toward zero, inverse
// quantiser truncating toward zero with [-2048, 2047] saturation and mismatch control, blocks that are not coded are
// not reconstructed.  The IDCT needs no change: for in-range coefficients its 18-bit row store never wraps and the
// +-255 clip gives the same pixel after the final clip to 0..255.  Checked against the oracle's conformant mode.
// EDGE = the strip's first and last macroblock row in one launch (strip mode, m2v_strip_encode): besides everything else, the
// outer 2 VL luma / VL chroma rows of the reconstruction also go to `halo_up` / `halo_down` - the buffers the neighbours
// receive - in k_halo_pack's layout, so the step needs no pack kernel between the edge rows and the send; and the window rows
// that lie in a neighbour's strip are read from `nb_up` / `nb_down`, the buffers the neighbours' rows of the reference frame
// were received in, so there is no unpack kernel (and no launch gap) between the receive and the next step either.
// PEER (with EDGE) = the peer transport's form of the same: ALL rows of the strip in one launch, edge rows first in dispatch order; the
// halo buffers are the neighbours' memory, written through and published by an arrival counter; see PeerStep (m2v_types.hpp).
// Wait (bounded) until `*seen` has reached `need`: the neighbour's edge blocks of the previous GOP steps have all delivered.  Every lane
// loads the same word (one request); system scope: the counter is written by another GPU (or another process on this one).
__device__ __forceinline__ void peer_wait(const unsigned int *seen, unsigned int need, unsigned int *gaveup, unsigned int budget, int lane)
{
    typedef __attribute__((address_space(1))) unsigned int *gu32p;
    unsigned int v = __hip_atomic_load((gu32p)seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= need) return;
    const long long t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load((gu32p)seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned int)__builtin_amdgcn_readfirstlane((int)v) >= need) return;
        // somebody of this launch has given up already (the sequence is lost anyway: nobody waits a second time), or this wait's time is up
        const unsigned int gu = __hip_atomic_load((gu32p)gaveup, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((int)gu) != 0 || wall_clock64() - t0 > (long long)budget) {
            if (lane == 0) __hip_atomic_store((gu32p)gaveup, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
}

template <int VL, bool P, bool CONF = false, bool MFMA = false, bool FILL = false, bool EDGE = false, bool PEER = false>
__global__ __launch_bounds__(64, 8) void k_mb(const FrameJob *__restrict__ jobs, const MbMap *__restrict__ mbmap,
                                           Geom g, uint32_t *__restrict__ mbinfo, MbAux *__restrict__ mbaux,
                                           uint32_t *__restrict__ slots_small, uint32_t *__restrict__ slots,
                                           int16_t *__restrict__ coef_dbg, uint8_t *__restrict__ halo_up = nullptr,
                                           uint8_t *__restrict__ halo_down = nullptr, const uint8_t *__restrict__ nb_up = nullptr,
                                           const uint8_t *__restrict__ nb_down = nullptr, PeerStep ps = PeerStep{})
{
    static_assert(!PEER || EDGE, "the peer form is a form of the edge-row kernel");
    constexpr int UR = VL, YR = 2 * VL;
    constexpr int WROWS = 16 + 2 * YR;         // luma window rows -YR .. 16+YR-1 (RTL:1446)
    constexpr int CROWS = 8 + 2 * UR;          // chroma window rows -UR .. 8+UR-1 (RTL:1447)

    // LDS map (bytes).  Region R1 holds the reference windows + current luma until the prediction is
    // formed, then the VLC symbol list; the DCT/IDCT scratch doubles as the VLC bit buffer.
    // The luma window rows are 8 dwords of data on a stride of kWS = 12 dwords: the search lanes (dy, 4-dx group) and the
    // half-pel lanes (row, 4-px group) then fall on different banks (stride 8 is 2-way conflicted: rows r and r+4 share
    // banks).  It is stored twice, the second copy shifted by one dword (B[j] = A[j+1]): v_qsad takes its 8 reference
    // bytes from an even-aligned register pair and the four pairs of a row overlap ((w0,w1) (w1,w2) (w2,w3) (w3,w4)), so
    // every lane reads two of them from the copy where they are 8-byte aligned and two from the other one - four
    // ds_read_b64 per row (2 LDS cycles each), no register shuffling.  B starts 32 banks (mod 64) after A, which keeps
    // the 32 lanes of a ds_read_b64 group on 64 different banks.  Both copies sit LAST in R1 and run over into the
    // prediction / residual / transform regions, which are first written after the window's last read (one wavefront:
    // LDS operations execute in program order), so neither the padding nor the copy costs LDS.
    constexpr int kWS = kWinStride;
    constexpr int kWinBytes = P ? WROWS * kWS * 4 : 0, kCwinBytes = P ? CROWS * 16 : 0;
    constexpr int kOffWin = 2 * kCwinBytes + 256;
    constexpr int kWinBGap = win_b_gap(WROWS);                                 // dwords from A to B: >= window, = 32 mod 64
    constexpr int kOffWinB = kOffWin + kWinBGap * 4;
    constexpr int kR1 = 1600;
    // The transform's input tiles s_cp (signed current | prediction bytes, 768 bytes) live INSIDE R1: they are written when the
    // prediction is formed - after the luma window's last read, beside the chroma windows and the current rows, which end below
    // byte 704 - and are dead before the symbol list is written.  4 288 bytes per wavefront (5 056 before round 3); eight wavefronts per SIMD -
    // the hardware's limit - with the 64 VGPRs of the P-frame kernel.
    constexpr int kOffCp = 704;
    static_assert(kOffCp == kMfmaCpOff, "MfmaLane::a1c is filled by the host");
    static_assert(2 * kCwinBytes + 256 <= kOffCp && kOffCp + 768 <= kR1, "s_cp behind the chroma windows and the current rows, inside R1");
    constexpr int kOffPred = kR1, kOffT = kOffPred + 384, kOffZig = kOffT + 6 * kTileStride * 4;
    static_assert(!P || (kOffWin % 8 == 0 && kOffWinB + kWinBytes <= kS3Scratch && kS3Scratch <= kOffZig), "window copies may run over s_pred and s_t only, and end in front of the search's scratch");
    constexpr int kLdsBytes = kOffZig + 768;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kLdsBytes];   // static base: every DS access uses an immediate offset
    uint32_t (*const s_cwin)[CROWS * 4] = (uint32_t (*)[CROWS * 4])lds;       // chroma windows: 16 bytes/row = cols 8bx-4 .. 8bx+11
    uint32_t *const s_cur = (uint32_t *)(lds + 2 * kCwinBytes);                // current luma, dword [row][4-px group]
    uint32_t *const s_win = (uint32_t *)(lds + kOffWin);                       // luma window: row stride kWS dwords, 32 bytes = frame cols 16bx-8 .. 16bx+23
    uint32_t *const s_winb = (uint32_t *)(lds + kOffWinB);                     // the same, one dword to the left
    uint32_t *const s_sym = (uint32_t *)(lds + 16);                            // VLC symbol list (<= 3 + 6 * 64 entries; [-1] is read), reuses R1
    uint8_t (*const s_pred)[64] = (uint8_t (*)[64])(lds + kOffPred);           // prediction, later reconstruction, tile layout
    uint8_t (*const s_cp)[8][16] = (uint8_t (*)[8][16])(lds + kOffCp);         // signed current | prediction bytes per tile row: the transform's input
    int32_t (*const s_t)[kTileStride] = (int32_t (*)[kTileStride])(lds + kOffT);                 // DCT phase 1; then the dequantised coefficients (as int32: the row pass of
                                                                               // the IDCT reads them without unpacking and works in place), then the bit buffer
    uint32_t *const s_bits = (uint32_t *)(lds + kOffT);                        // VLC bit segments (<= 1216 bytes), reuses s_t
    int16_t (*const s_zig)[64] = (int16_t (*)[64])(lds + kOffZig);             // quantised levels in zig-zag order

    const int lane = threadIdx.x;
    auto lds_off = [](const void *q) { return (uint32_t)(uintptr_t)(LdsU32 *)q; };     // LDS byte address
    auto lane_consts = [&]() {
        LaneK k{};
        const int r = lane >> 2, c4 = lane & 3;
        k.win_st = lds_off(&s_win[((lane >> 1) < WROWS ? lane >> 1 : WROWS - 1) * kWS + 4 * (lane & 1)]);
        k.hp = lds_off(&s_win[r * kWS + c4]);
        const int tile = ((r >> 3) << 1) | (c4 >> 1), ti = ((r & 7) << 3) | ((c4 & 1) << 2);
        k.pred_st = lds_off(&s_pred[tile][ti]);
        k.cp_st = lds_off(&s_cp[tile][r & 7][(c4 & 1) << 2]);
        k.cpc_st = lds_off(&s_cp[4][r >> 1][2 * c4]);
        const int pl = lane >> 5, yc = (lane >> 2) & 7, xc = 2 * c4;
        k.cpred_st = lds_off(&s_pred[4 + pl][(yc << 3) | xc]);
        k.cpc2_st = lds_off(&s_cp[4 + pl][yc][8 + xc]);
        k.cwin_rd = lds_off(&s_cwin[0][0] + pl * (CROWS * 4) + (yc + UR) * 4);
        {
            const int mg = lane >> 4, mc = lane & 15;
            const int rslot = (mg & 1) ? slot_of_tile(4 + (mc >> 3)) : slot_of_tile(((mc >> 3) << 1) | (mg >> 1));
            k.row_st = lds_off(&s_t[rslot][(mc & 7) << 3]);
        }
        const int di = lane >> 3;
        k.cp_rd = lds_off(&s_cp[0][di][0]);
        const int mg = lane >> 4, mc = lane & 15;
        k.a1 = lds_off(&s_cp[((mc >> 3) << 1) | (mg & 1)][mc & 7][8 * (mg >> 1)]);
        k.xrow = lds_off(&s_t[slot_of_tile(((mc >> 3) << 1) | (mg >> 1))][((mc & 7) << 3) | ((mg & 1) << 2)]);
        k.zz2 = 2u * c_zigzag[lane];
        const int ct = lane < 48 ? lane >> 3 : 5;               // column pass: 48 lanes, lane group = SLOT of the coefficient buffer
        k.col_rd = lds_off(&s_t[ct][lane & 7]);
        k.col_pred = lds_off(&s_pred[tile_of_slot(ct)][lane & 7]);
        const int pl2 = (lane >> 4) & 1, l16 = lane & 15, yc2 = l16 >> 1, half = l16 & 1;     // chroma store: 32 lanes
        k.crec_rd = lds_off(&s_pred[4 + pl2][(yc2 << 3) | (half << 2)]);
        k.crec_r = (uint32_t)yc2; k.crec_c = (uint32_t)(4 * half); k.crec_p = (uint32_t)pl2;
        return k;
    };
    if constexpr (FILL) {
        const LaneK k = lane_consts();
        for (int q = 0; q < kQuadsLaneK; ++q) d_lanetab[VL - 1][P ? 1 : 0][q][lane] = ((const u32x4_t *)&k)[q];
        return;
    }
    // quad-major tables: ONE scalar base (pinned: the compiler would re-derive the symbol's address with s_getpc at every use),
    // 1 KB per quad as the load's immediate, 16 * lane in a register
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // is this block one of the strip's edge rows?  (EDGE alone: every block of the launch is; PEER: the first n_edge; wave-uniform)
    bool edge_blk = EDGE;
    if constexpr (PEER) edge_blk = sgpr((int)((blockIdx.x - ps.n_edge) >> 31)) != 0;
    typedef const __attribute__((address_space(1))) u32x4_t *gld128;
    const uint8_t *ltab = (const uint8_t *)&d_lanetab[VL - 1][P ? 1 : 0][4][0];
    asm volatile("" : "+s"(ltab));
    const uint8_t *ltab2 = ltab + 8 * 1024;       // quads 8 .. (their offsets do not fit the immediate: without a base of their own the
    asm volatile("" : "+s"(ltab2));               // compiler forms 64-bit vector addresses)
#define M2V_QUAD(q0, T, member) ((q0) + (int)(offsetof(T, member) / 16) < 8                                              \
                                     ? *(gld128)(ltab + ((q0) + (int)(offsetof(T, member) / 16) - 4) * 1024 + lane16)  \
                                     : *(gld128)(ltab2 + ((q0) + (int)(offsetof(T, member) / 16) - 12) * 1024 + lane16))
#define M2V_LANEK4(member) M2V_QUAD(0, LaneK, member)
    // Requested with the pixels (whose own addresses stay arithmetic: a table value in front of them would put a second memory
    // round trip before the first load); the last two quads follow before the transform.
    // An I frame (no window, no search to hide behind) requests everything here, and of the first quad only the two words it uses.
    u32x4_t kq0 = {0, 0, 0, 0};
    if constexpr (P) kq0 = M2V_LANEK4(win_st);
    else {
        // (M2V_LANEK4 names the quad a member lies in: the member's own offset inside the quad is added here)
        const u32x2_t t = *(const __attribute__((address_space(1))) u32x2_t *)((const uint8_t *)&M2V_LANEK4(pred_st) + offsetof(LaneK, pred_st) % 16);
        kq0.z = t.x; kq0.w = t.y;
    }
    const u32x4_t kq1 = M2V_LANEK4(cpc_st);
    constexpr bool kMfmaLuma = MFMA && !CONF;   // the transform of the four luma tiles on the matrix cores ...
    // ... and of the two chroma tiles as well, in I frames only: same box, P kernel 0.904 -> 0.924 ms per sequence with it (twelve vector
    // instructions more on the unit that is 89 % busy there outweigh three loads and eighteen LDS instructions less), I kernel of config c2
    // 0.2135 -> 0.2059 ms (profiles/r04_experiments.txt item 15)
    constexpr bool kMfmaChroma = kMfmaLuma && !P;
    long mf_b1 = 0, mf_a2lo = 0, mf_a2hi = 0;   // matrix-core operands of the lane (c_mfma), requested with group 3
    uint32_t mf_a1c = 0;
    u32x4_t mf_zoff = {0, 0, 0, 0};
    u32x4_t kq2, kq3;                           // the lane table's last three quads (not before the search: registers)
    u32x3_t kq4 = {0, 0, 0};
#define M2V_REQUEST_G3()                                                                                             \
    do {                                                                                                             \
        kq2 = M2V_LANEK4(row_st); kq3 = M2V_LANEK4(zz2);                                                                \
        /* only the words that are used: a dead register of a wide load is reused at once, and the write-after-write   \
           wait then stalls the wavefront for the whole round trip */                                                  \
        if constexpr (EDGE) { if (edge_blk) kq4 = *(const __attribute__((address_space(1))) u32x3_t *)&M2V_LANEK4(crec_r); }   /* halo rows only */ \
        if constexpr (MFMA && !CONF) {                                                                               \
            /* ONE load for the three words (a vector memory instruction costs what ten arithmetic ones do) */           \
            /* ONE load for the three / four words (only the words that are used, see above) */                         \
            u32x4_t m0 = {0, 0, 0, 0};                                                                                \
            if constexpr (kMfmaChroma) m0 = M2V_QUAD(kQuadMfma0, MfmaLane, b1[0]);                                    \
            else { const u32x3_t m3 = *(const __attribute__((address_space(1))) u32x3_t *)&M2V_QUAD(kQuadMfma0, MfmaLane, b1[0]); m0.x = m3.x; m0.y = m3.y; m0.z = m3.z; } \
            mf_b1 = (long)(((unsigned long long)m0.y << 32) | m0.x);                                                 \
            mf_a2lo = (long)(unsigned long long)m0.z; mf_a2hi = (long)((unsigned long long)m0.z << 32);              \
            mf_a1c = m0.w;                                                                                           \
        }                                                                                                            \
    } while (0)
    if constexpr (!P) M2V_REQUEST_G3();
    // the intra quantiser's lane constants (weights, reciprocals): two loads; an I frame asks up front, a P frame inside its (rare) intra branch
    u32x4_t iq_recip = {0, 0, 0, 0}, iq_w = {0, 0, 0, 0};
#define M2V_REQUEST_INTRA() do { iq_recip = M2V_QUAD(kQuadMfma0, MfmaLane, irecip[0]); iq_w = M2V_QUAD(kQuadSearch0, SearchLane, dead_hi); } while (0)
    if constexpr (!P) M2V_REQUEST_INTRA();
    // the transform's basis rows: basis row i = lane >> 3 widened to int32, basis row j = lane & 7 and its negative as int8 x 8
    const int dj = lane & 7;
    const int di = lane >> 3;
    int bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint2 mj = {0, 0}, nj = {0, 0};
#define M2V_REQUEST_BASIS()                                                                                                             \
    do {                                                                                                                                \
        _Pragma("unroll") for (int k = 0; k < 8; ++k)                                                                                   \
            bi[k] = *(const __attribute__((address_space(1))) int32_t *)(ltab2 - 1024 + kConstDct32 + (uint32_t)(di * 32) + 4 * k);    \
        const u32x4_t mnv = *(gld128)(ltab2 - 1024 + kConstDct + (uint32_t)(dj * 16));     /* row j | minus row j */                    \
        mj = uint2{mnv.x, mnv.y}; nj = uint2{mnv.z, mnv.w};                                                                             \
    } while (0)
    if constexpr (!P && !kMfmaChroma) M2V_REQUEST_BASIS();     // (the matrix-core transform needs none of them)
    // grid = (macroblocks of one frame's share, frames of the launch list): the frame comes from blockIdx.y, no division for it
    const uint32_t li = blockIdx.y;                                            // which frame of the launch list
    // block -> macroblock: the host's table for this launch shape (MbMap: the XCD-aware permutation, and for the strip kernels the edge
    // rows first), one scalar load beside the job's
    const MbMap me = mbmap[blockIdx.x];
    const FrameJob job = jobs[li];                 // `jobs` = the launch list as jobs: one dependent scalar load, not list -> job
    const int fidx = (int)job.fidx;
    const int mb = (int)(me.mb & 0xFFFFFFu), by = (int)(me.byx >> 16), bx = (int)(me.byx & 0xFFFFu);
    const int W = g.W;
    const uint32_t tile = (uint32_t)mb + (uint32_t)by;           // by * (mbw + 1) + bx: the reconstruction tile (luma and chroma) whose right half this macroblock fills
    const int r = lane >> 2, c4 = lane & 3;
    // 1 = the macroblock has a neighbour on that side (left, right, up, down), from the host's table.  Integers, not compares: these
    // wave-uniform flags feed range limits and candidate masks that must stay on the scalar unit, and a compare that is
    // also needed as a lane mask somewhere gets computed by the vector ALU - with everything downstream of it.
    const int in_l = sgpr((int)((me.mb >> 24) & 1u)), in_r = sgpr((int)((me.mb >> 25) & 1u));
    const int in_u = sgpr((int)((me.mb >> 26) & 1u)), in_d = sgpr((int)((me.mb >> 27) & 1u));

    // ---- stages A..E: current macroblock; 4:4:4 -> 4:2:0 with two-stage rounding -------------
    // (RTL:1086-1089 horizontal mean2, RTL:1167-1170 vertical mean2 of the two means)
    // Every global load of the macroblock (3 current rows, 2 luma window passes, the chroma windows) is issued
    // back to back through explicit global-address-space pointers and waited for once: the wavefront pays ONE
    // memory round trip.  Window samples outside the frame can never be selected (RTL:1642-1645), so their
    // addresses are clamped into the frame instead of being branched around.
    // The windows arrive in ONE load each for an interior macroblock - 95 % of a frame -: 16 bytes per lane for the luma window (lane =
    // row, tile column), 8 bytes per lane for both chroma windows (lane = plane, row, tile column); the shifted tiles of the reconstruction
    // (rec_luma_off) make every window row two full tile rows.  Vector memory instructions are the second thing this kernel is sensitive
    // to - one costs what ten arithmetic instructions do (profiles/r04_experiments.txt item 14).  At the frame border, where every dword
    // is clamped separately, it is four loads + two.
    typedef const __attribute__((address_space(1))) uint32_t *gld32;
    const uint8_t *inY = job.in, *inU = inY + g.ysz, *inV = inU + g.ysz;
    const uint32_t pix_off = __umul24((uint32_t)(16 * by + r), (uint32_t)W) + (uint32_t)(16 * bx + 4 * c4);   // rows, W < 2^12
    uint32_t cur4 = *(gld32)(inY + pix_off);
    uint32_t u4 = *(gld32)(inU + pix_off);
    uint32_t v4 = *(gld32)(inV + pix_off);
    typedef const __attribute__((address_space(1))) u32x2_t *gld64;
    // the windows: lane = (window row, half) takes the 16 bytes of one luma tile row (2 WROWS lanes), lane = (plane, row, half) the 8 bytes
    // of one chroma tile row
    u32x4_t wwin = {0, 0, 0, 0};
    u32x2_t wc = {0, 0};
    const int wrow = (lane >> 1) < WROWS ? lane >> 1 : WROWS - 1, whalf = lane & 1;
    const int cpl = lane >> 5, crow0 = (lane & 31) >> 1, chalf = lane & 1;       // chroma windows: plane, row, half
    const int crow = crow0 < CROWS ? crow0 : CROWS - 1;
    SearchLane sl{};                             // the lane's search addresses: loaded with the pixels, one memory round trip
    if constexpr (P && VL == 3) {
        const u32x4_t s0 = M2V_QUAD(kQuadSearch0, SearchLane, even), s1 = M2V_QUAD(kQuadSearch0, SearchLane, plus);
        // one dword of the third quad: a dwordx4 load whose other three registers are dead gets them reused right away, and the
        // compiler then waits for the load (write-after-write) before it issues the window loads - a second round trip
        const uint32_t s2 = *(const __attribute__((address_space(1))) uint32_t *)(ltab + (kQuadSearch0 + (int)(offsetof(SearchLane, dead_hi) / 16) - 4) * 1024 + lane16);
        sl.even = s0.x; sl.odd = s0.y; sl.cur = s0.z; sl.cur12 = s0.w;
        sl.plus = s1.x; sl.minus = s1.y; sl.cb4 = s1.z; sl.dead_lo = s1.w; sl.dead_hi = s2;
    }
    if constexpr (P) {
        const uint8_t *refY = job.ref;              // tiled (rec_luma_off / rec_chroma_off)
        const uint32_t trow = (uint32_t)g.mbw + 1u;              // tiles per tile row
        if (EDGE && edge_blk) {
            // window rows above the strip's first / below its last macroblock row belong to a neighbour: they were received, for
            // this frame's reference, at position rhidx of nb_up / nb_down ([YR rows of W luma][UR rows of cw U][UR of V] per
            // frame, rows top to bottom); everything else as the clamped form below
            const bool ext_u = nb_up != nullptr && by == g.edge_top, ext_d = nb_down != nullptr && by == g.edge_bot;     // wave-uniform
            // peer transport: the neighbour's rows were written into this rank's memory by ANOTHER GPU's kernel (or another process's on
            // this one) - wait for its arrival counter, then read with loads that no cache of this GPU can answer from an older copy
            // (system-scope relaxed atomic loads = global_load .. sc0 sc1; the whole window of such a block, six loads per lane)
            bool far = false;
            if constexpr (PEER) {
                // (-DM2V_DEBUG, option ablate bits 22-25: the hand-off's parts switched off one by one, to see what each costs - results invalid)
                if (!(kDebug && (g.ablate & (1 << 24)))) {
                    // every block of the neighbour's edge row, for every earlier frame of this GOP (they all are referenced: each has delivered)
                    const unsigned int need = (unsigned int)job.i_frame * (unsigned int)g.mbw;
                    const uint32_t slot = (uint32_t)job.rhidx * (uint32_t)kPeerCntStride;
                    if (ext_u) peer_wait(ps.seen_up + slot, need, ps.gaveup, ps.budget, lane);
                    if (ext_d) peer_wait(ps.seen_down + slot, need, ps.gaveup, ps.budget, lane);
                }
                far = (ext_u || ext_d) && !(kDebug && (g.ablate & (1 << 25)));
            }
            auto ld = [&](const uint8_t *base, uint32_t off) -> uint32_t {
                typedef __attribute__((address_space(1))) unsigned int *gu32p;
                if (PEER && far) return __hip_atomic_load((gu32p)(base + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return *(const uint32_t *)(base + off);
            };
            const uint32_t chunk = (uint32_t)(YR + UR) * (uint32_t)W, fb = (uint32_t)job.rhidx * chunk;
            {
                int yy = 16 * by - YR + wrow;
                yy = yy < 0 ? 0 : yy > g.H - 1 ? g.H - 1 : yy;
                const bool up = ext_u && wrow < YR, dn = ext_d && wrow >= YR + 16;
                const uint8_t *src = up ? nb_up : dn ? nb_down : refY;
                const uint32_t rowbase = fb + (uint32_t)(up ? wrow : wrow - (YR + 16)) * (uint32_t)W;      // (the halo buffers hold plain rows)
                uint32_t w4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int xk = 16 * bx - 8 + 16 * whalf + 4 * k;
                    xk = xk < 0 ? 0 : xk > W - 4 ? W - 4 : xk;
                    const uint32_t o = (up || dn) ? rowbase + (uint32_t)xk : rec_luma_off((uint32_t)xk, (uint32_t)yy, g);
                    w4[k] = ld(src, o);
                }
                wwin = u32x4_t{w4[0], w4[1], w4[2], w4[3]};
            }
            int yy = 8 * by - UR + crow, x0 = 8 * bx - 4 + 8 * chalf, x1 = x0 + 4;
            yy = yy < 0 ? 0 : yy > g.ch - 1 ? g.ch - 1 : yy;
            x0 = x0 < 0 ? 0 : x0 > g.cw - 4 ? g.cw - 4 : x0;
            x1 = x1 < 0 ? 0 : x1 > g.cw - 4 ? g.cw - 4 : x1;
            const uint8_t *sc = refY;
            uint32_t c0o = rec_chroma_off((uint32_t)cpl, (uint32_t)x0, (uint32_t)yy, g), c1o = rec_chroma_off((uint32_t)cpl, (uint32_t)x1, (uint32_t)yy, g);
            const uint32_t cbase = fb + (uint32_t)YR * (uint32_t)W + __umul24((uint32_t)cpl, (uint32_t)UR * (uint32_t)g.cw);   // V sits UR rows behind U in a halo chunk
            if (ext_u && crow < UR) { sc = nb_up; c0o = cbase + (uint32_t)crow * (uint32_t)g.cw + (uint32_t)x0; c1o = c0o - (uint32_t)x0 + (uint32_t)x1; }
            if (ext_d && crow >= UR + 8) { sc = nb_down; c0o = cbase + (uint32_t)(crow - (UR + 8)) * (uint32_t)g.cw + (uint32_t)x0; c1o = c0o - (uint32_t)x0 + (uint32_t)x1; }
            wc.x = ld(sc, c0o);
            wc.y = ld(sc, c1o);
        } else
        if (sgpr(in_l & in_r & in_u & in_d)) {
            // interior macroblock (wave-uniform test): the whole window lies inside the frame, no clamping: window row yrel = -YR .. 15 + YR
            // is row yrel & 15 of tile row by + (yrel >> 4), tile columns bx (window columns -8 .. 7) and bx + 1 (8 .. 23); ONE load each for
            // the luma window and for both chroma windows
            // (row + 16 and row + 8 keep the tile-row factor non-negative: a 24-bit multiply, not the quarter-rate 32-bit one; every lane
            // loads - lanes >= 2 WROWS fetch the last row again, nobody stores it)
            const uint32_t yr16 = (uint32_t)(wrow - YR + 16), cr8 = (uint32_t)(crow - UR + 8);
            typedef const __attribute__((address_space(1))) u32x4_t *gld128w;
            wwin = *(gld128w)(refY + (__umul24(yr16 >> 4, trow * 256u) + ((yr16 & 15u) << 4) + ((tile + (uint32_t)whalf - trow) * 256u)));
            wc = *(gld64)(refY + (__umul24(cr8 >> 3, trow * 128u) + ((cr8 & 7u) << 3) + ((uint32_t)cpl << 6) + (g.rysz + (tile + (uint32_t)chalf - trow) * 128u)));
        } else {
            {
                int yy = 16 * by - YR + wrow;
                yy = yy < 0 ? 0 : yy > g.H - 1 ? g.H - 1 : yy;
                uint32_t w4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int xk = 16 * bx - 8 + 16 * whalf + 4 * k;
                    xk = xk < 0 ? 0 : xk > W - 4 ? W - 4 : xk;
                    w4[k] = *(gld32)(refY + rec_luma_off((uint32_t)xk, (uint32_t)yy, g));      // 32-bit offsets from a uniform base
                }
                wwin = u32x4_t{w4[0], w4[1], w4[2], w4[3]};
            }
            int yy = 8 * by - UR + crow, x0 = 8 * bx - 4 + 8 * chalf, x1 = x0 + 4;
            yy = yy < 0 ? 0 : yy > g.ch - 1 ? g.ch - 1 : yy;
            x0 = x0 < 0 ? 0 : x0 > g.cw - 4 ? g.cw - 4 : x0;
            x1 = x1 < 0 ? 0 : x1 > g.cw - 4 ? g.cw - 4 : x1;
            wc.x = *(gld32)(refY + rec_chroma_off((uint32_t)cpl, (uint32_t)x0, (uint32_t)yy, g));
            wc.y = *(gld32)(refY + rec_chroma_off((uint32_t)cpl, (uint32_t)x1, (uint32_t)yy, g));
        }
    }
    if (sgpr((int)((job.valid_beats - (g.ysz >> 2)) >> 31))) {    // a frame cut short by i_sequence_stop: wave-uniform, almost never
        // the empty asm keeps this a scalar branch (flattened, it is a compare and three selects in every macroblock)
        asm volatile("");
        if ((pix_off >> 2) >= job.valid_beats) {                 // beats after the stop are black (RTL:1036-1056)
            cur4 = 0u; u4 = 0x80808080u; v4 = 0x80808080u;
        }
    }
    s_cur[lane] = cur4;
    if constexpr (P && VL == 3) {               // full-pel search with helper lanes: their copy of rows 13..15, the zeroed sum cells
        static_assert(2 * kCwinBytes == kS3Cur && kOffWin == kS3Win && kOffT + 6 * 64 * 4 == kS3Scratch && kS3Zero + 8 <= kLdsBytes, "c_search holds these offsets");
        if (lane >= 52) {
            uint32_t *const rep = (uint32_t *)(lds + kS3Rep) + (lane - 52);
            rep[0] = cur4; rep[12] = cur4; rep[24] = cur4; rep[36] = cur4;
        }
        if (lane < 10) ((uint32_t *)(lds + kS3Sum12))[lane] = 0u;
    }
    uint32_t cuv;                               // this lane's two 4:2:0 samples of U (bytes 0, 1) and of V (bytes 2, 3); even rows only
    {
        const uint32_t hu = avg2x4(u4, u4 >> 8);          // bytes 0 and 2: horizontal means of the two pixel pairs
        const uint32_t hv = avg2x4(v4, v4 >> 8);
        const uint32_t hu_p = lane_plus4(hu);                           // the odd row below (only even rows use the result)
        const uint32_t hv_p = lane_plus4(hv);
        const uint32_t cu = avg2x4(hu, hu_p), cv = avg2x4(hv, hv_p);
        cuv = __builtin_amdgcn_perm(cv, cu, 0x06040200u);
    }

    int inter = 0, mvx = 0, mvy = 0;
    uint32_t pred4 = 0x80808080u;               // intra prediction (RTL:1894-1903)

    if constexpr (P) {
        // ---- stages X..Z: reference window of recon(f-1) into LDS (RTL:1350-1425, 1612-1629) --
        typedef __attribute__((address_space(3))) u32x2_t *LdsW64;
        if (lane < 2 * WROWS) {
            // s_win[row * kWS + 4 half .. + 3] (16-byte aligned: one ds_write_b128) and the same four elements of copy B, one dword to the
            // left (column 0 lands in padding): the lane's address from the table, everything else an immediate
            typedef __attribute__((address_space(3))) uint32_t *LdsW;
            typedef __attribute__((address_space(3))) u32x4_t *LdsW128;
            *(LdsW128)(uintptr_t)kq0.x = wwin;
            *(LdsW)(uintptr_t)(kq0.x + (uint32_t)(kWinBGap * 4 - 4)) = wwin.x;
            *(LdsW64)(uintptr_t)(kq0.x + (uint32_t)(kWinBGap * 4)) = u32x2_t{wwin.y, wwin.z};
            *(LdsW)(uintptr_t)(kq0.x + (uint32_t)(kWinBGap * 4 + 8)) = wwin.w;
        }
        if (crow0 < CROWS)                      // s_cwin[plane][row * 4 + 2 * half]: (lane & 31) * 8 bytes into the plane's window
            *(LdsW64)(uintptr_t)(lds_off(&s_cwin[0][0]) + (uint32_t)cpl * (uint32_t)kCwinBytes + (uint32_t)(lane & 31) * 8u) = wc;
        M2V_WAVE_SYNC();
        M2V_STOP(1);        // loads, chroma subsampling, window staging

        // ---- full-pel search: (2YR+1)^2 SADs (RTL:1634-1715) ---------------------------------
        // lane = (dy, group of 4 consecutive dx); v_qsad_pk_u16_u8 slides the 4 current pixels
        // over 8 reference bytes and accumulates the 4 SADs as packed u16.
        int fy = 0, fx = 0;
        // the pixel sum of the intra cost (half-pel phase below) is formed here: its reduction chain then runs beside the search's
        // LDS reads instead of at the head of the half-pel phase's dependency chain
        const uint32_t S = (uint32_t)sgpr(wave_sum((int)__builtin_amdgcn_sad_u8(cur4, 0u, 0u)));
        {
            uint32_t key = 0xFFFFFFFFu;
            // dy = dyi - YR, dx = 4*gq - 8 + j
            // VECTOR_LEVEL 1 / 2: kPairs (dy, group) pairs - only the groups that hold a live dx: 1 and 2 (dx -4 .. 3) for +-2, 1 .. 3 for +-4 -, each
            // given to kParts lanes kLanesPerPart apart, lane (part, pair) running rows [part * kNR, + kNR).  The lanes beyond the last pair
            // repeat its work (their loads stay inside the window) and own nothing.
            constexpr int kParts = VL == 1 ? 4 : VL == 2 ? 2 : 1, kLanesPerPart = 64 / kParts, kPairs = VL == 1 ? 10 : VL == 2 ? 27 : 64, kNR = 16 / kParts;
            const int part = VL == 3 ? 0 : lane / kLanesPerPart, pair0 = lane & (kLanesPerPart - 1), pair = pair0 < kPairs ? pair0 : kPairs - 1;
            const int dyi = VL == 3 ? s3_dy(lane) : VL == 1 ? pair >> 1 : (pair * 11) >> 5;
            const int gq = VL == 3 ? s3_group(lane) : VL == 1 ? 1 + (pair & 1) : 1 + pair - 3 * dyi;
            const bool owner = VL == 3 || (part == 0 && pair0 < kPairs);
            const uint32_t sl_cb4 = sl.cb4, sl_dead_lo = sl.dead_lo, sl_dead_hi = sl.dead_hi;
            if (!(kDebug && (g.ablate & 1))) {
                unsigned long long acc = 0;
                QsadRow ra, rb{};
                if constexpr (VL == 3) {
                    // all 64 lanes, 13 steps (see kS3Cur); every LDS address of the lane comes from the lane table
                    const uint32_t lds0 = (uint32_t)(uintptr_t)(LdsU32 *)lds;      // 0: the block's only LDS object
                    const uint32_t ae = lds0 + sl.even, ao = lds0 + sl.odd;
                    const unsigned long long hmask = kS3Helpers;
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %4 offset:8\n\tds_read_b64 %3, %5 offset:8"
                                 : "=&v"(ra.w01), "=&v"(ra.w12), "=&v"(ra.w23), "=&v"(ra.w34) : "v"(ae), "v"(ao));
                    search_rows13<0>(lds0 + sl.cur, lds0 + sl.cur12, ae, ao, lds0 + sl.plus, hmask, acc, ra, rb);
                    asm volatile("s_mov_b64 exec, %2\n\tds_add_u64 %1, %0\n\ts_mov_b64 exec, -1"
                                 : : "v"(acc), "v"(lds0 + sl.minus), "s"(hmask) : "memory");
                    // owner lanes: + rows 13..15 = a difference of two running sums (no field borrows or carries: the sums grow
                    // monotonically and a complete SAD is at most 65280); the helper lanes compute garbage and are dead below
                    const u32x2_t sp = *(LdsU2)(uintptr_t)(lds0 + sl.plus), sm = *(LdsU2)(uintptr_t)(lds0 + sl.minus);
                    const uint32_t lo = (uint32_t)acc + (sp.x - sm.x), hi = (uint32_t)(acc >> 32) + (sp.y - sm.y);
                    acc = ((unsigned long long)hi << 32) | lo;
                } else {
                    // the pairs (w0,w1) (w2,w3) start at dword gq, the pairs (w1,w2) (w3,w4) at gq + 1: one of the two is even
                    // in copy A, the other one in copy B (which holds dword j + 1 at index j)
                    // hand-issued ds_read_b64, one row ahead (left to itself the compiler fuses them into ds_read2_b64,
                    // which runs at half the LDS rate - MI355X_MICROARCH.md, LDS table)
                    const int wrow = dyi + kNR * part;                 // the window row of the part's first macroblock row
                    const uint32_t *const pe = (gq & 1) ? s_winb + wrow * kWS + gq - 1 : s_win + wrow * kWS + gq;
                    const uint32_t *const po = (gq & 1) ? s_win + wrow * kWS + gq + 1 : s_winb + wrow * kWS + gq;
                    const uint32_t ae = (uint32_t)(uintptr_t)(LdsU32 *)pe, ao = (uint32_t)(uintptr_t)(LdsU32 *)po;
                    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %4 offset:8\n\tds_read_b64 %3, %5 offset:8"
                                 : "=&v"(ra.w01), "=&v"(ra.w12), "=&v"(ra.w23), "=&v"(ra.w34) : "v"(ae), "v"(ao));
                    search_rows_part<0, kNR, kWS>((uint32_t)(uintptr_t)(LdsU32 *)s_cur + (uint32_t)(part * kNR * 16), ae, ao, acc, ra, rb);
                    // the parts' sums of a pair sit kLanesPerPart lanes apart: folded with the lane swaps (two rows of 32, then two rows of 16).
                    // No field carries into its neighbour: a complete SAD is at most 65280.
                    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
                    {
                        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
                        lo = a[0] + a[1]; hi = b[0] + b[1];
                    }
                    if constexpr (kParts == 4) {
                        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
                        lo = a[0] + a[1]; hi = b[0] + b[1];
                    }
                    acc = ((unsigned long long)hi << 32) | lo;
                }
                // minimum SAD; among equals the largest dy, then the largest dx (RTL:1694-1710): key = sad << 8 | (255 - index),
                // index = dy' << 4 | dx + 8.  A SAD >= 4096 kills a candidate (RTL:1669-1670): such keys are >= 1 << 20 and lose
                // against every live one, so the test is made once on the reduced key instead of per candidate.
                // index has its two low bits clear, so 255 - index ends in 11 and candidate j's key is (sad_j << 8 | cbase) - j.
                const uint32_t cbase = 255u - (uint32_t)((dyi << 4) | (4 * gq));
                if (YR == 6 && sgpr(in_l & in_r & in_u & in_d)) {
                    // VECTOR_LEVEL 3, a macroblock with all four neighbours (wave-uniform, 95 % of a frame): every dy is live and
                    // the dead dx are the three slots beyond +-6 (dx = -8, -7 in group 0, dx = 7 in group 3): a dead slot gets SAD 0xFFFF
                    // (the lane table holds those bits and the four position bytes cbase - j); one v_perm per key: [0, sad_hi, sad_lo, pos]
                    const uint32_t l32 = (uint32_t)acc | sl_dead_lo, h32 = (uint32_t)(acc >> 32) | sl_dead_hi;
                    const uint32_t k0 = __builtin_amdgcn_perm(l32, sl_cb4, 0x0C050400u), k1 = __builtin_amdgcn_perm(l32, sl_cb4, 0x0C070601u);
                    const uint32_t k2 = __builtin_amdgcn_perm(h32, sl_cb4, 0x0C050402u), k3 = __builtin_amdgcn_perm(h32, sl_cb4, 0x0C070603u);
                    key = umin32(umin32(k0, k1), umin32(k2, k3));
                } else {
                    // live dy / dx range at the frame border (RTL:1642-1645), wave-uniform and kept on the scalar unit: one
                    // unsigned range compare per axis per candidate
                    const int lo = -YR & -in_l, hi = YR & -in_r, ylo = -YR & -in_u, yhi = YR & -in_d;
                    const uint32_t span = (uint32_t)(hi - lo), yspan = (uint32_t)(yhi - ylo);
                    const bool rowok = (uint32_t)(dyi - YR - ylo) <= yspan;
                    const int d0 = 4 * gq - 8 - lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t sad = (uint32_t)(acc >> (16 * j)) & 0xFFFFu;
                        const uint32_t k = (sad << 8) | (cbase - (uint32_t)j);
                        const bool ok = owner && rowok && (uint32_t)(d0 + j) <= span;
                        if (ok && k < key) key = k;
                    }
                }
            }
            key = wave_min_u32(key);
            key = (uint32_t)sgpr(uniform((int)key));
            if (key < (4096u << 8)) {           // no live candidate: (0,0) (RTL:1695, 1707)
                const int c = 255 - (int)(key & 255u);
                fy = (c >> 4) - YR;
                fx = (c & 15) - 8;
            }
            fy = sgpr(fy);
            fx = sgpr(fx);
        }

        M2V_STOP(2);        // ... up to the full-pel search
        if (kDebug && ((g.ablate >> 16) & 15)) {
            // -DM2V_DEBUG, option ablate bits 16-19: 64 extra INDEPENDENT vector instructions of one kind per macroblock - what does an
            // instruction of that kind cost this kernel? (tools/valu_kind.sh; results unaffected: the values are discarded)
            const int kind = sgpr((g.ablate >> 16) & 15);
            uint32_t d0, d1, d2, d3;
            const uint32_t a = (uint32_t)lane, b = cur4;
#define M2V_PAD4(INS) INS(%0) INS(%1) INS(%2) INS(%3)
#define M2V_PAD64(INS) asm volatile(M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) \
                                    M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) M2V_PAD4(INS) \
                                    : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a), "v"(b), "s"(kind))
#define M2V_I_ADD(D)   "v_add_u32 " #D ", %4, %5\n\t"
#define M2V_I_LSHL(D)  "v_lshlrev_b32 " #D ", 3, %4\n\t"
#define M2V_I_MAD(D)   "v_mad_i32_i24 " #D ", %4, %5, %4\n\t"
#define M2V_I_ADDS(D)  "v_add_u32 " #D ", %6, %4\n\t"
#define M2V_I_ASHR(D)  "v_ashrrev_i32 " #D ", 3, %4\n\t"
#define M2V_I_PERM(D)  "v_perm_b32 " #D ", %4, %5, %4\n\t"
            if (kind == 7) {
                // 16 LDS reads of 8 bytes per lane (64 LDS data cycles: + 5 % of the macroblock's), the lane's own window words
                unsigned long long t0, t1, t2, t3;
                asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:48\n\tds_read_b64 %2, %4 offset:96\n\tds_read_b64 %3, %4 offset:144\n\t"
                             "ds_read_b64 %0, %4 offset:192\n\tds_read_b64 %1, %4 offset:240\n\tds_read_b64 %2, %4 offset:288\n\tds_read_b64 %3, %4 offset:336\n\t"
                             "ds_read_b64 %0, %4 offset:8\n\tds_read_b64 %1, %4 offset:56\n\tds_read_b64 %2, %4 offset:104\n\tds_read_b64 %3, %4 offset:152\n\t"
                             "ds_read_b64 %0, %4 offset:200\n\tds_read_b64 %1, %4 offset:248\n\tds_read_b64 %2, %4 offset:296\n\tds_read_b64 %3, %4 offset:344\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(kq0.x) : "memory");
            } else if (kind == 8) {
                // 8 vector loads of one dword per lane from the lane table (L1 / L2 hits): + 24 % vector memory instructions
                asm volatile("global_load_dword %0, %4, %5 offset:-4096\n\tglobal_load_dword %1, %4, %5 offset:-3072\n\t"
                             "global_load_dword %2, %4, %5 offset:-2048\n\tglobal_load_dword %3, %4, %5 offset:-1024\n\t"
                             "global_load_dword %0, %4, %5 offset:-4092\n\tglobal_load_dword %1, %4, %5 offset:-3068\n\t"
                             "global_load_dword %2, %4, %5 offset:-2044\n\tglobal_load_dword %3, %4, %5 offset:-1020\n\t"
                             "s_waitcnt vmcnt(0)" : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(lane16), "s"(ltab) : "memory");
            } else if (kind == 9) {
                // ONE vector load and the wait for it: what an exposed memory round trip costs
                asm volatile("global_load_dword %0, %1, %2 offset:-4096\n\ts_waitcnt vmcnt(0)" : "=&v"(d0) : "v"(lane16), "s"(ltab) : "memory");
            } else if (kind == 11) {
                // 64 scalar ALU instructions (four independent chains)
                int a0 = sgpr(kind), a1 = a0, a2 = a0, a3 = a0;
#define M2V_S4 "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                asm volatile(M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4 M2V_S4
                             : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3) : : "scc");
#undef M2V_S4
            } else if (kind == 10) {
                // ONE LDS read and the wait for it
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d0) : "v"(kq0.x) : "memory");
            } else
            if (kind == 1) M2V_PAD64(M2V_I_ADD);
            else if (kind == 2) M2V_PAD64(M2V_I_LSHL);
            else if (kind == 3) M2V_PAD64(M2V_I_MAD);
            else if (kind == 4) M2V_PAD64(M2V_I_ADDS);
            else if (kind == 5) M2V_PAD64(M2V_I_ASHR);
            else M2V_PAD64(M2V_I_PERM);
#undef M2V_PAD4
#undef M2V_PAD64
        }
        // ---- half-pel refinement + intra cost (RTL:1743-1816), four pixels per lane, packed bytes ----
        // T[y][x] = window[y+fy+YR][x+fx+8]; L/C/R = T[.][x-1 .. x+2], T[.][x .. x+3], T[.][x+1 .. x+4]
        uint32_t L0, C0, R0, L1, C1, R1, L2, C2, R2;
        {
            // window row r + fy + YR, first byte 4 c4 + 7 + fx: the dword index splits into a lane part (r, c4) and a
            // wave-uniform part (fy, fx), and the byte shift (7 + fx) & 3 is wave-uniform: one address add per lane, every
            // other offset is an immediate.  Row -1 (r = 0, fy = -YR) and row WROWS (r = 15, fy = YR) lie outside the window
            // but inside this kernel's LDS; they only feed half-pel candidates that are dead in exactly those cases
            // (RTL:1757-1760), and so does the third dword of a row when it is the padding dword 8.
            const int kx = 7 + fx;                             // 1 .. 13
            const uint32_t sft = (uint32_t)kx & 3u;
            // based at the row ABOVE (all three rows at non-negative immediate offsets: a negative one costs an address add per read)
            const uint32_t *const pw = (const uint32_t *)(LdsU32 *)(uintptr_t)(kq0.y + 4u * (uint32_t)sgpr((fy + YR - 1) * kWS + (kx >> 2)));
#define M2V_ROW3(OFF, L, C, R)                                                                  \
            {                                                                                   \
                const uint32_t a0 = pw[(OFF)], a1 = pw[(OFF) + 1], a2 = pw[(OFF) + 2];          \
                const uint32_t lo = __builtin_amdgcn_alignbyte(a1, a0, sft);                    \
                const uint32_t hi = __builtin_amdgcn_alignbyte(a2, a1, sft);                    \
                L = lo;                                                                         \
                C = __builtin_amdgcn_alignbyte(hi, lo, 1u);                                     \
                R = __builtin_amdgcn_alignbyte(hi, lo, 2u);                                     \
            }
            M2V_ROW3(0, L0, C0, R0)
            M2V_ROW3(kWS, L1, C1, R1)
            M2V_ROW3(2 * kWS, L2, C2, R2)
#undef M2V_ROW3
        }
        uint32_t hp[9];                                         // the nine half-pel predictions (RTL:1746-1752)
        hp[0] = avg4<CONF>(L0, C0, L1, C1);  hp[1] = avg2x4(C0, C1);  hp[2] = avg4<CONF>(C0, R0, C1, R1);
        hp[3] = avg2x4(L1, C1);          hp[4] = C1;              hp[5] = avg2x4(C1, R1);
        hp[6] = avg4<CONF>(L1, C1, L2, C2);  hp[7] = avg2x4(C1, C2);  hp[8] = avg4<CONF>(C1, R1, C2, R2);
        // The decision (RTL:1784-1816): the ten costs as KEYS (cost << 16 | rank) whose minimum IS the RTL's tree with its tie-breaks (kHpRank
        // above).  A cost that the RTL caps at 4096 ("over", RTL:1784-1785) never wins - the intra cost is at most 4095 - so it is left
        // as it is, and a dead candidate just gets bit 12 set: five ORs on the packed pairs of totals, from a table word per pair.
        uint32_t best = 2u;                                     // (search-less debug runs: the centre candidate at cost 0)
        if (!(kDebug && (g.ablate & 2))) {
            // "intra cost" accumulates the absolute deviation from the mean on top of the pixel sum, 16-bit wrap
            // (RTL:1744, 1774-1777, 1791): the pixel sum S (formed in front of the search), then the deviation rides along with the nine SADs
            const uint32_t m = (S >> 8) & 255u;
            // half-pel candidates that would reach outside the frame or beyond the search range are dead (RTL:1757-1760)
            // as 0 / 1 integers by sign-bit arithmetic: fx + YR - 1 is negative exactly for fx = -YR, and so on
            const int no_l = (in_l ^ 1) | (int)((uint32_t)(fx + YR - 1) >> 31), no_r = (in_r ^ 1) | (int)((uint32_t)(YR - 1 - fx) >> 31);
            const int no_u = (in_u ^ 1) | (int)((uint32_t)(fy + YR - 1) >> 31), no_d = (in_d ^ 1) | (int)((uint32_t)(YR - 1 - fy) >> 31);
            typedef const __attribute__((address_space(4))) uint32_t *sld;
            const sld dtab = (sld)(ltab2 - 1024 + kConstHpDead + (uint32_t)sgpr(no_l | (no_r << 1) | (no_u << 2) | (no_d << 3)) * (uint32_t)kHpDeadStride);
            const uint32_t d0 = dtab[0], d1 = dtab[1], d2 = dtab[2], d3 = dtab[3], d4 = dtab[4];
            // ten sums as five packed pairs (each total <= 65280; v_sad_hi_u8 packs for free), four of them reduced
            // together by wave_sum4
            uint32_t pk[5];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                pk[k] = __builtin_amdgcn_sad_hi_u8(cur4, hp[2 * k + 1], __builtin_amdgcn_sad_u8(cur4, hp[2 * k], 0u));
            pk[4] = __builtin_amdgcn_sad_hi_u8(cur4, m * 0x01010101u, __builtin_amdgcn_sad_u8(cur4, hp[8], 0u));
            const int q4 = wave_sum4((int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]);
            const uint32_t t01 = (uint32_t)__builtin_amdgcn_readlane(q4, 15) | d0, t23 = (uint32_t)__builtin_amdgcn_readlane(q4, 47) | d1;
            const uint32_t t45 = (uint32_t)__builtin_amdgcn_readlane(q4, 31) | d2, t67 = (uint32_t)__builtin_amdgcn_readlane(q4, 63) | d3;
            const uint32_t t8d = (uint32_t)wave_sum((int)pk[4]);
            // key = cost << 16 | rank: ONE scalar instruction per candidate (s_pack_ll / s_pack_lh take the low / high half of the pair)
#define M2V_KLO(p, k) ([](uint32_t pp) { uint32_t d; asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(d) : "n"(kHpRank[k]), "s"(pp)); return d; }(p))
#define M2V_KHI(p, k) ([](uint32_t pp) { uint32_t d; asm("s_pack_lh_b32_b16 %0, %1, %2" : "=s"(d) : "n"(kHpRank[k]), "s"(pp)); return d; }(p))
            const uint32_t S2 = (S + (t8d >> 16)) & 0xFFFFu;       // the intra cost, capped at 0xFFF (RTL:1791)
            const uint32_t k9 = M2V_KLO(umin32(S2, 0xFFFu), 9);
            // (every two-way minimum pinned to the scalar unit: left alone the compiler folds them into v_min3_u32, which only the vector
            // ALU has, with a v_mov per operand and a v_readfirstlane behind)
            auto smin = [](uint32_t x, uint32_t y) { return (uint32_t)sgpr((int)umin32(x, y)); };
            const uint32_t a = smin(smin(M2V_KLO(t01, 0), M2V_KHI(t01, 1)), smin(M2V_KLO(t23, 2), M2V_KHI(t23, 3)));
            const uint32_t b = smin(smin(M2V_KLO(t45, 4), M2V_KHI(t45, 5)), smin(M2V_KLO(t67, 6), M2V_KHI(t67, 7)));
            best = smin(smin(a, b), smin(M2V_KLO(t8d | d4, 8), k9));
#undef M2V_KLO
#undef M2V_KHI
        }
        const uint32_t rk = 2u * (best & 15u);                   // twice the winner's rank
        inter = (int)(rk != 2u);
        int hy = 0, hx = 0;
        if (inter) {
            const int hy1 = (int)((kHpRankHy >> rk) & 3u), hx1 = (int)((kHpRankHx >> rk) & 3u);
            hy = hy1 - 1; hx = hx1 - 1;
            // the winner is wave-uniform: the register is picked by VGPR index mode (s_set_gpr_idx_on, one v_mov), not by a jump tree
            typedef uint32_t u32x9_t __attribute__((ext_vector_type(9)));
            const u32x9_t hv = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8]};
            pred4 = hv[sgpr(2 * hy1 + hy1 + hx1)];
        }
        mvy = 2 * fy + hy;                                      // RTL:1827-1828
        mvx = 2 * fx + hx;
    }

    // group 3 of the lane table and the basis rows: one phase ahead of stage G (an I frame asked at the start).  In front of the half-pel
    // phase, where group 3 used to be asked for, its eleven registers cost more than the longer head start gained (- 0.5 % per step here)
    if constexpr (P) { M2V_REQUEST_G3(); M2V_REQUEST_BASIS(); }
    // ---- prediction into tile layout; current and predicted samples as SIGNED bytes for the transform (RTL:1891-1917,
    // 1980-2014).  The 9-bit residual c - p is never formed: stage G needs only sum_k M[j][k] (c_k - p_k), which is
    // sum_k M[j][k] (c_k - 128) + sum_k (-M[j][k]) (p_k - 128), two v_dot4_i32_i8 chains on the bytes XOR 0x80.
    // s_cp[tile][row] = 8 current bytes, then 8 prediction bytes: one 16-byte LDS read per tile row in stage G.
    typedef __attribute__((address_space(3))) uint32_t *LdsW32;
    typedef __attribute__((address_space(3))) uint16_t *LdsW16;
    keep_alive(kq0); keep_alive(kq1);
    {
        // tile = ((r >> 3) << 1) | (c4 >> 1), ti = ((r & 7) << 3) | ((c4 & 1) << 2)
        *(LdsW32)(uintptr_t)kq0.z = pred4;                                   // s_pred[tile][ti]
        *(LdsW32)(uintptr_t)kq0.w = cur4 ^ 0x80808080u;                      // s_cp[tile][r & 7][(c4 & 1) << 2]
        *(LdsW32)(uintptr_t)(kq0.w + 8u) = pred4 ^ 0x80808080u;              // ... [8 + ((c4 & 1) << 2)]
    }
    if (!(lane & 4)) {
        // the 4:2:0 samples of the current macroblock: even-row lanes own (r >> 1, 2 c4 .. 2 c4 + 1) of U and of V
        const uint32_t cs = cuv ^ 0x80808080u;
        *(LdsW16)(uintptr_t)kq1.x = (uint16_t)cs;                            // s_cp[4][r >> 1][2 * c4]
        *(LdsW16)(uintptr_t)(kq1.x + 128u) = (uint16_t)(cs >> 16);           // s_cp[5][..]
    }
    {
        // chroma prediction: integer part mv>>2 (floor), half flag = bit 1 of mv (RTL:1854-1887, 1904-1916).  Every lane owns
        // two samples (yc, xc .. xc + 1) of ONE plane - lanes 0-31 U, lanes 32-63 V - so both planes are predicted at once.
        // pl = lane >> 5, yc = (lane >> 2) & 7, xc = 2 * c4
        uint32_t pr = 0x8080u;                          // two packed prediction bytes
        if constexpr (P) {
            if (inter) {
                // chroma vector in chroma half samples: RTL floor; ISO 7.6.3.7 divides with truncation toward zero
                const int cmy = CONF ? (mvy - (mvy >> 31)) >> 1 : mvy >> 1, cmx = CONF ? (mvx - (mvx >> 31)) >> 1 : mvx >> 1;
                const int cyi = cmy >> 1, cxi = cmx >> 1, fyh = cmy & 1, fxh = cmx & 1;
                // col .. col+2 are needed, col <= 13: the fourth byte fetched may belong to the next row, it is never used.
                // Row row + 1 is only used with a vertical half sample, and then cyi <= UR - 1: it stays inside the window
                // (without one the fetch below may reach one row past it - inside this kernel's LDS, value unused).
                // row = yc + cyi + UR, col = xc + cxi + 4: the lane parts (plane, yc + UR; xc + 4) from the table
                const uint32_t col = 2u * (uint32_t)c4 + 4u + (uint32_t)cxi;      // (xc + 4 by arithmetic: the table's copy, kq2.x, was asked for a moment ago)
                const uint32_t sft = col & 3u;
                const uint32_t *const cw = (const uint32_t *)(LdsU32 *)(uintptr_t)(kq1.w + (uint32_t)sgpr(cyi * 16) + (col & ~3u));
                const uint32_t a = __builtin_amdgcn_alignbyte(cw[1], cw[0], sft);      // T[row][col..col+3]
                const uint32_t c = __builtin_amdgcn_alignbyte(cw[5], cw[4], sft);      // T[row+1][col..]
                const uint32_t b = a >> 8, d = c >> 8;
                if (fyh && fxh) pr = avg4<CONF>(a, b, c, d);
                else if (fxh)   pr = avg2x4(a, b);
                else if (fyh)   pr = avg2x4(a, c);
                else            pr = a;
                pr &= 0xFFFFu;
            }
        }
        *(LdsW16)(uintptr_t)kq1.y = (uint16_t)pr;                              // s_pred[4 + pl][(yc << 3) | xc]
        *(LdsW16)(uintptr_t)kq1.z = (uint16_t)(pr ^ 0x8080u);                 // s_cp[4 + pl][yc][8 + xc]
    }
    M2V_WAVE_SYNC();

    M2V_STOP(3);            // ... up to the prediction
    // ---- stage G: 2-D forward DCT (RTL:2029-2062); lane = (i = lane>>3, j = lane&7) ------------
    keep_alive(kq2);
    // DCT-as-GEMM trial (north star): the four luma tiles through the matrix cores, the two chroma tiles as before
    constexpr int kT0 = kMfmaChroma ? 6 : kMfmaLuma ? 4 : 0;         // first tile on the VALU path
#pragma unroll
    for (int t = kT0; t < 6; ++t) {
        // R1[r][j] = sum_k (c[r][k] - p[r][k]) * DCTM[j][k]: 16 bytes of LDS, 4 v_dot4
        const u32x4_t xr = *(LdsU4)(uintptr_t)(kq2.y + (uint32_t)(t * 128));       // s_cp[t][lane >> 3][0 .. 15]
        int acc = dot4_first(xr.x, mj.x);
        acc = __builtin_amdgcn_sdot4((int)xr.y, (int)mj.y, acc, false);
        acc = __builtin_amdgcn_sdot4((int)xr.z, (int)nj.x, acc, false);
        acc = __builtin_amdgcn_sdot4((int)xr.w, (int)nj.y, acc, false);
        s_t[slot_of_tile(t)][lane] = acc;
    }
    // The 16x16 luma block Z = current - prediction holds the 2x2 tiles; with B16 = blockdiag(DCTM, DCTM) the four 8x8
    // transforms are B16 . Z . B16^T in place (tools/ubench/mfma_dct_check.hip checks this formulation on its own).
    //   pass 1  T = Z . B16^T: ONE v_mfma_i32_16x16x32_i8, K = 16 current columns (+B16) and 16 prediction columns (-B16);
    //           A = a row of signed bytes straight from s_cp, B = per-lane constant
    //   pass 2  Y^T = T^T . B16^T: T is 19 bit, i8 operands: three signed byte limbs (T + 0x808080) ^ 0x808080, the accumulator
    //           layout of pass 1 (rows 4g .. 4g+3 of column c) IS the A layout of T^T for a K = 4g .. 4g+3 slice (and equally the B
    //           layout of T: which of the two operands the limbs are decides whether Y or Y^T comes out), a 4x4 byte transpose
    //           (7 v_perm) sorts the limbs, one MFMA per limb, recombined by two shift-adds per coefficient.
    // The two chroma tiles go through the same four instructions as the block [U x; x V]: A row c < 8 = row c of U in both halves of
    // K, row c >= 8 = row c - 8 of V; B16 being block diagonal, the quadrants beside the diagonal (x: U, V again) only reach output
    // quadrants that are never stored.  No basis rows in registers, no second pass through LDS (round 3 costed this at 49 against 44
    // vector instructions and left it; what it removes is three vector loads, eighteen LDS instructions and twelve registers).
    // transform sum of block row 4g + v, column c, PLUS kRound = 2048 + (2 << 12): the DCT's rounding constant and the inter
    // quantiser's "+ 2" (below), both added for free as the accumulator input 40 of the middle limb (40 << 8)
    constexpr int kRound = P ? 2048 + (2 << 12) : 2048;        // an I frame has no non-intra macroblock: just the transform's rounding
    int yacc[4] = {0, 0, 0, 0}, yacc_c[4] = {0, 0, 0, 0};
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4i_t *LdsW4i;
    // (matrix-core lane = (g = lane >> 4, c = lane & 15); its LDS slots come from the lane table)
    if constexpr (kMfmaLuma) {
        const long b1 = mf_b1, a2lo = mf_a2lo, a2hi = mf_a2hi;
        mf_zoff = M2V_QUAD(kQuadMfma0, MfmaLane, zoff[0]);     // for the quantiser
        auto mfma_dct = [&](const long a1, int (&ya)[4]) {
            const v4i_t zero4 = {0, 0, 0, 0};
            const v4i_t tt = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, b1, zero4, 0, 0, 0);
            uint32_t e[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) e[v] = (uint32_t)tt[v] + 0x808080u;       // bias here, the sign flip on the three sorted registers
            const uint32_t p01 = __builtin_amdgcn_perm(e[1], e[0], 0x05010400u), p23 = __builtin_amdgcn_perm(e[3], e[2], 0x05010400u);
            const uint32_t w0 = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ 0x80808080u, w1 = __builtin_amdgcn_perm(p23, p01, 0x07060302u) ^ 0x80808080u;
            const uint32_t q01 = __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0602u), q23 = __builtin_amdgcn_perm(e[3], e[2], 0x0c0c0602u);
            const uint32_t w2 = __builtin_amdgcn_perm(q23, q01, 0x05040100u) ^ 0x80808080u;
            // K = 32 per instruction, a limb fills 16: limbs 0 and 1 share one B operand (A = {a, 0} picks the low dword, {0, a} the
            // high one); limb 2 rides with a dword that A multiplies by zero - any register will do, none is written for it
            uint32_t junk;
            asm volatile("" : "=v"(junk));
            const long b01 = (long)(((unsigned long long)w1 << 32) | w0), b2 = (long)(((unsigned long long)junk << 32) | w2);
            // The limbs of T as the A operand, the basis as B: the product is Y TRANSPOSED, Y^T = T^T . B16^T - lane (g, c) ends up with
            // Y[c][4g .. 4g+3], four neighbouring coefficients of ONE ROW of tile 2 (c >> 3) + (g >> 1): half of what one lane of the
            // inverse transform's row pass needs, the other half is in lane (g ^ 1, c) - the dequantised luma values never go through LDS
            // (profiles/r05_experiments.txt item 12)
            const v4i_t y0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(b01, a2lo, zero4, 0, 0, 0);
            const v4i_t round4 = {kRound >> 8, kRound >> 8, kRound >> 8, kRound >> 8};
            const v4i_t y1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(b01, a2hi, round4, 0, 0, 0);
            const v4i_t y2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(b2, a2lo, zero4, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // two v_lshl_add_u32.  Both are left to the compiler: an instruction that reads a matrix-core result needs wait
                // states behind the MFMA which the compiler inserts for its own instructions only, never for inline asm.  The
                // empty asm in between merely keeps it from re-associating the chain into two shifts and a three-operand add.
                int t = (y2[v] << 8) + y1[v];
                asm("" : "+v"(t));
                ya[v] = (t << 8) + y0[v];
            }
        };
        // s_cp[((mc >> 3) << 1) | (mg & 1)][mc & 7][8 * (mg >> 1)] / s_cp[4 + (mc >> 3)][mc & 7][8 * (mg >> 1)]
        const long a1 = *(const __attribute__((address_space(3))) long *)(uintptr_t)kq2.z;
        long a1c = 0;
        if constexpr (kMfmaChroma) a1c = *(const __attribute__((address_space(3))) long *)(uintptr_t)mf_a1c;
        mfma_dct(a1, yacc);
        if constexpr (kMfmaChroma) mfma_dct(a1c, yacc_c);
    }
    M2V_WAVE_SYNC();

    M2V_STOP(4);            // ... up to the forward transform
    // ---- quantise (RTL:2065-2077), zig-zag + coded flags (RTL:2452-2468), dequantise (RTL:2129-2150)
    // the intra quantiser's lane constants: an I frame loads them up front, a P frame only inside its (rare) intra branch
    int wq = 0;
    uint32_t wrecip = 0;
    if constexpr (!P) { wq = (int)iq_w.z; wrecip = iq_w.w; }
    const int zz = (int)(kq3.x >> 1);                 // zig-zag position of the lane (the table holds the byte offset in a tile)
    const int Q = g.Q;
    // group 4 (column pass and chroma store of the reconstruction), requested two phases ahead
    const uint32_t k4r = kq4.y, k4p = kq4.z;
    const size_t mbidx = (size_t)fidx * g.mbs + mb;
    const bool need_rec = job.rec != nullptr && !(kDebug && (g.ablate & 8));
    int cbp = 0;
    v4i_t xl4 = {0, 0, 0, 0};           // matrix-core transform: the lane's four dequantised luma coefficients (row mc, columns 4 mg ..)
    const bool chroma_lane = ((((uint32_t)lane >> 5) ^ ((uint32_t)lane >> 3)) & 1u) == 0u;     // matrix-core layout: g >> 1 == c >> 3
    if (kDebug && (g.ablate & 16)) {
        for (int t = 0; t < 6; ++t) s_zig[t][lane] = 0;
        cbp = inter ? 0 : 63;
    } else if (inter) {
        const int qneg = sgpr(-(((1 << (4 + Q)) - 5) << 12));      // MINUS the bias of a negative value (it multiplies the sign mask)
        if constexpr (kMfmaLuma) {
            // the four luma tiles in (transposed) accumulator layout: lane (g, c) owns columns 4g .. 4g+3 of row c of the 16x16 block,
            // i.e. four coefficients of tile 2 (c >> 3) + (g >> 1); their s_zig slots come from the lane table, their
            // raster slots in s_t are neighbours
            const uint32_t zo[4] = {mf_zoff.x, mf_zoff.y, mf_zoff.z, mf_zoff.w};
            const uint32_t xrow = kq2.w;                     // &s_t[slot of tile ((mc >> 3) << 1) | (mg >> 1)][((mc & 7) << 3) | ((mg & 1) << 2)]
            // four coefficients of one pass of the transform: levels to s_zig (+ zadd bytes), dequantised values to s_t (+ xadd BYTES)
            auto quant4 = [&](const int (&ya)[4], const uint32_t zadd, const uint32_t xadd) {
                int nzor = 0, qv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    // see the VALU-path loop below for the arithmetic; ya already holds acc + (2 << 12)
                    const int q = mad24_ms(ya[v] >> 31, qneg, ya[v]) >> (16 + Q);
                    *(int16_t *)((uint8_t *)&s_zig[0][0] + zo[v] + zadd) = (int16_t)q;
                    if (kDebug && coef_dbg) coef_dbg[mbidx * 384 + tile_of_slot((int)((zo[v] + zadd) >> 7)) * 64 + ((zo[v] & 127u) >> 1)] = (int16_t)q;
                    nzor |= q;
                    qv[v] = q;
                }
                if (need_rec) {                     // one test for the four coefficients
                    v4i_t x4;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        // RTL:2134-2137 clamps to +-2047; the clamp cannot bind here: |q| <= 16322 >> (4 + Q), so (2 |q| + 1) << Q <= 2044
                        // for every Q_LEVEL (tests/test_host_logic.py::test_inverse_quantisers_never_reach_their_clamps)
                        x4[v] = (2 * qv[v] + sign_of(qv[v])) << Q;
                    }
                    if (xadd == 0u) xl4 = x4;                       // luma: the row pass takes them from the registers
                    else *(LdsW4i)(uintptr_t)(xrow + xadd) = x4;    // chroma block of an I frame: four neighbours of a row, ONE store
                }
                return nzor;
            };
            // coded flags of the four luma tiles: tile 2 ty + tx lives in lanes 32 tx + 16 h + 8 ty + (0 .. 7), h = 0, 1
            const unsigned long long nzm = ballot(quant4(yacc, 0u, 0u) != 0);
            const uint32_t lo = (uint32_t)sgpr((int)(uint32_t)nzm), hi = (uint32_t)sgpr((int)(uint32_t)(nzm >> 32));
            cbp = ((lo & 0x00FF00FFu) ? 8 : 0) | ((hi & 0x00FF00FFu) ? 4 : 0) | ((lo & 0xFF00FF00u) ? 2 : 0) | (int)(((hi & 0xFF00FF00u) | (0u - (hi & 0xFF00FF00u))) >> 31);
            // the chroma block: U where luma tile 0 is (g < 2, c < 8: lanes 0-7, 16-23), V where tile 3 is (lanes 40-47, 56-63), one slot
            // further in both buffers; the other lanes hold the quadrants nobody wants
            if constexpr (kMfmaChroma) {
                int nzor_c = 0;
                if (chroma_lane) nzor_c = quant4(yacc_c, 128u, 4u * kTileStride);
                const unsigned long long nzc = ballot(nzor_c != 0);
                cbp = (cbp << 2) | ((uint32_t)sgpr((int)(uint32_t)nzc) ? 2 : 0) | ((uint32_t)sgpr((int)(uint32_t)(nzc >> 32)) ? 1 : 0);
            }
        }
#pragma unroll
        for (int t = kT0; t < 6; ++t) {
            int acc = mad24_s(bi[0], s_t[slot_of_tile(t)][dj], kRound);   // C[i][j] = (sum_k DCTM[i][k] * R1[k][j] + 2048) >> 12, and + (2 << 12)
#pragma unroll
            for (int k = 1; k < 8; ++k) acc = mad24(bi[k], s_t[slot_of_tile(t)][k * 8 + dj], acc);   // |R1| < 2^18
            // RTL:2070: sign(C) * min((|C| + 2) >> s, 2047) with C = acc >> 12 and s = 4 + Q, computed on the signed value:
            // for C < 0 it is ceil((C - 2) / 2^s) = (C + 2^s - 3) >> s (identity checked over the 17-bit range in
            // tests/test_host_logic.py).  Two floor shifts with an integer added in between are one:
            // ((acc >> 12) + k) >> s = (acc + (k << 12)) >> (12 + s).  The clamp cannot bind on this path: the basis rows
            // sum to at most 512 in magnitude, so |C| <= (255 * 512 * 512 + 2048) >> 12 = 16320 and |q| <= 16322 >> 5 = 510
            // (tests/test_host_logic.py::test_inter_quantiser_never_reaches_its_clamp).
            // The "+ 2" rides in the accumulator, so the sign taken is that of C + 2, not of C: they differ for C = -2, -1,
            // where both formulas give 0 (s >= 4).  sign mask * (-bias) + acc is one v_mad_i32_i24.
            const int q = mad24_ms(acc >> 31, qneg, acc) >> (16 + Q);
            *(LdsW16)(uintptr_t)(lds_off(&s_zig[slot_of_tile(t)][0]) + kq3.x) = (uint16_t)q;      // s_zig[t][zz]
            if (kDebug && coef_dbg) coef_dbg[mbidx * 384 + t * 64 + zz] = (int16_t)q;
            cbp = (cbp << 1) | (int)any_lane(q != 0);
            if (need_rec) {                             // RTL:2134-2137: (2q + sign(q)) << Q, clamped to +-2047
                int x = (2 * q + sign_of(q)) << Q;
                if constexpr (CONF) {
                    // ISO 7.4.2.3 gives the same product; saturation to [-2048, 2047] (7.4.3), mismatch control (7.4.4);
                    // a block without coefficients is not reconstructed at all
                    x = x < -2048 ? -2048 : x > 2047 ? 2047 : x;
                    const bool coded = ballot(q != 0) != 0ull;
                    const bool even = (__popcll(ballot(x & 1)) & 1) == 0;
                    if (coded && even && lane == 63) x ^= 1;
                }                                       // (the reference's +-2047 clamp cannot bind: see the luma tiles above)
                s_t[slot_of_tile(t)][lane] = x;         // behind this tile's phase-2 reads of it (one wavefront: LDS operations execute in order)
            }
        }
    } else {
        if constexpr (P) { M2V_REQUEST_INTRA(); wq = (int)iq_w.z; wrecip = iq_w.w; }
        const uint32_t qoff = __umul24((uint32_t)wq, (3u << Q) + 2u) >> 3;
        if constexpr (kMfmaLuma) {
            const uint32_t zo[4] = {mf_zoff.x, mf_zoff.y, mf_zoff.z, mf_zoff.w};
            const uint32_t ml_wq = iq_w.y, ml_recip[4] = {iq_recip.x, iq_recip.y, iq_recip.z, iq_recip.w};
            const uint32_t xrow = kq2.w;
            // (the chroma block's coefficients sit at the raster positions of the lane's luma coefficients: same weights, same DC lane)
            auto quant4 = [&](const int (&ya)[4], const uint32_t zadd, const uint32_t xadd) {
                v4i_t x4 = {0, 0, 0, 0};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int wv = (int)((ml_wq >> (8 * v)) & 255u);
                    const uint32_t qo = __umul24((uint32_t)wv, (3u << Q) + 2u) >> 3;
                    const bool is_dc = v == 0 && (lane & 0x17) == 0;           // column 0 of a tile (g even, v = 0), row 0 of a tile
                    const int C = (ya[v] >> 12) - (kRound >> 12);          // ya carries kRound
                    const int sg = C >> 31;
                    uint32_t a = (uint32_t)((C ^ sg) - sg) & 0xFFFFu;
                    if (!is_dc) a = __umul24((a + qo) >> Q, ml_recip[v]) >> 21;
                    else        a = (a + 8u) >> 4;
                    // RTL:2075 clamps to 2047 here and RTL:2139-2144 keeps the inverse quantiser's product in 17 bits and clamps it to +-2047:
                    // an intra block is pixel - 128, so |C| <= 8192, the level is at most 512 (DC) / 272 (AC) and |level * W| < 2^14 - none
                    // of the three can bind (tests/test_host_logic.py::test_inverse_quantisers_never_reach_their_clamps).  The conformant
                    // build keeps its own saturation.
                    const int q = (int)(a ^ (uint32_t)sg) - sg;
                    *(int16_t *)((uint8_t *)&s_zig[0][0] + zo[v] + zadd) = (int16_t)q;
                    if (kDebug && coef_dbg) coef_dbg[mbidx * 384 + tile_of_slot((int)((zo[v] + zadd) >> 7)) * 64 + ((zo[v] & 127u) >> 1)] = (int16_t)q;
                    if (need_rec) {
                        int x;
                        if (!is_dc) {
                            x = __mul24(q, wv);
                            x = Q >= 3 ? x << (Q - 3) : x >> (3 - Q);
                        } else {
                            x = 2 * q;
                        }
                        x4[v] = x;
                    }
                }
                if (need_rec) {
                    if (xadd == 0u) xl4 = x4;
                    else *(LdsW4i)(uintptr_t)(xrow + xadd) = x4;
                }
            };
            quant4(yacc, 0u, 0u);
            if constexpr (kMfmaChroma) { if (chroma_lane) quant4(yacc_c, 128u, 4u * kTileStride); }
        }
#pragma unroll
        for (int t = kT0; t < 6; ++t) {
            int acc = mad24_s(bi[0], s_t[slot_of_tile(t)][dj], 2048);
#pragma unroll
            for (int k = 1; k < 8; ++k) acc = mad24(bi[k], s_t[slot_of_tile(t)][k * 8 + dj], acc);
            const int C = acc >> 12;
            const int sg = C >> 31;
            uint32_t a = (uint32_t)((C ^ sg) - sg) & 0xFFFFu;
            if (lane != 0) a = __umul24((a + qoff) >> Q, wrecip) >> 21;             // exact "/ W" (n < 2^14, recip < 2^19), RTL:2072
            else           a = (a + 8u) >> 4;                                       // (a >> 4) + bit 3, RTL:2074
            if constexpr (CONF) { if (a > 2047u) a = 2047u; }                       // RTL:2075; cannot bind (see the luma tiles)
            const int q = (int)(a ^ (uint32_t)sg) - sg;
            *(LdsW16)(uintptr_t)(lds_off(&s_zig[slot_of_tile(t)][0]) + kq3.x) = (uint16_t)q;      // s_zig[t][zz]
            if (kDebug && coef_dbg) coef_dbg[mbidx * 384 + t * 64 + zz] = (int16_t)q;
            if (need_rec) {
                int x;
                if constexpr (CONF) {
                    // ISO 7.4.2.3: (2 QF W quantiser_scale) / 32 with quantiser_scale = 2 << Q, truncating toward zero
                    if (lane != 0) {
                        const uint32_t m = (a * (uint32_t)wq) << Q;             // |QF| W (1 << Q): < 2^11 * 2^7 * 2^4
                        x = (int)((m >> 3) ^ (uint32_t)sg) - sg;
                        x = x < -2048 ? -2048 : x > 2047 ? 2047 : x;
                    } else {
                        x = 2 * q;
                    }
                    const bool even = (__popcll(ballot(x & 1)) & 1) == 0;      // mismatch control (7.4.4)
                    if (even && lane == 63) x ^= 1;
                } else if (lane != 0) {
                    x = __mul24(q, wq);                 // the 17-bit temporary of RTL:2093 / 2139 and the +-2047 clamp of RTL:2144 cannot bind
                    x = Q >= 3 ? x << (Q - 3) : x >> (3 - Q);
                } else {
                    x = 2 * q;
                }
                s_t[slot_of_tile(t)][lane] = x;         // behind this tile's phase-2 reads of it (one wavefront: LDS operations execute in order)
            }
        }
        cbp = 63;                                       // intra: every tile is coded (RTL:2461)
    }
    // the pattern's code (d_cbp_code[cbp], wave-uniform): a SCALAR load, issued here so that the wait below covers it
    const uint32_t cbp_word = *(const __attribute__((address_space(4))) uint32_t *)(ltab2 - 1024 + kConstCbp + 4u * ((uint32_t)sgpr(cbp) >> 1));
    M2V_WAVE_SYNC();

    M2V_STOP(5);            // ... up to the quantiser / inverse quantiser
    // ---- stage T, coefficient part: run/level VLC of the six tiles (RTL:2777-2847) -----------------
    // Pass 1 (per tile, lane = zig-zag index): ballot the non-zero levels, rank them, and append
    // {run, level} / raw-code symbols to one compact list.  Pass 2 (once per macroblock): table lookup,
    // wave prefix sum of the code lengths, codes ORed MSB-first into the LDS bit buffer.
    // Pass 1 and the table look-up of the first 64 symbols run BEFORE the inverse transform, the rest of pass 2 behind it: the look-up is a
    // memory round trip with nothing of its own to hide behind (an exposed one costs 3 % of the kernel, profiles/r04_experiments.txt item 14);
    // the symbol list lives in R1, which the reconstruction does not touch, the bit buffer (in s_t) is first written behind it.
    // Bits that need the left neighbour (motion vector deltas, DC of Y00 / U / V) are NOT produced here;
    // the rest forms three bit-contiguous segments: A = [cbp][all tiles] (inter) or
    // [AC of Y00][Y01][Y10][Y11] (intra), B = AC of U, C = AC of V.
    uint32_t nsym = 0, idxB = 0, idxC = 0;
    int dcs[6] = {0, 0, 0, 0, 0, 0};
    const uint32_t lane_pos = (uint32_t)lane << 20;
    if (!(kDebug && (g.ablate & 4))) {
        if (inter) {
            const uint32_t e = (cbp_word >> (16 * (cbp & 1))) & 0xFFFFu;                    // d_cbp_code[cbp]
            uint32_t nsym4 = ((uint32_t)-cbp >> 31) << 2;   // pattern 0 (motion vector only) has no code, and a raw symbol needs a length
            const uint32_t eob = (uint32_t)vgpr_const((int)sym_raw(2u, 2u, true));
            const uint32_t sym_base = lds_off(s_sym);
#pragma unroll
            for (int t = 0; t < 6; ++t)
                if ((cbp >> (5 - t)) & 1) nsym4 = vlc_tile_symbols_inter(s_zig[slot_of_tile(t)], sym_base, lane, lane_pos, nsym4, eob);
            if (lane == 0) s_sym[0] = sym_raw(e >> 8, e & 255u, true);
            nsym = nsym4 >> 2;
        } else {
            if (lane == 0) s_sym[-1] = sym_raw(1u, 0u, false);  // the symbol "before" the first one: a block start
            uint32_t nsym4 = 0;
            const uint32_t eob = (uint32_t)vgpr_const((int)sym_raw(2u, 2u, false));
            const uint32_t sym_base = lds_off(s_sym);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (t == 4) idxB = nsym4 >> 2;
                if (t == 5) idxC = nsym4 >> 2;
                nsym4 = vlc_tile_symbols_intra(s_zig[slot_of_tile(t)], sym_base, lane, lane_pos, nsym4, eob, dcs[t], t ? dcs[t - 1] : 0, t >= 1 && t <= 3,
                                               ltab2 - 1024 + kConstDcLuma);
            }
            nsym = nsym4 >> 2;
        }
    }
    M2V_WAVE_SYNC();
    // front half of pass 2 for symbol i: the symbol and, for a {run, level} one, its table entry (left in the load's register: anything
    // computed from it here would be waited for here)
    // run = zig-zag positions skipped since the symbol in front: the previous level of the block, or a raw code
    // (pattern code, DC code, the previous block's end code) that carries the position a block starts from
    auto vlc_run = [&](uint32_t sym, uint32_t before) { return (int)((sym >> 20) & 63u) - ((int)(before << 5) >> 25) - 1; };
    auto vlc_front = [&](uint32_t i, uint32_t &sym, uint32_t &e) {
        sym = s_sym[i];
        e = 0u;
        if (!(sym >> 27)) {
            const int v = (int16_t)(sym & 0xFFFFu);
            const uint32_t before = s_sym[(int)i - 1];
            const int run = vlc_run(sym, before);
            const uint32_t a = (uint32_t)iabs(v);
            // no range test, no select: clamped indices land on the table's zero row / column, the '1s' rule is bank 1
            typedef const __attribute__((address_space(1))) uint16_t *gld16;
            const uint32_t idx = __umul24(umin32((uint32_t)run, (uint32_t)kAcRuns - 1u), (uint32_t)kAcLevels) + umin32(a, (uint32_t)kAcLevels) - 1u +
                                 __umul24((before >> 26) & 1u, (uint32_t)(kAcRuns * kAcLevels));
            e = *(gld16)(ltab2 + 2u * idx);          // d_ac_code2[idx]
        }
    };
    uint32_t sym0 = 0u, e0 = 0u;
    // Every older vector load has been used by now on every path; saying so (s_waitcnt vmcnt(0), free here) keeps the compiler's
    // bookkeeping from putting that wait in front of the first register it is unsure about - in the middle of the inverse transform,
    // where it would wait for the look-up issued below
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if ((uint32_t)lane < nsym) vlc_front((uint32_t)lane, sym0, e0);
    // (the second 64 symbols of an I frame's macroblock are looked up inside pass 2: issued here as well, config c2 loses 3 %,
    // profiles/r04_experiments.txt item 14)

    // ---- stages H..R: Chen-Wang IDCT, reconstruction, store as next reference ------------------
    if (need_rec) {
        keep_alive(kq3);
        if constexpr (EDGE) keep_alive(kq4);
        if constexpr (kMfmaLuma) {
            // rows (RTL:2159-2189).  A luma row is split over lanes (g, c) and (g ^ 1, c), columns 0-3 and 4-7: one v_permlane16_swap per
            // register hands the odd rows' halves to the even rows (the register that receives them needs no content, the odd rows'
            // own copies are dead) - lanes 0-15 and 32-47 then hold the 32 luma rows.  Lanes 16-31 fetch the 16 chroma rows from s_t.
            typedef __attribute__((address_space(3))) v4i_t *LdsV4;
            int a[8], o[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int junk;
                asm volatile("" : "=v"(junk));
                const auto sw = __builtin_amdgcn_permlane16_swap((uint32_t)xl4[v], (uint32_t)junk, false, false);
                a[v] = (int)sw[0]; a[4 + v] = (int)sw[1];
            }
            if (lane < 48) {
                if (lane & 16) {
                    const v4i_t lo = *(LdsV4)(uintptr_t)kq2.x, hi = *(LdsV4)(uintptr_t)(kq2.x + 16u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[k] = lo[k]; a[4 + k] = hi[k]; }
                }
                idct_row(a, o);
                *(LdsV4)(uintptr_t)kq2.x = v4i_t{o[0], o[1], o[2], o[3]};
                *(LdsV4)(uintptr_t)(kq2.x + 16u) = v4i_t{o[4], o[5], o[6], o[7]};
            }
        } else if (lane < 48) {                         // rows: lane = slot*8 + row (RTL:2159-2189), in place
            const int t = lane >> 3, row = lane & 7;
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = s_t[t][row * 8 + k];
            idct_row(a, o);
#pragma unroll
            for (int k = 0; k < 8; ++k) s_t[t][row * 8 + k] = o[k];
        }
        M2V_WAVE_SYNC();
        if (lane < 48) {                                // columns: lane = tile*8 + col (RTL:2238-2279)
            typedef const __attribute__((address_space(3))) int32_t *LdsI32;
            typedef __attribute__((address_space(3))) uint8_t *LdsW8;
            int a[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = *(LdsI32)(uintptr_t)(kq3.y + (uint32_t)(k * 32));      // s_t[t][k * 8 + col]
            idct_col(a, o);
#pragma unroll
            for (int k = 0; k < 8; ++k) {               // add_clip_0_255 (RTL:786-795, 2352)
                LdsW8 const pp = (LdsW8)(uintptr_t)(kq3.z + (uint32_t)(k * 8));                        // s_pred[t][k * 8 + col]
                const int v = (int)*pp + o[k];
                *pp = (uint8_t)(v > 255 ? 255 : v < 0 ? 0 : v);
            }
        }
        M2V_WAVE_SYNC();
    }

    M2V_STOP(6);            // everything but the second half of the entropy coder
    {
        // clear what pass 2 can reach: a symbol is at most 26 bits (typically 25 symbols: ONE store of 64 words instead of five
        // predicated ones over the whole 304-word buffer); the first 64 words always, they are what a compact slot copies out
        s_bits[lane] = 0u;
        const int need = sgpr((int)((nsym * 26u) >> 5) + 2);
        for (int k = 64; k < need && k < kSlotWords; k += 64)
            if (k + lane < kSlotWords) s_bits[k + lane] = 0u;
        M2V_WAVE_SYNC();

        uint32_t pos = 0, offB = 0, offC = 0;
        for (uint32_t base = 0; base < nsym; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            uint32_t code = 0, len = 0;
            if (i < nsym) {
                uint32_t sym = sym0, e = e0;
                if (base) vlc_front(i, sym, e);             // (wave-uniform: more than 64 symbols is an I-frame matter)
                const uint32_t rawlen = sym >> 27;
                if (rawlen) {
                    code = sym & 0xFFFFFu;
                    len = rawlen;
                } else {
                    const int v = (int16_t)(sym & 0xFFFFu);
                    e &= 0xFFFFu;
                    if (e) {                                 // run/level VLC + sign (RTL:2535-2540)
                        code = ((e & 255u) << 1) | (v < 0 ? 1u : 0u);
                        len = (e >> 8) + 1u;
                    } else {                                 // escape (RTL:2542-2543); rare: the run is formed again from the list
                        code = (1u << 18) | ((uint32_t)vlc_run(sym, s_sym[(int)i - 1]) << 12) | ((uint32_t)v & 0xFFFu);
                        len = 24;
                    }
                }
            }
            const int incl = wave_scan_incl((int)len);
            const uint32_t excl = pos + (uint32_t)incl - len;
            if (len) lds_put(s_bits, excl, code, len);
            if (!inter) {                       // segment boundaries (intra only: a non-intra macroblock stores one segment)
                if (idxB >= base && idxB < base + 64) offB = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)(idxB - base));
                if (idxC >= base && idxC < base + 64) offC = (uint32_t)__builtin_amdgcn_readlane((int)excl, (int)(idxC - base));
            }
            pos += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
        }
        // the reconstruction goes out here, behind pass 2: in front of it the stores would be waited for together with the look-up
        // (the counter that tells when a load has arrived counts stores as well)
        if (need_rec) {
            // scalar base + 32-bit lane offset (a generic pointer costs a 64-bit vector add per store); V sits csz bytes behind U
            typedef __attribute__((address_space(1))) uint32_t *gst32;
            // tiled (rec_luma_off): the lane's four pixels (row r, columns 4 c4 ..) go to the right half of tile bx - byte 16 r + 8 + 4 c4 - or, from
            // column 8 on, to the left half of tile bx + 1 - byte 256 + 16 r + 4 (c4 - 2): 4 lane + 8 + 240 (c4 >> 1); chroma (lanes < 32:
            // plane, row, half) likewise 4 lane + 4 + 120 half
            uint8_t *recY = job.rec;
            {
                const uint32_t v = *(LdsU32 *)(uintptr_t)kq0.z;                  // the lane's four pixels: s_pred[tile][ti] again
                *(gst32)(recY + (tile * 256u + 8u + __umul24((uint32_t)lane & 2u, 120u) + (uint32_t)lane * 4u)) = v;
            }
            if (lane < 32) {
                // pl = lane >> 4, yc = (lane & 15) >> 1, half = lane & 1: s_pred[4 + pl][(yc << 3) | (half << 2)]
                const uint32_t v = *(LdsU32 *)(uintptr_t)kq3.w;
                *(gst32)(recY + (g.rysz + tile * 128u + 4u + __umul24((uint32_t)lane & 1u, 120u) + (uint32_t)lane * 4u)) = v;
            }
            if (EDGE && edge_blk) {
                // per frame of the step's halo list: [YR rows of W luma][UR rows of cw U][UR rows of cw V] (k_halo_pack's layout)
                constexpr uint32_t YR2 = 2 * VL, UR2 = VL;
                const uint32_t chunk = (YR2 + UR2) * (uint32_t)W, cw = (uint32_t)g.cw;
                const uint32_t fbase = (uint32_t)job.hidx * chunk;
                const uint32_t vy = *(LdsU32 *)(uintptr_t)kq0.z;                // the lane's four luma pixels of row r
                const uint32_t vc = *(LdsU32 *)(uintptr_t)kq3.w;                // lanes < 32: four chroma pixels of row kq4.x, plane k4p
                const uint32_t xl = (uint32_t)(16 * bx + 4 * c4), xc = (uint32_t)(8 * bx) + k4r;
                // peer transport: the buffers are the NEIGHBOURS' memory.  Write-through stores (system-scope relaxed atomic stores =
                // global_store .. sc0 sc1: the bytes leave this GPU's caches at once), then - below - the wavefront drains its stores and ONE
                // lane adds 1 to the neighbour's arrival counter: the hand-off of MI355X_MICROARCH.md "visibility" (write-through payload,
                // drain, flag), at system scope.  Measured for ALL rows of a frame at agent scope (profiles/r04_experiments.txt item 2) the
                // write-through costs 16 %; here it is 9 of a strip's 24 rows' worth of one macroblock row in sixteen.
                auto st = [&](uint8_t *base, uint32_t off, uint32_t v) {
                    typedef __attribute__((address_space(1))) unsigned int *gu32p;
                    if (PEER && !(kDebug && (g.ablate & (1 << 22)))) __hip_atomic_store((gu32p)(base + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    else *(gst32)(base + off) = v;
                };
                const bool put_u = halo_up != nullptr && by == g.edge_top, put_d = halo_down != nullptr && by == g.edge_bot;     // wave-uniform
                if (put_u) {
                    if ((uint32_t)r < YR2) st(halo_up, fbase + (uint32_t)r * (uint32_t)W + xl, vy);
                    if (lane < 32 && kq4.x < UR2) st(halo_up, fbase + YR2 * (uint32_t)W + (k4p * UR2 + kq4.x) * cw + xc, vc);
                }
                if (put_d) {
                    if ((uint32_t)r >= 16u - YR2) st(halo_down, fbase + ((uint32_t)r - (16u - YR2)) * (uint32_t)W + xl, vy);
                    if (lane < 32 && kq4.x >= 8u - UR2) st(halo_down, fbase + YR2 * (uint32_t)W + (k4p * UR2 + (kq4.x - (8u - UR2))) * cw + xc, vc);
                }
                if constexpr (PEER) {
                    if (put_u || put_d) {
                        typedef __attribute__((address_space(1))) unsigned int *gu32p;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every store of this wavefront has left (one wavefront per block)
                        if (lane == 0 && !(kDebug && (g.ablate & (1 << 23)))) {
                            const uint32_t slot = (uint32_t)job.hidx * (uint32_t)kPeerCntStride;
                            if (put_u) __hip_atomic_fetch_add((gu32p)(ps.cnt_up + slot), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            if (put_d) __hip_atomic_fetch_add((gu32p)(ps.cnt_down + slot), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                }
            }
        }
        uint32_t lenA, lenB = 0, lenC = 0;
        if (inter) lenA = pos;
        else { lenA = offB; lenB = offC - offB; lenC = pos - offC; }   // intra: tiles 4 and 5 always carry at least the EOB
        M2V_WAVE_SYNC();
        const uint32_t nwords = (pos + 31u) >> 5;
        if (nwords <= (uint32_t)kSmallSlotWords) {          // the common case: one 128-byte line, or a half / a quarter of one
            if (nwords <= (uint32_t)kMicroSlotWords) {
                if (lane < kMicroSlotWords) slots_small[g.s8_off + mbidx * kMicroSlotWords + lane] = s_bits[lane];
            } else if (nwords <= (uint32_t)kTinySlotWords) {
                if (lane < kTinySlotWords) slots_small[g.s16_off + mbidx * kTinySlotWords + lane] = s_bits[lane];
            } else if (lane < kSmallSlotWords) slots_small[mbidx * kSmallSlotWords + lane] = s_bits[lane];
        } else {
            uint32_t *slot = slots + mbidx * kSlotWords;
#pragma unroll 1
            for (uint32_t k = lane; k < nwords; k += 64) slot[k] = s_bits[k];
        }
        if (lane == 0) {
            mbinfo[mbidx] = (uint32_t)inter | ((uint32_t)cbp << 1) | (((uint32_t)(inter ? mvx : 0) & 255u) << 8) |
                            (((uint32_t)(inter ? mvy : 0) & 255u) << 16);
            MbAux aux;
            aux.w0 = lenA | (lenB << 16);
            aux.w1 = lenC | (((uint32_t)dcs[5] & 0xFFFFu) << 16);
            aux.w2 = ((uint32_t)dcs[0] & 0xFFFFu) | (((uint32_t)dcs[3] & 0xFFFFu) << 16);
            aux.w3 = (uint32_t)dcs[4] & 0xFFFFu;
            mbaux[mbidx] = aux;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// k_slice_scan: one block per (frame, slice): total bit length of each macroblock (stored segments + neighbour-dependent codes
// + 38-bit slice header on the first one) and, from their sum, the byte size of the slice - what k_frame_scan needs to place the
// slices.  The codes themselves and the bit offsets inside the slice are formed again by k_assemble from the same 20 bytes per
// macroblock (a round trip of 16 + 4 bytes per macroblock through memory cost more than the few dozen instructions).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mb_total_bits(const MbDep &d, const MbAux &aux, bool first_in_slice)
{
    return (first_in_slice ? 38u : 0u) + d.p1.len + d.p2.len + d.p3.len + (aux.w0 & 0xFFFFu) + (aux.w0 >> 16) + (aux.w1 & 0xFFFFu);
}

__global__ __launch_bounds__(128) void k_slice_scan(const FrameJob *__restrict__ jobs, Geom g,
                                                    const uint32_t *__restrict__ mbinfo, const MbAux *__restrict__ mbaux,
                                                    uint32_t *__restrict__ mb_len, uint32_t *__restrict__ slice_bytes, int f0)
{
    __shared__ uint32_t s_w[2];
    const int tid = threadIdx.x;
    const int rows = g.row1 - g.row0;
    const int f = f0 + (int)(blockIdx.x / rows), by = g.row0 + (int)(blockIdx.x % rows);      // blockIdx = (frame - f0) * rows + local row
    const size_t base = ((size_t)f * g.mbh + by) * g.mbw;
    uint32_t len = 0;
    if (tid < g.mbw) {
        const size_t idx = base + tid;
        const uint32_t info = mbinfo[idx];
        const MbAux aux = mbaux[idx];
        const bool has_left = tid > 0;
        const uint32_t linfo = has_left ? mbinfo[idx - 1] : 0u;
        const MbAux laux = has_left ? mbaux[idx - 1] : MbAux{0, 0, 0, 0};
        len = mb_total_bits(mb_dependent(info, aux, has_left, linfo, laux, jobs[f].i_frame), aux, tid == 0);
        mb_len[idx] = len;
    }
    // the slice's bits: sums inside the two wavefronts by DPP, one word each through LDS
    const uint32_t sum = (uint32_t)wave_scan_incl((int)len);
    if ((tid & 63) == 63) s_w[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) slice_bytes[(size_t)f * g.mbh + by] = (s_w[0] + s_w[1] + 7u) >> 3;   // next header aligns (RTL:2940-2943)
}

// ----------------------------------------------------------------------------------------------
// byte-aligned headers, written field by field by ONE thread (they are a few dozen bytes per frame)
// ----------------------------------------------------------------------------------------------
struct ByteWriter {
    uint8_t *p;
    uint32_t acc;
    int nbits;
    __device__ void put(uint32_t v, int len)
    {
        for (int i = len - 1; i >= 0; --i) {
            acc = (acc << 1) | ((v >> i) & 1u);
            if (++nbits == 8) { *p++ = (uint8_t)acc; acc = 0; nbits = 0; }
        }
    }
    __device__ void align() { if (nbits) put(0, 8 - nbits); }
};


// [group_of_pictures_header] picture_header picture_coding_extension of one frame (RTL:2645-2698)
__device__ inline void write_frame_headers(uint8_t *p, const FrameJob &job)
{
    ByteWriter w{p, 0u, 0};
    if (job.i_frame == 0) {
        // group_of_pictures_header, closed_gop = 1; time code of frame n at 24 fps (RTL:2645-2656, 2685-2698)
        const uint32_t n = job.n;
        const uint32_t hh = n / 86400u;
        w.put(0x000001B8u, 32);
        w.put(hh > 63u ? 63u : hh, 6);
        w.put((n / 1440u) % 60u, 6);
        w.put(1u, 1);
        w.put((n / 24u) % 60u, 6);
        w.put(n % 24u, 6);
        w.put(2u, 2);
        w.align();
    }
    // picture_header + picture_coding_extension (RTL:2670-2682)
    w.put(0x00000100u, 32);
    w.put((uint32_t)job.i_frame, 10);       // temporal_reference
    if (job.i_frame == 0) { w.put(1u, 3); w.put(0u, 16); w.put(0u, 3); }
    else                  { w.put(2u, 3); w.put(0u, 16); w.put(0u, 1); w.put(7u, 3); w.put(0u, 7); }
    w.put(0x000001B5u, 32);
    w.put(8u, 4);                           // picture coding extension
    w.put(0x1111u, 16);                     // f_code[s][t] = 1
    w.put(2u, 2);                           // intra_dc_precision 10 bit
    w.put(3u, 2);                           // frame picture
    w.put(1u, 1);                           // top_field_first
    w.put(1u, 1);                           // frame_pred_frame_dct
    w.put(0u, 8);
    w.put(0u, 6);
}

// sequence_header + sequence_extension + sequence_display_extension (RTL:2598-2617)
__device__ inline void write_sequence_headers(uint8_t *p, const Geom &g)
{
    ByteWriter w{p, 0u, 0};
    w.put(0x000001B3u, 32);
    w.put((uint32_t)g.W, 12); w.put((uint32_t)g.H, 12);
    w.put(1u, 4); w.put(2u, 4); w.put(10000u, 18); w.put(1u, 1); w.put(0u, 10); w.put(0u, 3);
    w.put(0x000001B5u, 32);
    w.put(1u, 4); w.put(0x44u, 8); w.put(0u, 1); w.put(1u, 2); w.put(0u, 4); w.put(0u, 12); w.put(1u, 1);
    w.put(0u, 8); w.put(0u, 8);
    w.put(0x000001B5u, 32);
    w.put(2u, 4); w.put(1u, 3); w.put(1u, 1); w.put(5u, 8); w.put(5u, 8); w.put(5u, 8);
    w.put((uint32_t)g.W, 14); w.put(1u, 1); w.put((uint32_t)g.H, 14);
    w.align();
}

// ----------------------------------------------------------------------------------------------
// k_assemble: one workgroup per slice, one THREAD per macroblock.  A slice is [slice header] then per macroblock
// p1 A p2 B p3 C, MSB first.  The stored segments of the slice are staged in LDS (only the filled 16-byte chunks of the
// compact slots are fetched, coalesced); every thread then ORs its macroblock's pieces, shifted to their bit offset
// (k_slice_scan), into an LDS image of the slice with ds_or; the image goes out with coalesced dword stores.  A slice is
// byte aligned, not word aligned: only its first and last word can be shared with a neighbour (previous slice,
// headers) and use atomics.  Slices longer than the 16 KB image take several passes over it.
// (Round 1-2 had one thread per OUTPUT word, which finds its macroblocks by binary search and looks at all six
// pieces of each: 385 vector instructions per output word, 57 us per 90 frames; this form needs about a sixth.)
// ----------------------------------------------------------------------------------------------
// where the compact slot of macroblock `mb` lives, by the number of words it stores (k_mb's three compact classes)
__device__ __forceinline__ const uint32_t *compact_slot(const uint32_t *slots_small, const Geom &g, size_t mb, uint32_t nwords)
{
    return nwords <= (uint32_t)kMicroSlotWords ? slots_small + g.s8_off + mb * kMicroSlotWords
           : nwords <= (uint32_t)kTinySlotWords ? slots_small + g.s16_off + mb * kTinySlotWords
                                                 : slots_small + mb * kSmallSlotWords;
}

constexpr int kAsmThreads = 128;          // >= macroblocks per row (W <= 2048)
constexpr int kAsmImageWords = 1024;      // 4 KB image: a P-frame slice in one pass, an I-frame slice in a few (LDS bounds the occupancy
                                          // of this latency-bound kernel: 9 KB per workgroup = 16 workgroups, all 32 wavefronts, per CU)
constexpr int kAsmStageWords = 1024;      // packed staging of the slice's compact slots (16-byte chunks); what does not fit is read from memory

// OR the low `len` bits of `code` (len <= 32) into the MSB-first image at bit `pos`; words outside [0, nw) are dropped
__device__ __forceinline__ void asm_put(uint32_t *img, int nw, int pos, uint32_t code, int len)
{
    if (len <= 0) return;
    const int w = pos >> 5;
    if (w < -1 || w >= nw) return;
    const int b = (int)(pos & 31);
    const unsigned long long v = (unsigned long long)code << (64 - len - b);
    const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    if (w >= 0 && hi) atomicOr(&img[w], hi);
    if (w + 1 < nw && lo) atomicOr(&img[w + 1], lo);
}

// (eight wavefronts per SIMD = 16 workgroups per CU: 51 registers with the staging in two rounds; one round of eight 16-byte loads per thread
// needs 74 = six wavefronts, and fits eight only by spilling - 40 / 44 / 48 us per sequence in the profiled pass, item 14)
__global__ __launch_bounds__(kAsmThreads, 8) void k_assemble(const FrameJob *__restrict__ jobs, Geom g, int nframes,
                                                 const uint32_t *__restrict__ mbinfo, const MbAux *__restrict__ mbaux,
                                                 const uint32_t *__restrict__ slots_small, const uint32_t *__restrict__ slots,
                                                 const unsigned long long *__restrict__ slice_off,
                                                 uint32_t *__restrict__ out32, const StreamCtl *__restrict__ ctl,
                                                 int first, int last, const unsigned long long *__restrict__ frame_off,
                                                 const uint32_t *__restrict__ slice_bytes)
{
    __shared__ uint32_t s_img[kAsmImageWords];
    __shared__ uint32_t s_slot[kAsmStageWords];
    __shared__ uint32_t s_nw[kAsmThreads];                 // stored words of every macroblock (compact slots only)
    __shared__ uint32_t s_so[kAsmThreads];                 // where its chunks start in s_slot (inclusive scan of the chunk words)
    __shared__ uint32_t s_wave0, s_bits[2];
    static_assert(kAsmThreads == 128, "two wavefronts: the scans below");
    static_assert(kAsmThreads * 5 <= kAsmImageWords, "the neighbour exchange borrows the image");
    const int tid = threadIdx.x;
    const int rows = g.row1 - g.row0;
    const int f = blockIdx.x / rows, by = g.row0 + (int)(blockIdx.x % rows);
    if (f >= nframes) return;
    const size_t base = ((size_t)f * g.mbh + by) * g.mbw;
    const bool have = tid < g.mbw;
    uint32_t info = 0;
    MbAux aux{0, 0, 0, 0};
    if (have) { info = mbinfo[base + tid]; aux = mbaux[base + tid]; }
    const int i_frame = jobs[f].i_frame;
    const unsigned long long q = (ctl->base_bytes + slice_off[(size_t)f * g.mbh + by]) * 8ull;
    const bool overflow = ctl->overflow != 0;
    // the left neighbour's word and record, through LDS (the image is not in use yet): the neighbour-dependent codes - motion vector
    // deltas, DC differentials (RTL:2736-2748, 2808-2821) - are formed here, as k_slice_scan formed them for their lengths
    uint32_t *const s_x = s_img;
    s_x[tid] = info; s_x[128 + tid] = aux.w0; s_x[256 + tid] = aux.w1; s_x[384 + tid] = aux.w2; s_x[512 + tid] = aux.w3;
    const int lenA = (int)(aux.w0 & 0xFFFFu), lenB = (int)(aux.w0 >> 16), lenC = (int)(aux.w1 & 0xFFFFu);
    const uint32_t nwords = (uint32_t)(lenA + lenB + lenC + 31) >> 5;
    const bool small = nwords <= (uint32_t)kSmallSlotWords;
    s_nw[tid] = have && small ? nwords : 0u;
    // inclusive scan of the staged words: inside each wavefront by DPP, the first wavefront's total through LDS
    const uint32_t scan = (uint32_t)wave_scan_incl((int)(have && small ? (nwords + 3u) & ~3u : 0u));
    if (tid == 63) s_wave0 = scan;
    __syncthreads();
    if (overflow) return;
    const bool has_left = tid > 0;
    const int tl = has_left ? tid - 1 : 0;
    const uint32_t linfo = s_x[tl];
    const MbAux laux{s_x[128 + tl], s_x[256 + tl], s_x[384 + tl], s_x[512 + tl]};
    const MbDep dep = mb_dependent(info, aux, has_left, linfo, laux, i_frame);
    const uint32_t c1 = dep.p1.code, c2 = dep.p2.code, c3 = dep.p3.code;
    const int l1 = have ? (int)dep.p1.len : 0, l2 = have ? (int)dep.p2.len : 0, l3 = have ? (int)dep.p3.len : 0;
    // bit offset of every macroblock inside the slice: the same scan over the macroblocks' total lengths
    const uint32_t mylen = have ? mb_total_bits(dep, aux, tid == 0) : 0u;
    const uint32_t bscan = (uint32_t)wave_scan_incl((int)mylen);
    if ((tid & 63) == 63) s_bits[tid >> 6] = bscan;
    s_so[tid] = scan + (tid >= 64 ? s_wave0 : 0u);
    __syncthreads();
    const uint32_t off = bscan - mylen + (tid >= 64 ? s_bits[0] : 0u);
    const uint32_t total = s_bits[0] + s_bits[1];          // the slice's bits
    // stage the compact slots: up to kSlotChunks chunks of 16 bytes per macroblock, only the filled ones, packed.  The loads of a
    // round are issued before its first LDS store (two memory round trips, not one per chunk).
    {
        constexpr int kRounds = 2, kIter = kSlotChunks / kRounds;     // 128 macroblocks x kSlotChunks chunks / 128 threads, in two rounds (registers: see the launch bounds)
#pragma unroll
        for (int h = 0; h < kRounds; ++h) {
            uint4 v[kIter];
            uint32_t dst[kIter];
#pragma unroll
            for (int i = 0; i < kIter; ++i) {
                const int idx = tid + (h * kIter + i) * kAsmThreads, m = idx / kSlotChunks, c = idx % kSlotChunks;
                const uint32_t nwm = s_nw[m], end = s_so[m];   // end: one past the macroblock's last staged word
                const bool take = m < g.mbw && (uint32_t)(4 * c) < nwm && end <= (uint32_t)kAsmStageWords;
                dst[i] = take ? end - ((nwm + 3u) & ~3u) + 4u * (uint32_t)c : 0xFFFFFFFFu;
                const uint32_t *const src = compact_slot(slots_small, g, base + m, nwm);
                v[i] = take ? *(const uint4 *)(src + 4 * c) : uint4{0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < kIter; ++i)
                if (dst[i] != 0xFFFFFFFFu) {
                    uint32_t *d = &s_slot[dst[i]];
                    d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
                }
        }
    }
    const int sh = (int)(q & 31ull);
    const unsigned long long w0 = q >> 5;
    const uint32_t nout = ((uint32_t)sh + total + 31u) / 32u;
    // this thread's segments: staged, or in memory (the overflow slot, or a compact slot that found no room in the staging)
    const bool staged = small && s_so[tid] <= (uint32_t)kAsmStageWords;
    const uint32_t *const big = !small ? slots + (base + tid) * kSlotWords : compact_slot(slots_small, g, base + tid, nwords);
    const uint32_t *const stg = &s_slot[staged ? s_so[tid] - ((nwords + 3u) & ~3u) : 0u];

    for (uint32_t c0 = 0; c0 < nout; c0 += kAsmImageWords) {
        const int nw = (int)(nout - c0 < (uint32_t)kAsmImageWords ? nout - c0 : (uint32_t)kAsmImageWords);
        for (int k = tid; k < nw; k += kAsmThreads) s_img[k] = 0u;
        __syncthreads();                                   // image cleared, slots staged
        if (have) {
            int pos = sh + (int)off - 32 * (int)c0;        // this macroblock's first bit in the image (a slice is < 2^21 bits)
            if (tid == 0) {
                // slice header: start code, slice_vertical_position, quantiser_scale_code, extra_bit_slice (RTL:2708-2710)
                asm_put(s_img, nw, pos, 0x000001u, 24);
                asm_put(s_img, nw, pos + 24, ((uint32_t)(by + 1) << 6) | (2u << g.Q), 14);
                pos += 38;
            }
            const uint32_t codes[3] = {c1, c2, c3};
            const int clen[3] = {l1, l2, l3}, slen[3] = {lenA, lenB, lenC};
            int src = 0;                                   // bit offset of the segment inside the slot
#pragma unroll
            for (int sgm = 0; sgm < 3; ++sgm) {
                asm_put(s_img, nw, pos, codes[sgm], clen[sgm]);
                pos += clen[sgm];
                const int n = slen[sgm];
                // words of the segment that can touch the image: bits [pos, pos + n) against [0, 32 nw)
                if (n > 0 && pos + n > 0 && pos < 32 * nw) {
                    for (int b = 0; b < n; b += 32) {
                        const int sb = src + b, valid = n - b < 32 ? n - b : 32;
                        const uint32_t a0 = staged ? stg[sb >> 5] : big[sb >> 5];
                        uint32_t w = a0 << (sb & 31);
                        if ((sb & 31) && (sb & 31) + valid > 32) w |= (staged ? stg[(sb >> 5) + 1] : big[(sb >> 5) + 1]) >> (32 - (sb & 31));
                        asm_put(s_img, nw, pos + b, w >> (32 - valid), valid);
                    }
                }
                pos += n;
                src += n;
            }
        }
        __syncthreads();
        for (int k = tid; k < nw; k += kAsmThreads) {
            const uint32_t gk = c0 + (uint32_t)k;
            const uint32_t be = __builtin_bswap32(s_img[k]);
            if (gk == 0 || gk == nout - 1) {
                // a slice is byte aligned, not word aligned: of its first and its last word only the bytes that are its own are
                // written, one by one (the others belong to the previous / next slice or to headers).  No atomics and nothing to
                // clear beforehand.
                const int j0 = gk == 0 ? sh >> 3 : 0;
                const int j1 = gk == nout - 1 ? (int)((((uint32_t)sh + total + 7u) >> 3) - 1u) & 3 : 3;      // last byte of the slice inside this word
                uint8_t *const o8 = (uint8_t *)&out32[w0 + gk];
                for (int j = j0; j <= j1; ++j) o8[j] = (uint8_t)(be >> (8 * j));
            } else out32[w0 + gk] = be;
        }
        __syncthreads();                                   // the image is reused by the next pass
    }
    // The frame's headers travel with its first slice and the sequence end code with the very last one: plain byte
    // stores by one thread.  A header byte may share a dword with a slice's boundary word; that dword was cleared by
    // k_frame_scan, the atomic OR above only adds the slice's own bits and the byte store only touches its byte.
    if (!g.strip && tid == 0) {
        uint8_t *const out8 = (uint8_t *)out32 + ctl->base_bytes;
        if (by == g.row0) {
            if (first && f == 0) write_sequence_headers(out8, g);
            write_frame_headers(out8 + frame_off[f], jobs[f]);
        }
        if (last && f == nframes - 1 && by == g.row1 - 1) {
            uint8_t *e = out8 + slice_off[(size_t)f * g.mbh + by] + slice_bytes[(size_t)f * g.mbh + by];
            e[0] = 0x00; e[1] = 0x00; e[2] = 0x01; e[3] = 0xB7;       // sequence_end_code (RTL:2625-2628)
        }
    }
}

// ----------------------------------------------------------------------------------------------
// k_frame_scan: byte offsets of frames and slices inside the chunk; total stream length
// ----------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t frame_header_bytes(int i_frame)
{
    return (i_frame == 0 ? kGopHeaderBytes + 17u : 18u);     // RTL:2670-2682
}

// One block; thread t owns K consecutive (frame, slice) items, an item = a slice preceded by its frame's
// headers when it is the first slice of the frame: local sums -> block scan -> offsets.
// ctl_init: the control word starts here instead of in a kernel of its own (a launch in front of every chunk - and, in strip mode, on the
// tail of every sequence): 1 = a new stream (nothing precedes this chunk), 2 = this chunk's bytes leave in a buffer of their own but
// continue the previous chunk's stream (the port path: prior = everything so far, only the final padding rule needs it), 0 = the word is
// as the previous chunk of this call left it.  ctl_cap: capacity of the output buffer (modes 1, 2).
// px (strip mode, peer transport; all null / 0 otherwise): a sequence in which some wait ran out of budget is marked in the strip's size
// table ("encode it again, the ordinary way") where every rank sees it after the all-gather; the give-up word is cleared for the next
// sequence, and so are the arrival counters of the NEXT sequence's set (nobody touches those before this rank has contributed to this
// sequence's all-gather, which comes behind this kernel on the stream).
struct PeerScan { unsigned int *gaveup; unsigned int *clear; int clear_lines; unsigned long long mark; };

__global__ __launch_bounds__(1024) void k_frame_scan(const FrameJob *__restrict__ jobs, Geom g, int nframes, int first, int last,
                                                     const uint32_t *__restrict__ slice_bytes,
                                                     unsigned long long *__restrict__ slice_off,
                                                     unsigned long long *__restrict__ frame_off, StreamCtl *ctl,
                                                     int advance, uint32_t *__restrict__ out32, int ctl_init, unsigned long long ctl_cap, PeerScan px)
{
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_wtot[16];
    const int tid = threadIdx.x;
    // requested with everything else: this single workgroup is pure latency, and these were a round trip of their own after the scan
    unsigned long long c_prior = 0, c_cap = ctl_cap & ~3ull;
    uint32_t c_ov = 0;
    if (ctl_init == 0) { c_prior = ctl->prior_bytes; c_cap = ctl->cap_bytes; c_ov = ctl->overflow; }
    else if (ctl_init == 2) c_prior = ctl->prior_bytes + ctl->total_bytes;
    if (tid == 0) {
        // advance = this chunk continues the stream of the previous one in the same buffer: base = previous total
        unsigned long long b = 0;
        if (ctl_init == 0) {
            b = ctl->base_bytes;
            if (advance && !ctl->overflow) { b = ctl->total_bytes; ctl->base_bytes = b; }
        }
        s_base = b;
    }
    for (int i = tid; i < px.clear_lines; i += 1024) px.clear[i * 32] = 0u;        // one counter per 128-byte line
    const int rows = g.row1 - g.row0;
    const int S = nframes * rows;
    const int K = (S + 1023) / 1024;
    const int i0 = tid * K, i1 = i0 + K < S ? i0 + K : S;
    auto header_bytes = [&](int f) -> unsigned long long {     // bytes in front of the first slice of frame f
        if (g.strip) return 0ull;
        return (unsigned long long)frame_header_bytes(jobs[f].i_frame) + (first && f == 0 ? kSeqHeaderBytes : 0u);
    };
    // The thread's first kCached items are fetched in one go (all loads in flight together: this single workgroup is pure
    // latency) and kept for the second pass; longer lists (more than 8192 slices in a chunk) walk the rest one by one.
    constexpr int kCached = 8;
    const int f0 = i0 / rows, r0 = i0 - f0 * rows;
    uint32_t sbv[kCached], hbv[kCached];
    {
        int f = f0, r = r0;
#pragma unroll
        for (int j = 0; j < kCached; ++j) {
            const bool in = i0 + j < i1;
            sbv[j] = in ? slice_bytes[(size_t)f * g.mbh + g.row0 + r] : 0u;
            hbv[j] = in && r == 0 ? (uint32_t)header_bytes(f) : 0u;
            if (++r == rows) { r = 0; ++f; }
        }
    }
    unsigned long long sum = 0;
#pragma unroll
    for (int j = 0; j < kCached; ++j) sum += sbv[j] + hbv[j];
    for (int i = i0 + kCached; i < i1; ++i) {
        const int f = i / rows, r = i - f * rows;
        sum += slice_bytes[(size_t)f * g.mbh + g.row0 + r];
        if (r == 0) sum += header_bytes(f);
    }
    // block scan of the 64-bit sums: inside a wavefront by DPP - which moves 32 bits, so the sum goes as three parts whose
    // wavefront totals cannot wrap (bits 0-15, bits 16-31, the rest) and is put together again afterwards - then the 16
    // wavefront totals through LDS: one barrier instead of the twenty of a Hillis-Steele scan over 1024 threads
    const unsigned long long wscan = (unsigned long long)(uint32_t)wave_scan_incl((int)((uint32_t)sum & 0xFFFFu)) +
                                     ((unsigned long long)(uint32_t)wave_scan_incl((int)(((uint32_t)sum >> 16) & 0xFFFFu)) << 16) +
                                     ((unsigned long long)(uint32_t)wave_scan_incl((int)(uint32_t)(sum >> 32)) << 32);
    if ((tid & 63) == 63) s_wtot[tid >> 6] = wscan;
    __syncthreads();
    unsigned long long before = 0, grand = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned long long t = s_wtot[w];
        before += w < (tid >> 6) ? t : 0ull;
        grand += t;
    }
    const unsigned long long incl = before + wscan;
    // stream length and overflow: every thread derives them from the grand total (thread 1023 publishes them)
    const unsigned long long base = s_base, all_frames = grand;
    unsigned long long total = base + all_frames;
    if (last) {
        total += 4;                                           // sequence_end_code (RTL:2621-2628)
        const unsigned long long all = c_prior + total;
        total = (all / 32ull + 1ull) * 32ull - c_prior;                // final word always leaves (RTL:2932-2937)
    }
    const bool ov = total > c_cap || c_ov;
    // k_assemble writes every byte of every slice and header itself (whole dwords inside a slice, single bytes in its first and
    // last word), so nothing of the stream has to be cleared beforehand except the tail: end code + padding of the final
    // 32-byte word.
    unsigned long long run = incl - sum;
    auto place = [&](int f, int r, uint32_t sb, uint32_t hb) {
        if (r == 0) {
            frame_off[f] = run + (!g.strip && first && f == 0 ? kSeqHeaderBytes : 0u);   // the frame's own headers start here
            run += hb;
        }
        slice_off[(size_t)f * g.mbh + g.row0 + r] = run;
        run += sb;
    };
    {
        int f = f0, r = r0;
#pragma unroll
        for (int j = 0; j < kCached; ++j) {
            if (i0 + j < i1) place(f, r, sbv[j], hbv[j]);
            if (++r == rows) { r = 0; ++f; }
        }
    }
    for (int i = i0 + kCached; i < i1; ++i) {
        const int f = i / rows, r = i - f * rows;
        place(f, r, slice_bytes[(size_t)f * g.mbh + g.row0 + r], r == 0 ? (uint32_t)header_bytes(f) : 0u);
    }
    __syncthreads();                                          // every thread has read ctl->overflow / cap before they change
    if (tid == 1023) {
        frame_off[nframes] = all_frames;                      // one past the end (strip mode reads the sizes back)
        if (px.gaveup) {
            if (*px.gaveup) frame_off[nframes] = px.mark;
            *px.gaveup = 0u;
        }
        if (ctl_init) { ctl->base_bytes = 0; ctl->cap_bytes = c_cap; ctl->prior_bytes = c_prior; ctl->pad = 0; }
        ctl->total_bytes = total;
        ctl->overflow = ov ? 1u : 0u;
        if (!ov && last && !g.strip)
            for (unsigned long long w = (base + all_frames) >> 2; w < (total + 3ull) >> 2; ++w) out32[w] = 0u;
    }
}

// ----------------------------------------------------------------------------------------------
// strip mode (multi-GPU, config c5): rows of the reconstruction that the neighbouring strips need as
// reference: YR luma + UR chroma rows (U and V) on each side (RTL:1446-1448 window geometry).
// One block per (frame of the step, direction); packed layout per frame: [YR*W luma][UR*cw U][UR*cw V].
// ----------------------------------------------------------------------------------------------
__global__ void k_halo_pack(const FrameJob *__restrict__ jobs, const int *__restrict__ halo_list, Geom g, int YR, int UR,
                            uint8_t *__restrict__ up, uint8_t *__restrict__ down)
{
    const int k = blockIdx.x, dir = blockIdx.y;               // dir 0: my top rows -> rank above; 1: my bottom rows -> rank below
    uint8_t *dst = dir ? down : up;
    if (!dst) return;
    const uint8_t *rec = jobs[halo_list[k]].rec;
    const uint32_t chunk = (uint32_t)(YR + UR) * (uint32_t)g.W;
    dst += (size_t)k * chunk;
    const int y0 = dir ? 16 * g.row1 - YR : 16 * g.row0, c0 = dir ? 8 * g.row1 - UR : 8 * g.row0;
    const uint32_t nY = (uint32_t)YR * g.W, nC = (uint32_t)UR * g.cw;
    for (uint32_t i = threadIdx.x; i < chunk; i += blockDim.x) {
        uint8_t v;
        if (i < nY) v = rec[rec_luma_off(i % (uint32_t)g.W, (uint32_t)y0 + i / (uint32_t)g.W, g)];
        else if (i < nY + nC) v = rec[rec_chroma_off(0u, (i - nY) % (uint32_t)g.cw, (uint32_t)c0 + (i - nY) / (uint32_t)g.cw, g)];
        else v = rec[rec_chroma_off(1u, (i - nY - nC) % (uint32_t)g.cw, (uint32_t)c0 + (i - nY - nC) / (uint32_t)g.cw, g)];
        dst[i] = v;
    }
}

__global__ void k_halo_unpack(const FrameJob *__restrict__ jobs, const int *__restrict__ halo_list, Geom g, int YR, int UR,
                              const uint8_t *__restrict__ from_up, const uint8_t *__restrict__ from_down)
{
    const int k = blockIdx.x, dir = blockIdx.y;               // dir 0: rows above my strip (from the rank above); 1: rows below
    const uint8_t *src = dir ? from_down : from_up;
    if (!src) return;
    uint8_t *rec = jobs[halo_list[k]].rec;
    const uint32_t chunk = (uint32_t)(YR + UR) * (uint32_t)g.W;
    src += (size_t)k * chunk;
    const int y0 = dir ? 16 * g.row1 : 16 * g.row0 - YR, c0 = dir ? 8 * g.row1 : 8 * g.row0 - UR;
    const uint32_t nY = (uint32_t)YR * g.W, nC = (uint32_t)UR * g.cw;
    for (uint32_t i = threadIdx.x; i < chunk; i += blockDim.x) {
        const uint8_t v = src[i];
        if (i < nY) rec[rec_luma_off(i % (uint32_t)g.W, (uint32_t)y0 + i / (uint32_t)g.W, g)] = v;
        else if (i < nY + nC) rec[rec_chroma_off(0u, (i - nY) % (uint32_t)g.cw, (uint32_t)c0 + (i - nY) / (uint32_t)g.cw, g)] = v;
        else rec[rec_chroma_off(1u, (i - nY - nC) % (uint32_t)g.cw, (uint32_t)c0 + (i - nY - nC) / (uint32_t)g.cw, g)] = v;
    }
}

// ----------------------------------------------------------------------------------------------
// Final assembly of strip mode on the rank that owns the output (m2v_strip_assemble / m2v_strip_encode).  The stream is
//   sequence headers | per frame: [GOP header] picture headers, strip 0's slices, strip 1's, ... | end code + padding
// k_strip_layout (one block) turns the per-rank frame offsets - device memory: they come out of k_frame_scan, or out of the
// all-gather of the other ranks' - into one copy segment per (frame, rank) and the stream length, so the host neither reads the
// sizes nor uploads a table in between; k_strip_assemble then moves the bytes with 16-byte stores, writes the headers and the
// trailer in the same launch.
// ----------------------------------------------------------------------------------------------
constexpr int kLayoutThreads = 256;

// all_off: [nranks][nframes + 1] byte offsets of every frame inside its rank's strip buffer.  Outputs: segs[f * nranks + r],
// frame_pos[f] = where frame f's own headers start, the control word (total_bytes, overflow against `cap`).
__global__ __launch_bounds__(kLayoutThreads) void k_strip_layout(const unsigned long long *__restrict__ all_off, int nranks, int nframes, uint32_t gop,
                                                                   StripSrc src, CopySeg *__restrict__ segs,
                                                                   unsigned long long *__restrict__ frame_pos, StreamCtl *ctl, unsigned long long cap)
{
    __shared__ unsigned long long s_scan[kLayoutThreads];
    __shared__ unsigned long long s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = kSeqHeaderBytes;
    __syncthreads();
    for (int f0 = 0; f0 < nframes; f0 += kLayoutThreads) {
        const int f = f0 + tid;
        unsigned long long sz = 0;
        if (f < nframes) {
            sz = frame_header_bytes((int)((uint32_t)f % gop));
            for (int r = 0; r < nranks; ++r) sz += all_off[(size_t)r * (nframes + 1) + f + 1] - all_off[(size_t)r * (nframes + 1) + f];
        }
        s_scan[tid] = sz;
        __syncthreads();
        for (int o = 1; o < kLayoutThreads; o <<= 1) {
            const unsigned long long t = tid >= o ? s_scan[tid - o] : 0ull;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        const unsigned long long carry = s_carry;
        if (f < nframes) {
            unsigned long long pos = carry + s_scan[tid] - sz;
            frame_pos[f] = pos;
            pos += frame_header_bytes((int)((uint32_t)f % gop));
            for (int r = 0; r < nranks; ++r) {
                const unsigned long long a = all_off[(size_t)r * (nframes + 1) + f], b = all_off[(size_t)r * (nframes + 1) + f + 1];
                segs[(size_t)f * nranks + r] = CopySeg{src.strip[r] + a, pos, b - a};
                pos += b - a;
            }
        }
        __syncthreads();
        if (tid == kLayoutThreads - 1) s_carry = carry + s_scan[tid];
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long body = s_carry;
        const unsigned long long total = ((body + 4ull) / 32ull + 1ull) * 32ull;       // end code + the final word rule (RTL:2932-2937)
        frame_pos[nframes] = body;
        ctl->base_bytes = 0;
        ctl->total_bytes = total;
        ctl->cap_bytes = cap & ~3ull;
        ctl->prior_bytes = 0;
        ctl->overflow = total > (cap & ~3ull) ? 1u : 0u;
        ctl->pad = 0;
    }
}

// Blocks [0, nsegs * split): segment b / split, part b % split of its 16-byte destination cells; 16-byte stores to aligned cells,
// the source read as aligned dwords and shifted into place (segments start at any byte on both sides); the partial cells at
// a segment's ends go byte by byte (a neighbouring segment owns the other bytes of that cell).  Then ceil(nframes / 256)
// header blocks (one thread per frame; thread 0 also the sequence headers) and one trailer block.
constexpr int kCopyThreads = 256;
__global__ __launch_bounds__(kCopyThreads) void k_strip_assemble(const CopySeg *__restrict__ segs, int nsegs, int split, Geom g, int nframes, uint32_t gop,
                                                                   const unsigned long long *__restrict__ frame_pos,
                                                                   uint8_t *__restrict__ out, const StreamCtl *__restrict__ ctl)
{
    if (ctl->overflow) return;
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const int ncopy = nsegs * split;
    if (b < ncopy) {
        const CopySeg sg = segs[b / split];
        if (sg.len == 0) return;
        const int part = b % split;
        const unsigned long long d0 = sg.dst_off, d1 = sg.dst_off + sg.len;
        const unsigned long long c0 = d0 >> 4, c1 = (d1 + 15ull) >> 4;                   // destination cells [c0, c1)
        const unsigned long long nc = c1 - c0, per = (nc + (unsigned long long)split - 1ull) / (unsigned long long)split;
        const unsigned long long ca = c0 + per * (unsigned long long)part, cb = ca + per < c1 ? ca + per : c1;
        for (unsigned long long c = ca + (unsigned long long)tid; c < cb; c += kCopyThreads) {
            const unsigned long long lo = c << 4;
            if (lo >= d0 && lo + 16ull <= d1) {
                const uint8_t *sp = sg.src + (lo - d0);
                const uint32_t sh = (uint32_t)((uintptr_t)sp & 3u);
                const uint32_t *sw = (const uint32_t *)((uintptr_t)sp & ~(uintptr_t)3);
                const uint32_t w0 = sw[0], w1 = sw[1], w2 = sw[2], w3 = sw[3];
                const uint32_t w4 = sh ? sw[4] : 0u;                                      // holds source bytes of this cell whenever it is read
                uint4 v;
                v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
                v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
                *(uint4 *)(out + lo) = v;
            } else {
                const unsigned long long a = lo > d0 ? lo : d0, e = lo + 16ull < d1 ? lo + 16ull : d1;
                for (unsigned long long i = a; i < e; ++i) out[i] = sg.src[i - d0];
            }
        }
        return;
    }
    const int hb = b - ncopy, nhb = (nframes + kCopyThreads - 1) / kCopyThreads;
    if (hb < nhb) {
        const int f = hb * kCopyThreads + tid;
        if (f < nframes) {
            FrameJob job{};
            job.i_frame = (int32_t)((uint32_t)f % gop);
            job.n = (uint32_t)f;
            write_frame_headers(out + frame_pos[f], job);
        }
        if (f == 0) write_sequence_headers(out, g);
        return;
    }
    // trailer: sequence_end_code + zero padding up to the stream length (RTL:2621-2628, 2932-2937)
    const unsigned long long body = frame_pos[nframes], total = ctl->total_bytes;
    for (unsigned long long i = body + (unsigned long long)tid; i < total; i += kCopyThreads) {
        const unsigned long long r = i - body;
        out[i] = r == 2 ? 0x01 : r == 3 ? 0xB7 : 0x00;
    }
}

}  // namespace m2v
