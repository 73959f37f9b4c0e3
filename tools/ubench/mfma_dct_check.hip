// mfma_dct_check.hip - the 16x16 luma block of a macroblock through the MFMA formulation of the RTL's 2-D integer DCT
// (RTL:2029-2062), checked against the plain integer definition.  Development aid for k_mb's DCT-as-GEMM trial:
//   four 8x8 transforms of the 2x2 tiles = B16 . Z . B16^T with B16 = blockdiag(M, M), Z = current - prediction
//   pass 1: T = Z . B16^T        one v_mfma_i32_16x16x32_i8, K = 16 current columns + 16 prediction columns (-B16)
//   pass 2: Y = B16 . T          T is 19 bit: three signed byte limbs, one MFMA each, recombined with two shift-adds
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_dct_check.hip -o /tmp/mfma_dct_check && /tmp/mfma_dct_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
static const int8_t kM[64] = {64, 64, 64, 64, 64, 64, 64, 64, 89, 75, 50, 18, -18, -50, -75, -89, 84, 35, -35, -84, -84, -35, 35, 84,
                              75, -18, -89, -50, 50, 89, 18, -75, 64, -64, -64, 64, 64, -64, -64, 64, 50, -89, 18, 75, -75, -18, 89, -50,
                              35, -84, 84, -35, -35, 84, -84, 35, 18, -50, 75, -89, 89, -75, 50, -18};
__constant__ int8_t c_m[64];

__global__ __launch_bounds__(64) void k_dct(const uint8_t *cur, const uint8_t *pred, int *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_cp[4][8][16];       // [tile][row][8 current | 8 prediction], signed bytes
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    {   // fill like k_mb does: lane = (row r, 4-pixel group c4)
        const int r = lane >> 2, c4 = lane & 3;
        const uint32_t cu = *(const uint32_t *)(cur + r * 16 + 4 * c4) ^ 0x80808080u, pr = *(const uint32_t *)(pred + r * 16 + 4 * c4) ^ 0x80808080u;
        const int tile = ((r >> 3) << 1) | (c4 >> 1);
        *(uint32_t *)&s_cp[tile][r & 7][(c4 & 1) << 2] = cu;
        *(uint32_t *)&s_cp[tile][r & 7][8 + ((c4 & 1) << 2)] = pr;
    }
    __syncthreads();
    // ---- pass 1 ----
    // A1: lane (g, r): row r of the block, bytes = 8 consecutive k: g = 0/1: current columns 8(g&1) .., g = 2/3: prediction columns
    const long a1 = *(const long *)&s_cp[((c >> 3) << 1) | (g & 1)][c & 7][8 * (g >> 1)];
    // B1: lane (g, col c): B1[k][c] = +-B16[c][k]: the basis row (c & 7) where the 8-column group of k matches c's tile column
    long b1 = 0;
    if ((c >> 3) == (g & 1)) {
        long m = *(const long *)&c_m[(c & 7) * 8];
        if (g >= 2) {       // negate each byte (|m| <= 89)
            long n = 0;
            for (int b = 0; b < 8; ++b) n |= (long)(uint8_t)(-(int8_t)(m >> (8 * b))) << (8 * b);
            m = n;
        }
        b1 = m;
    }
    v4i zero = {0, 0, 0, 0};
    const v4i t = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, b1, zero, 0, 0, 0);        // t[v] = T[4g + v][c]
    // ---- pass 2 ----
    // three signed byte limbs of every T: (T + 0x808080) ^ 0x808080 has them in bytes 0..2 (T = l0 + 256 l1 + 65536 l2)
    uint32_t e[4];
    for (int v = 0; v < 4; ++v) e[v] = ((uint32_t)t[v] + 0x808080u) ^ 0x808080u;
    // 4x4 byte transpose: w[n] = limb n of e[0..3]
    const uint32_t p01 = __builtin_amdgcn_perm(e[1], e[0], 0x05010400u);      // l0(e0) l0(e1) l1(e0) l1(e1)
    const uint32_t p23 = __builtin_amdgcn_perm(e[3], e[2], 0x05010400u);
    const uint32_t w0 = __builtin_amdgcn_perm(p23, p01, 0x05040100u), w1 = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
    const uint32_t q01 = __builtin_amdgcn_perm(e[1], e[0], 0x0c0c0602u), q23 = __builtin_amdgcn_perm(e[3], e[2], 0x0c0c0602u);
    const uint32_t w2 = __builtin_amdgcn_perm(q23, q01, 0x05040100u);
    // A2: lane (g, row i): k-slot 8g + s (s < 4) <-> block row 4g + s: B16[i][4g + s]; slots 8g + 4 .. 7 unused
    uint32_t a2 = 0;
    if ((c >> 3) == (g >> 1)) a2 = *(const uint32_t *)&c_m[(c & 7) * 8 + 4 * (g & 1)];
    const v4i y0 = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a2, (long)w0, zero, 0, 0, 0);
    const v4i y1 = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a2, (long)w1, zero, 0, 0, 0);
    const v4i y2 = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)a2, (long)w2, zero, 0, 0, 0);
    for (int v = 0; v < 4; ++v) {
        const int y = y0[v] + (y1[v] << 8) + (y2[v] << 16);
        out[(4 * g + v) * 16 + c] = (y + 2048) >> 12;           // C at block position (row 4g + v, column c)
    }
}

int main()
{
    hipMemcpyToSymbol(HIP_SYMBOL(c_m), kM, 64);
    uint8_t cur[256], pred[256];
    uint8_t *d_cur, *d_pred;
    int *d_out, out[256];
    hipMalloc(&d_cur, 256); hipMalloc(&d_pred, 256); hipMalloc(&d_out, 1024);
    int bad = 0;
    for (int trial = 0; trial < 200; ++trial) {
        srand(trial);
        for (int i = 0; i < 256; ++i) {
            const int mode = trial % 4;
            cur[i] = mode == 0 ? rand() & 255 : mode == 1 ? ((i / 16 + i) & 1 ? 255 : 0) : mode == 2 ? 255 : rand() & 255;
            pred[i] = mode == 0 ? rand() & 255 : mode == 1 ? ((i / 16 + i) & 1 ? 0 : 255) : mode == 2 ? 0 : 128 + (rand() & 7);
        }
        if (trial == 5) for (int i = 0; i < 256; ++i) { cur[i] = 0; pred[i] = 255; }
        hipMemcpy(d_cur, cur, 256, hipMemcpyHostToDevice); hipMemcpy(d_pred, pred, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_dct, dim3(1), dim3(64), 0, 0, d_cur, d_pred, d_out);
        hipMemcpy(out, d_out, 1024, hipMemcpyDeviceToHost);
        for (int ty = 0; ty < 2; ++ty) for (int tx = 0; tx < 2; ++tx) {
            int r1[8][8];
            for (int r = 0; r < 8; ++r) for (int j = 0; j < 8; ++j) {
                int s = 0;
                for (int k = 0; k < 8; ++k) s += ((int)cur[(8 * ty + r) * 16 + 8 * tx + k] - (int)pred[(8 * ty + r) * 16 + 8 * tx + k]) * kM[j * 8 + k];
                r1[r][j] = s;
            }
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) {
                int s = 2048;
                for (int r = 0; r < 8; ++r) s += kM[i * 8 + r] * r1[r][j];
                const int want = s >> 12, got = out[(8 * ty + i) * 16 + 8 * tx + j];
                if (want != got && bad++ < 10) printf("trial %d tile (%d,%d) coef (%d,%d): want %d got %d\n", trial, ty, tx, i, j, want, got);
            }
        }
    }
    printf(bad ? "MISMATCHES: %d\n" : "mfma_dct_check: 200 blocks x 256 coefficients identical to the integer definition (%d mismatches)\n", bad);
    return bad != 0;
}
