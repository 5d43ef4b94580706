// conv.hip -- 3x3 / stride 1 / pad 1 convolution (forward and backward-data) for gfx950: Winograd F(2x2, 3x3) on the
// fp32 matrix pipe.
//
// Where it sits: the 3x3 convolutions of DLA-34's BasicBlocks (DGDE/model/backbone/dla_dcn.py:71-101) and of the eleven
// CenterNet head trunks (DGDE/model/head/detector_predictor.py:52-60,112-120) are 45 % of the train step's FLOPs.  The
// stock path (MIOpen's fp32 Winograd, miopenSp3AsmConv f2x3) runs them on the vector ALUs at ~97 TFLOP/s effective; the
// same algorithm on v_mfma_f32_32x32x2_f32 has a 2.25 x 157 TFLOP/s ceiling.
//
// Algorithm (Lavin & Gray): Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A per 2x2 output tile, 16 multiplies per 4 outputs.
//   * One workgroup = 8 waves = 2 tile groups x 4 transform rows (xi); it owns an 8 x 32 pixel region (64 tiles) and 64
//     output channels.  Lane = (tile l&31, channel parity l>>5): the lane computes the transformed input value
//     V[xi][nu][c][tile] = +-d[r1][c1] +- d[r1][c2] +- d[r2][c1] +- d[r2][c2] straight from the raw input window in LDS
//     (two ds_read2_b32, three adds) -- that value IS its MFMA B operand, so V is never materialised.
//   * Per chunk of 8 input channels the 10 x 40 window and the transformed-weight slab U[16][64][8] are staged in LDS,
//     double-buffered through registers (one barrier per chunk), exactly like the DCN forward tile kernel.
//   * A wave accumulates M[xi][nu] for its four nu (4 x 2 x 16 accumulators); the output transform is linear, so the
//     nu-reduction happens in registers and the xi-reduction through LDS once, after the channel loop.
// Layout notes: window row stride 48 and plane stride 481 (odd) make the stride-2 tile addressing conflict free: the 16
// tile columns hit 16 even (or odd) banks, the second tile row lands 32 banks further, the other lane half (next
// channel) on the opposite parity.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dcd_hip.h"
#include "tuning_env.h"
#include "lds_limit.h"
#include "bf16_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int WN_CH = 8;                          // input channels per chunk
// output channels per workgroup: 32 NB (NB = 2; NB = 1 for layers of at most 32 outputs -- DCN's 27-channel offset convs)
constexpr int WN_NT = 512;

// Region geometry: a tile group (4 waves, one per transform row) owns TR x TC Winograd tiles (<= 32: lane & 31 = tile), the
// workgroup's two groups sit on top of each other, so the region is 4 TR x 2 TC pixels.
//   <2,16>: 8 x 32 px (the maps whose width is a multiple of 32)
//   <3,10>: 12 x 20 px, 30 of 32 lanes live: 24x80 and 12x40 maps divide exactly (no padded columns, and 256 workgroups = one
//           per CU for 256 -> 256 @ 24x80 x 8 images where the 8 x 32 regions give 288 = two rounds)
// Window reads are ds_read_b64 (bank = dword address mod 64, the two lane halves are separate groups; ds_read2_b32 banks mod 32,
// where the tile rows of a half collide pairwise -- measured as 46 % of the LDS cycles of round 3's kernel being conflict cycles):
// a lane's four values d[3 + 2 tcol .. 6 + 2 tcol] of a window row come from three aligned pairs starting at 2 + 2 tcol.  Window
// row stride RS: 2 RS mod 64 is the bank step between tile rows -- 32 for 16 tile columns (32 banks each), 20 for 10 (banks
// 2-21 / 22-41 / 42-61).  Plane and row strides are even (pairs stay aligned); with RS a multiple of 4 the staging stores are
// ds_write_b128, otherwise two ds_write_b64.
template <int TR, int TC, int NB>
struct WinoGeom {
    static constexpr int KS = 32 * NB;
    static constexpr int U = 16 * KS * WN_CH;                 // 8192 / 4096 floats: [pos][h][k][4 steps]
    static constexpr int NTILE = TR * TC;
    static constexpr int ROWS = 4 * TR + 2;                   // window rows r0-1 .. r0+4TR
    static constexpr int Q = (2 * TC + 8) / 4;                // staged dwordx4 per row: columns c0-4 .. c0+2TC+3
    static constexpr int RS = TC == 16 ? 48 : 42;
    static constexpr int PLANE = ROWS * RS;
    static constexpr int IN = WN_CH * PLANE;
    static constexpr int BUF = IN + U;
    static_assert(NTILE <= 32 && 4 * Q <= RS && RS % 2 == 0 && (2 * TC) % 4 == 0, "region geometry");
};

// Four consecutive window values starting at the ODD dword e + 1 (e even): three aligned ds_read_b64.
__device__ __forceinline__ void win_read4(const float *e, float (&q)[4])
{
    const f32x2 a = *reinterpret_cast<const f32x2 *>(e), b = *reinterpret_cast<const f32x2 *>(e + 2),
                c = *reinterpret_cast<const f32x2 *>(e + 4);
    q[0] = a.y; q[1] = b.x; q[2] = b.y; q[3] = c.x;
}

// One staged dwordx4 into a window row (16-byte aligned when the row stride is a multiple of 4, else 8).
template <int RS>
__device__ __forceinline__ void win_store4(float *d, const f32x4 v)
{
    if constexpr (RS % 4 == 0) {
        *reinterpret_cast<f32x4 *>(d) = v;
    } else {
        *reinterpret_cast<f32x2 *>(d) = f32x2{v.x, v.y};
        *reinterpret_cast<f32x2 *>(d + 2) = f32x2{v.z, v.w};
    }
}

// Workgroups are dealt to the 8 XCDs round-robin: walk contiguous runs of regions per XCD (L2 locality of the halos).
__device__ __forceinline__ void xcd_remap(int &bx, int &by)
{
    const int gx = gridDim.x, NT = gx * gridDim.y;
    const int L = bx + gx * by;
    const int xc = L & 7, slot = L >> 3;
    const int q = NT >> 3, r = NT & 7;
    const int Lp = (xc < r ? xc * (q + 1) : r * (q + 1) + (xc - r) * q) + slot;
    by = Lp / gx;
    bx = Lp - by * gx;
}

// U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// ul[z][chunk][pos = 4 xi + nu][h (2)][kk (KS)][s (4)] for output channel z*KS+kk and contraction channel chunk*8 + 2s + h
// (a lane half's ds_read_b128 of its A operands walks 16-byte slots at stride 1: no two lanes of a 16-lane group on one bank).
// mode 0: forward        (contraction over Cin:  g = w[k][c][a][b])
// mode 1: backward-data  (contraction over Cout: g = w[c][k][2-a][2-b], i.e. output channel = input channel of w)
__device__ __forceinline__ void wino_prep_body(const float *__restrict__ w, float *__restrict__ ul, int Cc, int Kk, int mode, int nchunk,
                                               int nz, int WN_KS, int first, int step)
{
    const int n = nz * nchunk * WN_KS * WN_CH;                    // one item per (z, chunk, kk, h, s): writes 16 positions
    for (int idx = first; idx < n; idx += step) {
        int r = idx;
        const int s = r & 3; r >>= 2;
        const int h = r & 1; r >>= 1;
        const int kk = r % WN_KS; r /= WN_KS;
        const int ck = r % nchunk, z = r / nchunk;
        const int k = z * WN_KS + kk, c = ck * WN_CH + 2 * s + h;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float v = 0.f;
                if (k < Kk && c < Cc)
                    v = mode == 0 ? w[(((size_t)k * Cc + c) * 3 + a) * 3 + b] : w[(((size_t)c * Kk + k) * 3 + (2 - a)) * 3 + (2 - b)];
                g[a][b] = v;
            }
        float t[4][3];                       // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            const float u[4] = {t[xi][0], 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]), 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]), t[xi][2]};
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
                ul[((((size_t)(z * nchunk + ck) * 16 + xi * 4 + nu) * 2 + h) * WN_KS + kk) * 4 + s] = u[nu];
        }
    }
}

__global__ void wino_prep_weights(const float *__restrict__ w, float *__restrict__ ul, int Cc, int Kk, int mode, int nchunk, int nz,
                                  int WN_KS)
{
    wino_prep_body(w, ul, Cc, Kk, mode, nchunk, nz, WN_KS, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Both directions of one layer in one launch (blockIdx.y = mode): the forward call of a training step prepares the weights of its
// own backward-data call as well -- they do not change in between, and a prep launch costs a dispatch however small it is.
struct PrepBoth {
    float *ul[2];
    int Cc[2], Kk[2], nchunk[2], nz[2], ks[2];
};

__global__ void wino_prep_weights_both(const float *__restrict__ w, PrepBoth p)
{
    const int mode = blockIdx.y;
    if (!p.ul[mode]) return;
    wino_prep_body(w, p.ul[mode], p.Cc[mode], p.Kk[mode], mode, p.nchunk[mode], p.nz[mode], p.ks[mode],
                   blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The same transform for MANY layers in one launch (blockIdx.z = table entry, blockIdx.y = direction): a train step prepares the
// weights of all its 3x3 convolutions right after the optimizer step that changed them -- 37 launches of a few microseconds each
// become one.  Table entry = five 64-bit words {weight, forward_out, backward_out, Cin, Cout} (include/dcd_hip.h).
__global__ void wino_prep_weights_table(const long long *__restrict__ table)
{
    const long long *e = table + 5 * (size_t)blockIdx.z;
    const float *w = reinterpret_cast<const float *>(e[0]);
    const int mode = blockIdx.y;
    float *ul = reinterpret_cast<float *>(e[1 + mode]);
    if (!w || !ul) return;
    const int Cin = (int)e[3], Cout = (int)e[4];
    const int Cc = mode ? Cout : Cin, Kk = mode ? Cin : Cout;
    const int WN_KS = Kk <= 32 ? 32 : 64;
    wino_prep_body(w, ul, Cc, Kk, mode, (Cc + WN_CH - 1) / WN_CH, (Kk + WN_KS - 1) / WN_KS, WN_KS,
                   blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// grid = (regions, B, K/64 * ksplit); block = 512.  x: (B, Cc, H, W) -> y: (B, Kk, H, W).
// ksplit > 1: split ks contracts the chunks [ks nchunk / ksplit, (ks+1) nchunk / ksplit); split 0 writes y, split ks >= 1 the
// partial image part + (ks-1) B Kk H W, wino_sum_partials adds them in a fixed order.
// bias (may be null): added to the output channels by split 0; residual (may be null, may be y itself): a (B, Kk, H, W) image
// added to the result by split 0 (a gradient that is already there: the caller's accumulation without a separate pass).
template <int TR, int TC, int NB>
__global__ __launch_bounds__(WN_NT) void wino_conv3x3_f32(const float *__restrict__ x, const float *__restrict__ ul, float *y,
                                                          float *__restrict__ part, const float *__restrict__ bias,
                                                          const float *residual, int Cc, int H, int W, int Kk, int tiles_x,
                                                          int nchunk, int nz)
{
    using G = WinoGeom<TR, TC, NB>;
    constexpr int WN_ROWS = G::ROWS, WN_RS = G::RS, WN_PLANE = G::PLANE, WN_IN = G::IN, WN_BUF = G::BUF, WN_Q = G::Q;
    constexpr int WN_KS = G::KS, WN_U = G::U;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x WN_BUF (epilogue: 8 x 32 x 64 exchange)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int xi = wave & 3, tg = wave >> 2;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int r0 = ty * (4 * TR), c0 = tx * (2 * TC);
    const int z = blockIdx.z % nz, ks = blockIdx.z / nz, ksplit = gridDim.z / nz;
    const int ck0 = (int)((int64_t)ks * nchunk / ksplit), ck1 = (int)((int64_t)(ks + 1) * nchunk / ksplit);
    const int HW = H * W;
    const bool live = p < G::NTILE;                       // idle lanes (TR TC < 32) read tile 0's window and store nothing
    const int pe = live ? p : 0;
    const int trow = pe / TC, tcol = pe - trow * TC;

    // B^T row pair and signs of this wave's xi:  0: d0 - d2   1: d1 + d2   2: -d1 + d2   3: d1 - d3
    const int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    const float sa = xi == 2 ? -1.f : 1.f, sb = (xi == 0 || xi == 3) ? -1.f : 1.f;
    const int lanebase = (2 * TR * tg + 2 * trow) * WN_RS + 2 + 2 * tcol + h * WN_PLANE;      // even: the pair below the lane's first value
    const int base1 = lanebase + ra * WN_RS, base2 = lanebase + rb * WN_RS;

    const float *x_b = x + (size_t)b * Cc * HW;
    const float *ul_z = ul + (size_t)z * nchunk * WN_U;

    // ---- staging map (chunk invariant): window 8 ch x ROWS x Q dwordx4 (800 / 784 items); weight slab 2048 dwordx4
    constexpr int KIN = (WN_CH * WN_ROWS * WN_Q + WN_NT - 1) / WN_NT;    // 2
    constexpr int KW = WN_U / 4 / WN_NT;                                 // 4 (2 with one output block)
    int sg[KIN], sl[KIN];
    bool sv_[KIN];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const int e = tid + WN_NT * k;
        const int ch = e / (WN_ROWS * WN_Q), rem = e - ch * (WN_ROWS * WN_Q);
        const int row = rem / WN_Q, q = rem - row * WN_Q;
        const int yy = r0 - 1 + row, xx = c0 - 4 + 4 * q;
        sv_[k] = e < WN_CH * WN_ROWS * WN_Q;
        sg[k] = (sv_[k] && yy >= 0 && yy < H && xx >= 0 && xx < W) ? ch * HW + yy * W + xx : -1;
        sl[k] = ch * WN_PLANE + row * WN_RS + 4 * q;
    }
    f32x4 rin[KIN];
    // weight slab of chunk ck -> buffer `buf`: a straight copy, so it goes global -> LDS directly (16 bytes per lane, LDS address =
    // wave base + 16 lane; no staging registers, no ds_write); the caller waits for vmcnt(0) before the barrier that publishes it
    auto issue_w = [&](int ck, float *buf) {
        const float *src = ul_z + (size_t)ck * WN_U + 4 * tid;
        float *dst = buf + WN_IN + 4 * 64 * __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < KW; ++k)
            __builtin_amdgcn_global_load_lds(src + 4 * WN_NT * k, (__attribute__((address_space(3))) void *)(dst + 4 * WN_NT * k), 16, 0, 0);
    };
    auto issue = [&](int ck) {
        const float *src = x_b + (size_t)ck * WN_CH * HW;
        const int cleft = Cc - ck * WN_CH;
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            rin[k] = zero4;
            if (sv_[k] && sg[k] >= 0 && (tid + WN_NT * k) / (WN_ROWS * WN_Q) < cleft) rin[k] = *reinterpret_cast<const f32x4 *>(src + sg[k]);
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int k = 0; k < KIN; ++k)
            if (sv_[k]) win_store4<WN_RS>(buf + sl[k], rin[k]);
    };

    f32x16 acc[4][NB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][mb][r] = 0.f;

    issue_w(ck0, lds);
    issue(ck0);
    commit(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ck = ck0; ck < ck1; ++ck) {
        const float *buf = lds + ((ck - ck0) & 1) * WN_BUF;
#ifndef WN_ABL_NOSTAGE
        if (ck + 1 < ck1) {
            issue_w(ck + 1, lds + ((ck + 1 - ck0) & 1) * WN_BUF);      // that buffer was released by the barrier that ended chunk ck - 1
            issue(ck + 1);
        }
#endif
        const float *ub = buf + WN_IN + ((xi * 4 * 2 + h) * WN_KS + p) * 4;
        const float *cp1 = buf + base1, *cp2 = buf + base2;
        // A operands of the chunk: 4 nu x 2 output blocks x 4 channel steps
        f32x4 a0[4], a1[4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            a0[nu] = *reinterpret_cast<const f32x4 *>(ub + (nu * WN_KS) * 8);
            if (NB == 2) a1[nu] = *reinterpret_cast<const f32x4 *>(ub + (nu * WN_KS) * 8 + 32 * 4);
        }
        // software pipeline over the four channel steps: the eight LDS values of step s+1 are requested before the eight
        // MFMAs of step s are issued (in-order issue: otherwise their latency is exposed once the matrix pipe drains)
        float qa[4], qb[4];
#ifndef WN_ABL_NOLDS
        win_read4(cp1, qa);
        win_read4(cp2, qb);
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // row transform once per channel step: t[b] = sa d[ra][b] + sb d[rb][b], b = 0..3 (four ds_read2_b32), then the
            // four column combinations  nu 0: t0 - t2   1: t1 + t2   2: t2 - t1   3: t1 - t3
#ifdef WN_ABL_NOLDS
            const float t0 = sa * (float)s, t1 = sb, t2 = sa + sb, t3 = (float)ck;
#else
            const float t0 = sa * qa[0] + sb * qb[0], t1 = sa * qa[1] + sb * qb[1];
            const float t2 = sa * qa[2] + sb * qb[2], t3 = sa * qa[3] + sb * qb[3];
            if (s + 1 < 4) {
                win_read4(cp1 + 2 * (s + 1) * WN_PLANE, qa);
                win_read4(cp2 + 2 * (s + 1) * WN_PLANE, qb);
            }
#endif
            const float v[4] = {t0 - t2, t1 + t2, t2 - t1, t1 - t3};
            __builtin_amdgcn_sched_barrier(0);               // keep the loads above ahead of the MFMAs below
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[nu][s], v[nu], acc[nu][0], 0, 0, 0);
                if (NB == 2) acc[nu][NB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[nu][s], v[nu], acc[nu][NB - 1], 0, 0, 0);
            }
        }
#ifndef WN_ABL_NOSTAGE
        if (ck + 1 < ck1) commit(lds + ((ck + 1 - ck0) & 1) * WN_BUF);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the next chunk's weights have landed in LDS
        __syncthreads();
    }
#ifdef WN_ABL_NOEPI
    if (acc[0][0][0] != 123.456f) return;
#endif

    // ---- output transform.  A^T = [[1,1,1,0],[0,1,-1,-1]]: over nu in registers, over xi through LDS.
    // ex[wave][j*16 + r][lane]; per mb round 8 x 32 x 64 floats = 64 KB.  Every wave finalises four of the sixteen
    // accumulator rows (r = 4 xi .. 4 xi + 3) for both output rows, so the exchange reads and the stores are balanced.
    float *ex = lds;
    float *y_b = (ks == 0 ? y : part + (size_t)(ks - 1) * gridDim.y * Kk * HW) + (size_t)b * Kk * HW;
    const float *r_b = (residual && ks == 0) ? residual + (size_t)b * Kk * HW : nullptr;
    const int orow0 = r0 + 2 * TR * tg + 2 * trow, ocol = c0 + 2 * tcol;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
        float *mine = ex + (size_t)wave * 32 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mine[r * 64] = acc[0][mb][r] + acc[1][mb][r] + acc[2][mb][r];             // j = 0
            mine[(16 + r) * 64] = acc[1][mb][r] - acc[2][mb][r] - acc[3][mb][r];      // j = 1
        }
        __syncthreads();
        const float *t0 = ex + (size_t)(tg * 4 + 0) * 32 * 64 + lane, *t1 = t0 + 32 * 64, *t2 = t1 + 32 * 64, *t3 = t2 + 32 * 64;
        if (live && ocol < W) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * xi + rr;
                const int k = z * WN_KS + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float a0 = t0[r * 64], a1 = t1[r * 64], a2 = t2[r * 64], a3 = t3[r * 64];
                const float b0 = t0[(16 + r) * 64], b1 = t1[(16 + r) * 64], b2 = t2[(16 + r) * 64], b3 = t3[(16 + r) * 64];
                if (k < Kk) {
                    const float bk = (bias && ks == 0) ? bias[k] : 0.f;
                    const size_t o = (size_t)k * HW + (size_t)orow0 * W + ocol;
                    float *dst = y_b + o;
                    f32x2 r0 = {bk, bk}, r1 = {bk, bk};
                    if (r_b) {                                       // same thread reads and writes the element: y may alias
                        if (orow0 < H) r0 += *reinterpret_cast<const f32x2 *>(r_b + o);
                        if (orow0 + 1 < H) r1 += *reinterpret_cast<const f32x2 *>(r_b + o + W);
                    }
                    if (orow0 < H) *reinterpret_cast<f32x2 *>(dst) = f32x2{a0 + a1 + a2 + r0.x, b0 + b1 + b2 + r0.y};
                    if (orow0 + 1 < H) *reinterpret_cast<f32x2 *>(dst + W) = f32x2{a1 - a2 - a3 + r1.x, b1 - b2 - b3 + r1.y};
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Split-bf16 form of the same kernel (DCD_PREC_BF16X3 for the 3x3 convolutions): the sixteen Winograd-domain products of a chunk
// run on v_mfma_f32_32x32x16_bf16 with split operands (bf16_split.h): 16 input channels per chunk, 3 x 8 matrix instructions of 32
// cycles per wave and chunk where the fp32 form issues 64 of 64 -- and the lanes' transform work (now also the splits of the
// transformed input values) overlaps with them, which it cannot do with fp32 MFMAs on this part.
//   * window: as the fp32 kernel's, 16 channels, ONE buffer (committed between two barriers);
//   * weights: prepared in split form (wino_prep_weights_split) in exactly the order the lanes read them,
//     us[z][chunk][pos 16][ob NB][hi|lo][lane 64][4 dwords], so a chunk's slab is a straight copy: global_load_lds (16 bytes per
//     lane, no staging registers) into one of TWO LDS buffers, issued a chunk ahead;
//   * lane = (tile l & 31, channel parity l >> 5): the lane's eight k-slots of an instruction are the channels 2 i + parity,
//     i = 0..7, of the chunk (the weights are prepared in the same order), so the two lane halves read planes of opposite bank
//     parity exactly like the fp32 kernel;
//   * region geometry <2,16> only (the 12 x 20 px regions' window does not fit beside two 64 KB weight buffers).
constexpr int WS_CH = 16;

template <int NB>
__device__ __forceinline__ void wino_prep_split_body(const float *__restrict__ w, unsigned *__restrict__ us, int Cc, int Kk, int mode,
                                                     int nchunk, int nz, int first, int step)
{
    constexpr int KS = 32 * NB;
    const int n = nz * nchunk * KS * 2 * 4;                       // one thread per (z, chunk, kk, parity, dword): 16 positions x (hi, lo)
    for (int idx = first; idx < n; idx += step) {
        int r = idx;
        const int d = r & 3; r >>= 2;
        const int h = r & 1; r >>= 1;
        const int kk = r % KS; r /= KS;
        const int ck = r % nchunk, z = r / nchunk;
        const int k = z * KS + kk;
        float u[2][16];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = ck * WS_CH + 2 * (2 * d + e) + h;
            float g[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float v = 0.f;
                    if (k < Kk && c < Cc)
                        v = mode == 0 ? w[(((size_t)k * Cc + c) * 3 + a) * 3 + b] : w[(((size_t)c * Kk + k) * 3 + (2 - a)) * 3 + (2 - b)];
                    g[a][b] = v;
                }
            float t[4][3];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                t[0][b] = g[0][b];
                t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
                t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
                t[3][b] = g[2][b];
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                u[e][xi * 4 + 0] = t[xi][0];
                u[e][xi * 4 + 1] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
                u[e][xi * 4 + 2] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
                u[e][xi * 4 + 3] = t[xi][2];
            }
        }
        const int ob = kk >> 5, lane = h * 32 + (kk & 31);
        unsigned *dst = us + (size_t)(z * nchunk + ck) * (16 * NB * 2 * 256);
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            unsigned hi, lo;
            sp_split_pair(u[0][pos], u[1][pos], hi, lo);
            dst[((pos * NB + ob) * 2 + 0) * 256 + lane * 4 + d] = hi;
            dst[((pos * NB + ob) * 2 + 1) * 256 + lane * 4 + d] = lo;
        }
    }
}

#include "conv_direct_bf16.inc"
#include "conv1x1_bf16.inc"
#include "conv1x1_f32.inc"
#include "conv_s2_f32.inc"

// NP = 3: split-bf16 (hi*hi + hi*lo + lo*hi).  NP = 1 (DCD_PREC_BF16): ONE bf16 product per operand pair -- both operands rounded
// to bf16 (round to nearest even), fp32 accumulate: the mixed-precision form (MODEL.FP16); the lo halves of the prepared weights
// are neither copied to LDS nor read, the transformed inputs are converted (v_cvt_pk_bf16_f32) instead of split.
template <int NB>
__global__ void wino_prep_weights_split(const float *__restrict__ w, unsigned *__restrict__ us, int Cc, int Kk, int mode, int nchunk,
                                        int nz)
{
    wino_prep_split_body<NB>(w, us, Cc, Kk, mode, nchunk, nz, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The split-form transform for MANY layers in one launch (blockIdx.z = table entry, blockIdx.y = direction), like
// wino_prep_weights_table: same five-word entries, the two output pointers being the layer's split-layout buffers.
__global__ void wino_prep_weights_split_table(const long long *__restrict__ table)
{
    const long long *e = table + 5 * (size_t)blockIdx.z;
    const float *w = reinterpret_cast<const float *>(e[0]);
    const int mode = blockIdx.y;
    unsigned *us = reinterpret_cast<unsigned *>(e[1 + mode]);
    if (!w || !us) return;
    const int Cin = (int)e[3], Cout = (int)e[4];
    const int Cc = mode ? Cout : Cin, Kk = mode ? Cin : Cout;
    const int nb = Kk <= 32 ? 1 : 2, nchunk = (Cc + WS_CH - 1) / WS_CH, nz = (Kk + 32 * nb - 1) / (32 * nb);
    const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    if (nb == 2) wino_prep_split_body<2>(w, us, Cc, Kk, mode, nchunk, nz, first, step);
    else wino_prep_split_body<1>(w, us, Cc, Kk, mode, nchunk, nz, first, step);
}

template <int NB, int NP>
__global__ __launch_bounds__(WN_NT) void wino_conv3x3_split(const float *__restrict__ x, const unsigned *__restrict__ us, float *y,
                                                            float *__restrict__ part, const float *__restrict__ bias,
                                                            const float *residual, int Cc, int H, int W, int Kk, int tiles_x,
                                                            int nchunk, int nz)
{
    constexpr int TR = 2, TC = 16;
    using G = WinoGeom<TR, TC, NB>;
    constexpr int WN_ROWS = G::ROWS, WN_RS = G::RS, WN_PLANE = G::PLANE, WN_Q = G::Q, WN_KS = G::KS;
    constexpr int WIN = (WS_CH * WN_PLANE + 3) & ~3;                  // floats of the window buffer (16-byte multiple)
    constexpr int WSLAB = 16 * NB * 2 * 256;                          // dwords of one weight buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [WIN][2][WSLAB] (epilogue: 8 x 32 x 64 exchange)
    float *win = lds;
    unsigned *wbuf = reinterpret_cast<unsigned *>(lds + WIN);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int xi = wave & 3, tg = wave >> 2;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int r0 = ty * (4 * TR), c0 = tx * (2 * TC);
    const int z = blockIdx.z % nz, ks = blockIdx.z / nz, ksplit = gridDim.z / nz;
    const int ck0 = (int)((int64_t)ks * nchunk / ksplit), ck1 = (int)((int64_t)(ks + 1) * nchunk / ksplit);
    const int HW = H * W;
    const int trow = p / TC, tcol = p - trow * TC;

    const int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    const float sa = xi == 2 ? -1.f : 1.f, sb = (xi == 0 || xi == 3) ? -1.f : 1.f;
    const int lanebase = (2 * TR * tg + 2 * trow) * WN_RS + 2 + 2 * tcol + h * WN_PLANE;      // even: the pair below the lane's first value
    const int base1 = lanebase + ra * WN_RS, base2 = lanebase + rb * WN_RS;

    const float *x_b = x + (size_t)b * Cc * HW;
    const unsigned *us_z = us + (size_t)z * nchunk * WSLAB;

    // ---- window staging map (chunk invariant): 16 ch x ROWS x Q dwordx4
    constexpr int NITEM = WS_CH * WN_ROWS * WN_Q;
    constexpr int KIN = (NITEM + WN_NT - 1) / WN_NT;                  // 4
    int sg[KIN], sl[KIN];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const int e = tid + WN_NT * k;
        const int ch = e / (WN_ROWS * WN_Q), rem = e - ch * (WN_ROWS * WN_Q);
        const int row = rem / WN_Q, q = rem - row * WN_Q;
        const int yy = r0 - 1 + row, xx = c0 - 4 + 4 * q;
        const bool ok = e < NITEM && yy >= 0 && yy < H && xx >= 0 && xx < W;
        sg[k] = ok ? ch * HW + yy * W + xx : -1;
        sl[k] = e < NITEM ? ch * WN_PLANE + row * WN_RS + 4 * q : -1;
    }
    f32x4 rin[KIN];
    auto issue_in = [&](int ck) {
        const float *src = x_b + (size_t)ck * WS_CH * HW;
        const int cleft = Cc - ck * WS_CH;
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            rin[k] = zero4;
            if (sg[k] >= 0 && (tid + WN_NT * k) / (WN_ROWS * WN_Q) < cleft) rin[k] = *reinterpret_cast<const f32x4 *>(src + sg[k]);
        }
    };
    auto commit_in = [&]() {
#pragma unroll
        for (int k = 0; k < KIN; ++k)
            if (sl[k] >= 0) win_store4<WN_RS>(win + sl[k], rin[k]);
    };
    // weight slab of chunk ck -> buffer `buf`: straight copy, 16 bytes per lane per instruction, LDS address = wave base + 16 lane
    auto issue_w = [&](int ck, int buf) {
        const unsigned *src = us_z + (size_t)ck * WSLAB + tid * 4;
        unsigned *dst = wbuf + buf * WSLAB + wave * 256;
        if (NP == 1 && (wave & 1)) return;                   // 256-dword blocks alternate hi | lo: odd waves would copy lo halves
#pragma unroll
        for (int q = 0; q < WSLAB / (WN_NT * 4); ++q)
            __builtin_amdgcn_global_load_lds(src + q * (WN_NT * 4), (__attribute__((address_space(3))) void *)(dst + q * (WN_NT * 4)), 16, 0, 0);
    };

    f32x16 acc[4][NB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][mb][r] = 0.f;

    issue_w(ck0, 0);
    issue_in(ck0);
    commit_in();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ck = ck0; ck < ck1; ++ck) {
        const int cur = (ck - ck0) & 1;
        const bool more = ck + 1 < ck1;
        if (more) {
            issue_w(ck + 1, cur ^ 1);
            issue_in(ck + 1);
        }
        const unsigned *wl = wbuf + cur * WSLAB + lane * 4;
        const float *cp1 = win + base1, *cp2 = win + base2;
        // transformed input of the lane's eight channels (2 i + parity): row transform, then the four column combinations
        float vv[4][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float q1[4], q2[4];
            win_read4(cp1 + 2 * i * WN_PLANE, q1);
            win_read4(cp2 + 2 * i * WN_PLANE, q2);
            const float t0 = sa * q1[0] + sb * q2[0], t1 = sa * q1[1] + sb * q2[1];
            const float t2 = sa * q1[2] + sb * q2[2], t3 = sa * q1[3] + sb * q2[3];
            vv[0][i] = t0 - t2; vv[1][i] = t1 + t2; vv[2][i] = t2 - t1; vv[3][i] = t1 - t3;
        }
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            if constexpr (NP == 1) {
                const sp_bf16x8 bh = sp_round8(vv[nu]);
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    const unsigned *wa = wl + (((xi * 4 + nu) * NB + ob) * 2) * 256;
                    const sp_bf16x8 ah = __builtin_bit_cast(sp_bf16x8, *reinterpret_cast<const sp_u32x4 *>(wa));
                    acc[nu][ob] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nu][ob], 0, 0, 0);
                }
            } else {
                const SpSplit8 bo = sp_split8(vv[nu]);
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    const unsigned *wa = wl + (((xi * 4 + nu) * NB + ob) * 2) * 256;
                    const sp_bf16x8 ah = __builtin_bit_cast(sp_bf16x8, *reinterpret_cast<const sp_u32x4 *>(wa));
                    const sp_bf16x8 al = __builtin_bit_cast(sp_bf16x8, *reinterpret_cast<const sp_u32x4 *>(wa + 256));
                    acc[nu][ob] = sp_mfma_x3(ah, al, bo, acc[nu][ob]);
                }
            }
        }
        __syncthreads();                                      // every wave is done with the window
        if (more) commit_in();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next chunk's weights have landed
        __syncthreads();
    }

    // ---- output transform: identical to the fp32 kernel's
    float *ex = lds;
    float *y_b = (ks == 0 ? y : part + (size_t)(ks - 1) * gridDim.y * Kk * HW) + (size_t)b * Kk * HW;
    const float *r_b = (residual && ks == 0) ? residual + (size_t)b * Kk * HW : nullptr;
    const int orow0 = r0 + 2 * TR * tg + 2 * trow, ocol = c0 + 2 * tcol;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
        float *mine = ex + (size_t)wave * 32 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mine[r * 64] = acc[0][mb][r] + acc[1][mb][r] + acc[2][mb][r];
            mine[(16 + r) * 64] = acc[1][mb][r] - acc[2][mb][r] - acc[3][mb][r];
        }
        __syncthreads();
        const float *t0 = ex + (size_t)(tg * 4 + 0) * 32 * 64 + lane, *t1 = t0 + 32 * 64, *t2 = t1 + 32 * 64, *t3 = t2 + 32 * 64;
        if (ocol < W) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 4 * xi + rr;
                const int k = z * WN_KS + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float a0 = t0[r * 64], a1 = t1[r * 64], a2 = t2[r * 64], a3 = t3[r * 64];
                const float b0 = t0[(16 + r) * 64], b1 = t1[(16 + r) * 64], b2 = t2[(16 + r) * 64], b3 = t3[(16 + r) * 64];
                if (k < Kk) {
                    const float bk = (bias && ks == 0) ? bias[k] : 0.f;
                    const size_t o = (size_t)k * HW + (size_t)orow0 * W + ocol;
                    float *dst = y_b + o;
                    f32x2 q0 = {bk, bk}, q1 = {bk, bk};
                    if (r_b) {
                        if (orow0 < H) q0 += *reinterpret_cast<const f32x2 *>(r_b + o);
                        if (orow0 + 1 < H) q1 += *reinterpret_cast<const f32x2 *>(r_b + o + W);
                    }
                    if (orow0 < H) *reinterpret_cast<f32x2 *>(dst) = f32x2{a0 + a1 + a2 + q0.x, b0 + b1 + b2 + q0.y};
                    if (orow0 + 1 < H) *reinterpret_cast<f32x2 *>(dst + W) = f32x2{a1 - a2 - a3 + q1.x, b1 - b2 - b3 + q1.y};
                }
            }
        }
        __syncthreads();
    }
}

// y += part[0] + part[1] + ... (fixed order), n4 = float4 count of one image set.
__global__ __launch_bounds__(256) void wino_sum_partials(float *__restrict__ y, const float *__restrict__ part, size_t n4, int np)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = reinterpret_cast<const f32x4 *>(y)[i];
        for (int k = 0; k < np; ++k) a += reinterpret_cast<const f32x4 *>(part)[(size_t)k * n4 + i];
        reinterpret_cast<f32x4 *>(y)[i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the same convolution, also in the Winograd domain:
//   dL/dg = G^T [ sum_tiles (A E A^T) . (B^T d B) ] G        (E = 2x2 tile of dL/dY, d = 4x4 input tile)
// i.e. sixteen [Cout x tiles] x [tiles x Cin] products (4 multiplies per output pixel and (k,c) pair instead of 9), then a
// 4x4 -> 3x3 transform of the sums.  The contraction runs over TILES, so both MFMA operands want lane = channel:
//   A[i = k][kk = tile parity]  = (A E A^T)[xi][nu] of output channel k,   B[kk][j = c] = (B^T d B)[xi][nu] of channel c,
// both built on the fly from raw tiles in LDS (plane strides 2 * odd: 32 channels -> 32 distinct even banks, the ds_read2
// partner on the odd ones).  One workgroup = 8 waves = 4 transform rows (xi) x 2 input-channel blocks; it owns 64 output x
// 64 input channels (wave: 4 nu x 2 output blocks = 8 accumulators) and walks its share of the (image, tile row, 32-column)
// strips: 64 x 4 x 40 input window + 64 x 2 x 32 dY tile per strip, prefetched through registers into one of two LDS buffers.  Partial sums go to the
// workspace with coalesced stores; wino_wrw_reduce adds the splits in a fixed order and applies G^T . G.
// Workgroups that share strips (the other channel blocks of the same split) are dealt to the same XCD so the re-reads hit L2.
// ---------------------------------------------------------------------------------------------
constexpr int WW_NT = 512;
constexpr int WW_IROW = 40;                        // staged columns c0-4 .. c0+35
constexpr int WW_IPLANE = 4 * WW_IROW + 2;         // 162 = 2 * 81
constexpr int WW_DPLANE = 2 * 32 + 2;              // 66 = 2 * 33
constexpr int WW_IN = 64 * WW_IPLANE;              // 10 368 floats
// NOB = 32-channel output blocks per workgroup: 2, or 1 for layers of at most 32 outputs (DCN's 27-channel offset convs)
template <int NOB>
struct WrwGeom {
    static constexpr int KO = 32 * NOB;
    static constexpr int DY = KO * WW_DPLANE;      // 4 224 / 2 112 floats
    static constexpr int BUF = WW_IN + DY;         // 14 592 floats (58 KB) per buffer with NOB = 2
    static constexpr int PART = 12 * KO * 64;      // floats per partial result: [xi 4][b 3][k][c] (the column half of G^T . G is applied before the store)
};

// BF (DCD_PREC_BF16, the mixed-precision form): the lane's eight steps of a strip ARE the eight k-slots of its operands of
// v_mfma_f32_32x32x16_bf16 (A[i = l & 31][k = 8 (l >> 5) + j], B[k][n = l & 31]: lane (p, h) walks the tiles 8 h + j), so the
// transformed values of the steps are rounded to bf16 pair by pair and every four steps end in 4 NOB matrix instructions
// (v_mfma_f32_32x32x8_bf16_1k: the K = 16 form would hold the packed operands of all eight steps, 48 registers the kernel does not
// have beside its 128 accumulators) where the fp32 form issues 16 NOB of 64 cycles -- the kernel is then bound by its transform
// arithmetic and LDS reads.
template <int NOB, bool BF>
__global__ __launch_bounds__(WW_NT) void wino_wrw3x3_f32(const float *__restrict__ x, const float *__restrict__ gy,
                                                         float *__restrict__ part, int Cin, int Cout, int B, int H, int W,
                                                         int strips_x, int S, int ncg, int nblk)
{
    constexpr int KO = WrwGeom<NOB>::KO, WW_BUF = WrwGeom<NOB>::BUF, WW_PART = WrwGeom<NOB>::PART;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x [WW_IN | WW_DY]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int xi = wave & 3, cb = wave >> 2;
    // (split, channel block) of this workgroup; with S % 8 == 0 the nblk blocks of one split sit on one XCD
    int split, blk;
    {
        const int L = blockIdx.x;
        if ((S & 7) == 0) {
            const int xcd = L & 7, u = L >> 3;
            blk = u % nblk;
            split = (u / nblk) * 8 + xcd;
        } else {
            blk = L % nblk;
            split = L / nblk;
        }
    }
    const int og = blk / ncg, cg = blk - og * ncg;
    const int HW = H * W;
    const int tiles_y = H >> 1;
    const int T = B * tiles_y * strips_x;
    const int t0 = (int)((int64_t)split * T / S), t1 = (int)((int64_t)(split + 1) * T / S);

    const int ra = xi == 0 ? 0 : 1, rb = xi == 3 ? 3 : 2;
    const float sa = xi == 2 ? -1.f : 1.f, sb = (xi == 0 || xi == 3) ? -1.f : 1.f;
    // (A E)[xi][j] = ea E[0][j] + eb E[1][j]
    const float ea = xi == 3 ? 0.f : 1.f, eb = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);

    // ---- staging maps (strip invariant)
    constexpr int KIN = 64 * 4 * 10 / WW_NT;       // 5 dwordx4 per thread
    constexpr int KDY = KO * 2 * 8 / WW_NT;        // 2 (1 with one output block)
    int in_ch[KIN], in_row[KIN], in_q[KIN], dy_o[KDY], dy_row[KDY], dy_q[KDY];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const int e = tid + WW_NT * k;
        in_ch[k] = e / 40;
        const int rem = e - in_ch[k] * 40;
        in_row[k] = rem / 10;
        in_q[k] = rem - in_row[k] * 10;
    }
#pragma unroll
    for (int k = 0; k < KDY; ++k) {
        const int e = tid + WW_NT * k;
        dy_o[k] = e >> 4;
        const int rem = e & 15;
        dy_row[k] = rem >> 3;
        dy_q[k] = rem & 7;
    }
    f32x4 rin[KIN], rdy[KDY];
    // BF: another staging map and buffer loads.  The fp32 form's map (above) costs it nothing -- its addresses and masks live in
    // registers the 64-cycle MFMAs leave time to spill around -- but in the one-product form the same code spilled the per-load
    // 64-bit addresses, and every reload's s_waitcnt vmcnt(0) also waited for the PREVIOUS prefetch load: seven serialised round
    // trips per strip (the kernel ran slower than the fp32 one).  Here thread = (channel tid >> 3, j = tid & 7) takes the items
    // i = j + 8 k (k < 5) of its channel's 4 x 10 dwordx4 (row = i / 10, q = i % 10 by one compare against a constant; LDS offset
    // ch * plane + 4 i), dY thread = (o = tid >> 4 (+ 32 k), row, q); one descriptor per tensor (base + byte count in SGPRs), ONE
    // invariant 32-bit lane offset per load, the strip's part a scalar, out-of-range = the load is dropped and returns zeros.
    // (the thread's indices are re-derived from its id inside each call, behind an opaque copy: hoisted out of the strip loop they
    // are five more live registers in a kernel that has none to spare -- two spilled, and their reloads brought the waits back)
    constexpr int BF_OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t bf_rx, bf_rg;
    if constexpr (BF) {
        bf_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((size_t)B * Cin * HW * 4), 0x00020000);
        bf_rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(gy), 0, (int)((size_t)B * Cout * HW * 4), 0x00020000);
    }
    auto issue_bf = [&](int t) {
        const int b = t / (tiles_y * strips_x);
        const int rem = t - b * (tiles_y * strips_x);
        const int ty = rem / strips_x, sx = rem - ty * strips_x;
        const int r0 = 2 * ty, c0 = 32 * sx;
        const int sx_ = ((b * Cin + cg * 64) * HW + (r0 - 1) * W + c0 - 4) * 4;          // scalar parts (may be negative)
        const int sg_ = ((b * Cout + og * KO) * HW + r0 * W + c0) * 4;
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const int bf_ch = tt >> 3, bf_j = tt & 7;
        const int bf_do = tt >> 4, bf_drow = (tt >> 3) & 1, bf_dq = tt & 7;
        const bool chv = cg * 64 + bf_ch < Cin;
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            constexpr int RB[5] = {0, 0, 1, 2, 3}, QB[5] = {0, 8, 6, 4, 2};               // divmod(8 k, 10)
            const bool wrap = bf_j + QB[k] >= 10;
            const int row = RB[k] + (wrap ? 1 : 0), q = bf_j + QB[k] - (wrap ? 10 : 0);
            const int yy = r0 - 1 + row, xx = c0 - 4 + 4 * q;
            const bool ok = chv && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int voff = ok ? sx_ + (bf_ch * HW + row * W + 4 * q) * 4 : BF_OOB;
            rin[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bf_rx, voff, 0, 0));
        }
#pragma unroll
        for (int k = 0; k < KDY; ++k) {
            const int o = bf_do + 32 * k;
            const bool ok = og * KO + o < Cout && c0 + 4 * bf_dq < W;
            const int voff = ok ? sg_ + (o * HW + bf_drow * W + 4 * bf_dq) * 4 : BF_OOB;
            rdy[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bf_rg, voff, 0, 0));
        }
    };
    auto commit_bf = [&](float *buf) {
        int tt = tid;
        asm volatile("" : "+v"(tt));
        const int bf_ch = tt >> 3, bf_j = tt & 7;
        const int bf_do = tt >> 4, bf_drow = (tt >> 3) & 1, bf_dq = tt & 7;
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            float *d = buf + bf_ch * WW_IPLANE + 4 * bf_j + 32 * k;
            *reinterpret_cast<f32x2 *>(d) = f32x2{rin[k].x, rin[k].y};
            *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rin[k].z, rin[k].w};
        }
#pragma unroll
        for (int k = 0; k < KDY; ++k) {
            float *d = buf + WW_IN + (bf_do + 32 * k) * WW_DPLANE + bf_drow * 32 + 4 * bf_dq;
            *reinterpret_cast<f32x2 *>(d) = f32x2{rdy[k].x, rdy[k].y};
            *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rdy[k].z, rdy[k].w};
        }
    };
    auto issue_f32 = [&](int t) {
        const int b = t / (tiles_y * strips_x);
        const int rem = t - b * (tiles_y * strips_x);
        const int ty = rem / strips_x, sx = rem - ty * strips_x;
        const int r0 = 2 * ty, c0 = 32 * sx;
        const float *x_b = x + ((size_t)b * Cin + (size_t)cg * 64) * HW;
        const float *g_b = gy + ((size_t)b * Cout + (size_t)og * KO) * HW;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            const int yy = r0 - 1 + in_row[k], xx = c0 - 4 + 4 * in_q[k];
            rin[k] = zero4;
            if (cg * 64 + in_ch[k] < Cin && yy >= 0 && yy < H && xx >= 0 && xx < W)
                rin[k] = *reinterpret_cast<const f32x4 *>(x_b + (size_t)in_ch[k] * HW + yy * W + xx);
        }
#pragma unroll
        for (int k = 0; k < KDY; ++k) {
            const int yy = r0 + dy_row[k], xx = c0 + 4 * dy_q[k];
            rdy[k] = zero4;
            if (og * KO + dy_o[k] < Cout && xx < W)
                rdy[k] = *reinterpret_cast<const f32x4 *>(g_b + (size_t)dy_o[k] * HW + yy * W + xx);
        }
    };
    auto commit_f32 = [&](float *buf) {
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            float *d = buf + in_ch[k] * WW_IPLANE + in_row[k] * WW_IROW + 4 * in_q[k];
            *reinterpret_cast<f32x2 *>(d) = f32x2{rin[k].x, rin[k].y};
            *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rin[k].z, rin[k].w};
        }
#pragma unroll
        for (int k = 0; k < KDY; ++k) {
            float *d = buf + WW_IN + dy_o[k] * WW_DPLANE + dy_row[k] * 32 + 4 * dy_q[k];
            *reinterpret_cast<f32x2 *>(d) = f32x2{rdy[k].x, rdy[k].y};
            *reinterpret_cast<f32x2 *>(d + 2) = f32x2{rdy[k].z, rdy[k].w};
        }
    };

    auto issue = [&](int t) { if constexpr (BF) issue_bf(t); else issue_f32(t); };
    auto commit = [&](float *buf) { if constexpr (BF) commit_bf(buf); else commit_f32(buf); };

    f32x16 acc[4][NOB];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][ob][r] = 0.f;

    // lane = (channel p, strip half h): it walks the eight consecutive tiles 8h .. 8h+7, whose 4x4 input patches overlap by
    // two columns -- the row-transformed columns u2,u3 of one tile are u0,u1 of the next, so a step reads two new columns
    // per transform row and the 2x2 dY tiles of its two output blocks.  The two columns start at an odd dword; they are read
    // as the ALIGNED pair above them (ds_read_b64: banks mod 64, the plane stride 162 puts 32 lanes on 32 bank pairs), half
    // of which is carried to the next step -- ds_read2_b32 banks mod 32, where lanes p and p + 16 collide (32 % of this
    // kernel's LDS cycles in round 3's counters).
    const int lane_in = (cb * 32 + p) * WW_IPLANE + 2 + 16 * h;
    const int lane_dy = p * WW_DPLANE + 16 * h;

    // Two LDS buffers, one barrier per strip: while strip t is multiplied out of buffer t&1, the registers holding strip
    // t+1 are written to the other buffer half-way through and re-loaded with strip t+2.
    if (t0 < t1) {
        issue(t0);
        commit(lds);
        if (t0 + 1 < t1) issue(t0 + 1);
    }
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
        const float *bw = lds + ((t - t0) & 1) * WW_BUF;
        float *nb = lds + (((t - t0) & 1) ^ 1) * WW_BUF;
        const float *ia = bw + lane_in + ra * WW_IROW, *ib = bw + lane_in + rb * WW_IROW;
        const float *d0 = bw + WW_IN + lane_dy, *d1 = d0 + (NOB - 1) * 32 * WW_DPLANE;        // one block: d1 = d0, unused
        // software pipeline over the eight steps: the raw LDS values of step s+1 are requested before the eight MFMAs of
        // step s are issued (in-order issue: loads placed after them would wait for the matrix pipe to accept all eight)
#ifdef WW_ABL_NOLDS
        float u0 = sa, u1 = sb;
#else
        // window columns E[1..4] of the first tile from the pairs (E0,E1) (E2,E3) (E4,E5); (E4,E5) is carried
        const f32x2 pa0 = *reinterpret_cast<const f32x2 *>(ia), pa1 = *reinterpret_cast<const f32x2 *>(ia + 2);
        const f32x2 pb0 = *reinterpret_cast<const f32x2 *>(ib), pb1 = *reinterpret_cast<const f32x2 *>(ib + 2);
        f32x2 ca = *reinterpret_cast<const f32x2 *>(ia + 4), cb2 = *reinterpret_cast<const f32x2 *>(ib + 4);
        float u0 = sa * pa0.y + sb * pb0.y, u1 = sa * pa1.x + sb * pb1.x;
        float qa2 = pa1.y, qa3 = ca.x, qb2 = pb1.y, qb3 = cb2.x;
        f32x2 g00 = *reinterpret_cast<const f32x2 *>(d0), g01 = *reinterpret_cast<const f32x2 *>(d0 + 32);
        f32x2 g10 = *reinterpret_cast<const f32x2 *>(d1), g11 = *reinterpret_cast<const f32x2 *>(d1 + 32);
#endif
        float ev[12];                                  // BF: the even step's values, waiting for their odd partners
        sp_u32x2 pk[12];                               // BF: [v nu 0..3 | wa nu 0..3 | wb nu 0..3] x bf16 pairs of steps (2 i, 2 i + 1)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#ifndef WW_ABL_NOSTAGE
            if (s == 4 && t + 1 < t1) {
                commit(nb);
                if (t + 2 < t1) issue(t + 2);
            }
#endif
#ifdef WW_ABL_NOLDS
            const float u2 = sa + (float)t, u3 = sb * (float)s;
            const float a0 = ea * (float)t, a1 = eb, b0 = ea + (float)s, b1 = eb * (float)s;
#else
            const float u2 = sa * qa2 + sb * qb2, u3 = sa * qa3 + sb * qb3;
            // 2x2 dY tile of each output block: two 8-byte-aligned pairs, 128 B apart (one ds_read2_b64)
            const float a0 = ea * g00.x + eb * g01.x, a1 = ea * g00.y + eb * g01.y;
            const float b0 = ea * g10.x + eb * g11.x, b1 = ea * g10.y + eb * g11.y;
            if (s + 1 < 8) {
                qa2 = ca.y; qb2 = cb2.y;
                ca = *reinterpret_cast<const f32x2 *>(ia + 2 * s + 6); cb2 = *reinterpret_cast<const f32x2 *>(ib + 2 * s + 6);
                qa3 = ca.x; qb3 = cb2.x;
                g00 = *reinterpret_cast<const f32x2 *>(d0 + 2 * s + 2); g01 = *reinterpret_cast<const f32x2 *>(d0 + 2 * s + 34);
                g10 = *reinterpret_cast<const f32x2 *>(d1 + 2 * s + 2); g11 = *reinterpret_cast<const f32x2 *>(d1 + 2 * s + 34);
            }
#endif
            const float v[4] = {u0 - u2, u1 + u2, u2 - u1, u1 - u3};
            const float wa[4] = {a0, a0 + a1, a0 - a1, -a1};
            const float wb[4] = {b0, b0 + b1, b0 - b1, -b1};
            if constexpr (BF) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    if (s & 1) {
                        pk[nu][(s >> 1) & 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{ev[nu], v[nu]}, sp_bf16x2));
                        pk[4 + nu][(s >> 1) & 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{ev[4 + nu], wa[nu]}, sp_bf16x2));
                        if (NOB == 2) pk[8 + nu][(s >> 1) & 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(sp_f32x2{ev[8 + nu], wb[nu]}, sp_bf16x2));
                    } else {
                        ev[nu] = v[nu]; ev[4 + nu] = wa[nu]; ev[8 + nu] = wb[nu];
                    }
                }
                // four steps = the four k-slots of a lane in v_mfma_f32_32x32x8_bf16_1k (A[i = l & 31][k = 4 (l >> 5) + j]): which tile sits in
                // which slot does not matter, the contraction runs over all of them and both operands of a lane come from the same steps
                if ((s & 3) == 3) {
#pragma unroll
                    for (int nu = 0; nu < 4; ++nu) {
                        const sp_s16x4 bv = __builtin_bit_cast(sp_s16x4, pk[nu]);
                        acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(sp_s16x4, pk[4 + nu]), bv, acc[nu][0], 0, 0, 0);
                        if (NOB == 2)
                            acc[nu][NOB - 1] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(sp_s16x4, pk[8 + nu]), bv, acc[nu][NOB - 1], 0, 0, 0);
                    }
                }
            } else {
                __builtin_amdgcn_sched_barrier(0);               // keep the loads above ahead of the MFMAs below
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    acc[nu][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[nu], v[nu], acc[nu][0], 0, 0, 0);
                    if (NOB == 2) acc[nu][NOB - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[nu], v[nu], acc[nu][NOB - 1], 0, 0, 0);
                }
            }
            u0 = u2;
            u1 = u3;
        }

#ifndef WW_ABL_NOBAR
        __syncthreads();
#endif
    }

    // acc[nu][ob][r] = M[xi][nu][k = 32 ob + (r&3) + 8 (r>>2) + 4h][c = 32 cb + p]  ->  part[blk][split][xi*3+b][k][c]
#ifdef WW_ABL_NOEPI
    if (acc[0][0][0] != 123.456f) return;
#endif
    // dL/dg = G^T M G: the sum over nu (columns of M: (1, .5, .5, 0), (0, .5, -.5, 0), (0, .5, .5, 1)) is linear and local to the wave,
    // so it is applied to the partial before the store -- 12 instead of 16 planes per partial to write here and to read back in
    // wino_wrw_reduce, which applies the row half over xi (the partials are 64 MB per 64 -> 64 @ 96x320 call)
    float *mine = part + ((size_t)blk * S + split) * WW_PART;
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
            const float m0 = acc[0][ob][r], m1 = acc[1][ob][r], m2 = acc[2][ob][r], m3 = acc[3][ob][r];
            float *dst = mine + ((xi * 3) * KO + k) * 64 + cb * 32 + p;
            dst[0] = m0 + 0.5f * (m1 + m2);
            dst[KO * 64] = 0.5f * (m1 - m2);
            dst[2 * KO * 64] = 0.5f * (m1 + m2) + m3;
        }
}

// grid = (nblk * KO * 64 / 16), block = 256 = 16 positions x 16 (k,c) pairs: sums the S partials of its 16 pairs (fixed order:
// reproducible), exchanges the 16 positions through LDS and writes dw = G^T M G.  KO = output channels per block (64 / 32).
__global__ __launch_bounds__(256) void wino_wrw_reduce(const float *__restrict__ part, float *__restrict__ gw, int Cin, int Cout, int S,
                                                       int ncg, int KO)
{
    __shared__ float m[12][17];
    const int tid = threadIdx.x;
    const int pos = tid >> 4, l = tid & 15;            // pos = xi * 3 + b (12 of the 16 thread rows have work)
    const int gidx = blockIdx.x * 16 + l;           // (blk, k, c)
    const int per = KO * 64;
    const int blk = gidx / per, kc = gidx - blk * per;
    if (pos < 12) {
        const float *src = part + ((size_t)blk * S * 12 + pos) * per + kc;
        float sum = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < S; ++sp) sum += src[(size_t)sp * 12 * per];
        m[pos][l] = sum;
    }
    __syncthreads();
    if (pos < 9) {
        const int a = pos / 3, b = pos - a * 3;
        // G^T rows: a = 0: (1, .5, .5, 0)   1: (0, .5, -.5, 0)   2: (0, .5, .5, 1)   (the column half was applied by the producer)
        const float ga[4] = {a == 0 ? 1.f : 0.f, 0.5f, a == 1 ? -0.5f : 0.5f, a == 2 ? 1.f : 0.f};
        float r = 0.f;
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) r += ga[xi] * m[xi * 3 + b][l];
        const int og = blk / ncg, cg = blk - og * ncg;
        const int k = og * KO + (kc >> 6), c = cg * 64 + (kc & 63);
        if (k < Cout && c < Cin) gw[((size_t)k * Cin + c) * 9 + pos] = r;
    }
}

}  // namespace

// Region geometry and contraction split of one call.  The workgroup holds 103 KB of LDS and 8 waves with 128 accumulator
// registers each, so exactly one sits on a CU: time ~ rounds of `cus` workgroups x chunks per workgroup.
struct ConvPlan {
    int geom;          // 0: 8 x 32 px regions, 1: 12 x 20 px
    int nb;            // 32-channel output blocks per workgroup: 2, or 1 for layers of at most 32 outputs
    int tiles_x, tiles_y, nchunk, nz, ksplit;
};

static int device_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    return cus;
}

static ConvPlan conv_plan(int B, int Cc, int H, int W, int Kk)
{
    ConvPlan best{};
    const int cus = device_cus();
    int64_t best_cost = -1;
    for (int g = 0; g < 2; ++g) {
        ConvPlan p;
        p.geom = g;
        p.tiles_x = g == 0 ? (W + 31) / 32 : (W + 19) / 20;
        p.tiles_y = g == 0 ? (H + 7) / 8 : (H + 11) / 12;
        p.nchunk = (Cc + WN_CH - 1) / WN_CH;
        p.nb = Kk <= 32 ? 1 : 2;
        p.nz = (Kk + 32 * p.nb - 1) / (32 * p.nb);
        const int64_t wgs = (int64_t)p.tiles_x * p.tiles_y * B * p.nz;
        // split the contraction while the launch leaves CUs idle and a split keeps at least 4 chunks (its output transform
        // and the partial image it writes are not free; 4 vs 8: one image per GPU, 256 -> 256 @ 24x80 39 -> 32 us)
        static const int min_chunks = dcd_env("DCD_CONV_MINCHUNK") ? atoi(dcd_env("DCD_CONV_MINCHUNK")) : 4;
        int ks = 1;
        while (ks < 8 && wgs * (ks * 2) <= cus && p.nchunk / (ks * 2) >= min_chunks) ks *= 2;
        p.ksplit = ks;
        const int64_t rounds = (wgs * ks + cus - 1) / cus;
        const int64_t cost = rounds * ((p.nchunk + ks - 1) / ks + 2) * (g == 0 ? 16 : 17);     // 12x20: 30 of 32 lanes, more halo
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = p;
        }
    }
    return best;
}

template <int TR, int TC, int NB>
static int conv_launch(hipStream_t stream, const ConvPlan &pl, const float *input, const float *ul, float *output, float *part,
                       const float *bias, const float *residual, int B, int Cc, int H, int W, int Kk)
{
    static LdsLimit lds_limit;
    size_t ldsb = (size_t)2 * WinoGeom<TR, TC, NB>::BUF * sizeof(float);
    if (ldsb < (size_t)8 * 32 * 64 * sizeof(float)) ldsb = (size_t)8 * 32 * 64 * sizeof(float);      // the epilogue's exchange
    if (!lds_limit.raise((int)ldsb, wino_conv3x3_f32<TR, TC, NB>)) return DCD_ERR_LAUNCH;
    hipLaunchKernelGGL((wino_conv3x3_f32<TR, TC, NB>), dim3(pl.tiles_x * pl.tiles_y, B, pl.nz * pl.ksplit), dim3(WN_NT), ldsb, stream,
                       input, ul, output, part, bias, residual, Cc, H, W, Kk, pl.tiles_x, pl.nchunk, pl.nz);
    return DCD_OK;
}

template <int NB, int NP>
static int conv_launch_split(hipStream_t stream, const ConvPlan &pl, const float *input, const unsigned *us, float *output, float *part,
                             const float *bias, const float *residual, int B, int Cc, int H, int W, int Kk)
{
    static LdsLimit lds_limit;
    using G = WinoGeom<2, 16, NB>;
    const size_t ldsb = ((size_t)((WS_CH * G::PLANE + 3) & ~3) + (size_t)2 * 16 * NB * 2 * 256) * sizeof(float);
    if (!lds_limit.raise((int)ldsb, wino_conv3x3_split<NB, NP>)) return DCD_ERR_LAUNCH;
    hipLaunchKernelGGL((wino_conv3x3_split<NB, NP>), dim3(pl.tiles_x * pl.tiles_y, B, pl.nz * pl.ksplit), dim3(WN_NT), ldsb, stream, input,
                       us, output, part, bias, residual, Cc, H, W, Kk, pl.tiles_x, pl.nchunk, pl.nz);
    return DCD_OK;
}

// split-bf16 form: 8 x 32 px regions, chunks of 16 channels; the contraction is split while CUs would idle and a split keeps two chunks
static ConvPlan conv_plan_split(int B, int Cc, int H, int W, int Kk)
{
    ConvPlan p;
    p.geom = 0;
    p.tiles_x = (W + 31) / 32;
    p.tiles_y = (H + 7) / 8;
    p.nchunk = (Cc + WS_CH - 1) / WS_CH;
    p.nb = Kk <= 32 ? 1 : 2;
    p.nz = (Kk + 32 * p.nb - 1) / (32 * p.nb);
    const int64_t wgs = (int64_t)p.tiles_x * p.tiles_y * B * p.nz;
    const int cus = device_cus();
    int ks = 1;
    while (ks < 8 && wgs * (ks * 2) <= cus && p.nchunk / (ks * 2) >= 2) ks *= 2;
    p.ksplit = ks;
    return p;
}

// direct one-product form (conv_direct_bf16.inc): rows per wave P (plan.geom) = 4 or 2, i.e. 16 x 32 or 8 x 32 px regions
static bool conv_direct_ok(int Cc, int H, int W)
{
    return (int64_t)(Cc + DC_CH) * H * W < (1ll << 29);
}

static ConvPlan conv_plan_direct(int B, int Cc, int H, int W, int Kk)
{
    static const int pin = dcd_env("DCD_CONV_DIRECT_P") ? atoi(dcd_env("DCD_CONV_DIRECT_P")) : 0;
    const int cus = device_cus();
    ConvPlan best{};
    int64_t best_cost = -1;
    for (int P = 2; P <= 4; P += 2) {
        if (P != (pin == 4 ? 4 : 2)) continue;                       // 16 x 32 px regions (one workgroup per CU) only when pinned
        ConvPlan p;
        // 8 x 64 px regions (eight waves) where the width divides: a third less L2 traffic for the column halo
        static const int wxpin = dcd_env("DCD_CONV_DIRECT_WX") ? atoi(dcd_env("DCD_CONV_DIRECT_WX")) : 0;
        const int wx = P == 2 && wxpin == 2 ? 2 : 1;
        p.geom = P | (wx << 4);
        p.tiles_x = (W + 32 * wx - 1) / (32 * wx);
        p.tiles_y = (H + 4 * P - 1) / (4 * P);
        p.nchunk = (Cc + DC_CH - 1) / DC_CH;
        p.nb = Kk <= 32 ? 1 : 2;
        p.nz = (Kk + 32 * p.nb - 1) / (32 * p.nb);
        const int64_t wgs = (int64_t)p.tiles_x * p.tiles_y * B * p.nz;
        // Three workgroups share a CU.  A long contraction is split while slots would stay empty and a split keeps 16 chunks
        // (256 -> 256 @ 24x80 x 8: 46.5 us split in two or not, and the split pays a wino_sum_partials launch); a launch of less
        // than 1.5 workgroups per CU (one or two images per GPU) is split down to two chunks -- there a workgroup's chunks are a
        // serial chain of load latencies (256 -> 256 @ 24x80 x 1: 27.7 us unsplit against the Winograd form's 14.5).
        static const int min_chunks = dcd_env("DCD_CONV_DIRECT_MINCHUNK") ? atoi(dcd_env("DCD_CONV_DIRECT_MINCHUNK")) : 16;
        int ks = 1;
        while (ks < 8) {
            const int64_t next = wgs * ks * 2;
            const int left = p.nchunk / (ks * 2);
            const bool fill = next <= 3 * cus && left >= min_chunks;
            const bool small = 2 * next <= 3 * cus && left >= 2;
            if (!fill && !small) break;
            ks *= 2;
        }
        p.ksplit = ks;
        best = p;
    }
    (void)best_cost;
    return best;
}

template <int NB, int P, int MODE, int WX>
static int conv_launch_direct(hipStream_t stream, const ConvPlan &pl, const float *input, const unsigned *wd, float *output, float *part,
                              const float *bias, const float *residual, int B, int Cc, int H, int W, int Kk)
{
    static LdsLimit lds_limit;
    const size_t ldsb = ((size_t)2 * (4 * P + 2) * (32 * WX + 8) + (size_t)2 * 9 * NB * 64) * 16;
    if constexpr (MODE == 2) {
        if (!lds_limit.raise((int)ldsb, conv3x3_direct_bf16_w4<NB, P, WX>)) return DCD_ERR_LAUNCH;
        hipLaunchKernelGGL((conv3x3_direct_bf16_w4<NB, P, WX>), dim3(pl.tiles_x * pl.tiles_y, B, pl.nz * pl.ksplit), dim3(DC_NT * WX), ldsb,
                           stream, input, wd, output, part, bias, residual, Cc, H, W, Kk, pl.tiles_x, pl.nchunk, pl.nz);
    } else {
        if (!lds_limit.raise((int)ldsb, conv3x3_direct_bf16<NB, P, MODE, WX>)) return DCD_ERR_LAUNCH;
        hipLaunchKernelGGL((conv3x3_direct_bf16<NB, P, MODE, WX>), dim3(pl.tiles_x * pl.tiles_y, B, pl.nz * pl.ksplit), dim3(DC_NT * WX), ldsb,
                           stream, input, wd, output, part, bias, residual, Cc, H, W, Kk, pl.tiles_x, pl.nchunk, pl.nz);
    }
    return DCD_OK;
}

static size_t tw_dwords_split(int Cin, int Cout, int backward_data)
{
    const int Cc = backward_data ? Cout : Cin, Kk = backward_data ? Cin : Cout;
    const int nb = Kk <= 32 ? 1 : 2;
    return (size_t)((Cc + WS_CH - 1) / WS_CH) * ((Kk + 32 * nb - 1) / (32 * nb)) * 16 * nb * 2 * 256;
}

extern "C" {

size_t dcd_conv3x3_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    const int cmax = Cin > Cout ? Cin : Cout;
    const size_t nchunk = (size_t)(cmax + WN_CH - 1) / WN_CH, nz = (size_t)(cmax + 63) / 64;
    // transformed weights + the partial images of a split contraction (either direction of the call)
    const ConvPlan f = conv_plan(B, Cin, H, W, Cout), d = conv_plan(B, Cout, H, W, Cin);
    const size_t pf = (size_t)(f.ksplit - 1) * B * Cout * H * W, pd = (size_t)(d.ksplit - 1) * B * Cin * H * W;
    return (nchunk * nz * 16 * 64 * WN_CH + (pf > pd ? pf : pd)) * sizeof(float);
}

// weights: raw (Cout, Cin, 3, 3) when !prepared (transformed into the head of the workspace first), else the transformed weights
// of this direction from dcd_conv3x3_transform_weights.
static int conv3x3_run(hipStream_t stream, const float *input, const float *weights, int prepared, const float *bias,
                       const float *residual, float *output, int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace,
                       size_t workspace_bytes)
{
    (void)hipGetLastError();
    if (!input || !weights || !output || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DCD_ERR_BAD_ARG;
    if ((W & 3) || (H & 1) || (int64_t)(Cin > Cout ? Cin : Cout) * H * W >= (1ll << 31) || (bias && backward_data)) return DCD_ERR_BAD_ARG;
    // contraction channels Cc and produced channels Kk of this call
    const int Cc = backward_data ? Cout : Cin, Kk = backward_data ? Cin : Cout;
    ConvPlan pl = conv_plan(B, Cc, H, W, Kk);
    if (const char *e = dcd_env("DCD_CONV_GEOM")) {                     // A/B timing: "0" / "1" pins the region shape
        const int g = atoi(e);
        if (g == 0 || g == 1) {
            ConvPlan q = pl;
            q.geom = g;
            q.tiles_x = g == 0 ? (W + 31) / 32 : (W + 19) / 20;
            q.tiles_y = g == 0 ? (H + 7) / 8 : (H + 11) / 12;
            q.ksplit = 1;
            pl = q;
        }
    }
    const int ksz = 32 * pl.nb;
    const size_t ul_floats = prepared ? 0 : (size_t)pl.nchunk * pl.nz * 16 * ksz * WN_CH;
    const size_t img = (size_t)B * Kk * H * W;
    const size_t need = (ul_floats + (size_t)(pl.ksplit - 1) * img) * sizeof(float);
    if (need && (!workspace || workspace_bytes < need)) return DCD_ERR_WORKSPACE;
    float *part = (float *)workspace + ul_floats;
    const float *ul = weights;
    if (!prepared) {
        const int nprep = pl.nz * pl.nchunk * ksz * WN_CH;
        // forward: w is (Cout, Cin, 3, 3) = (Kk, Cc); backward-data: w is (Cout, Cin) = (Cc, Kk), read transposed + flipped
        hipLaunchKernelGGL(wino_prep_weights, dim3((nprep + 255) / 256 < 4096 ? (nprep + 255) / 256 : 4096), dim3(256), 0, stream, weights,
                           (float *)workspace, Cc, Kk, backward_data ? 1 : 0, pl.nchunk, pl.nz, ksz);
        ul = (const float *)workspace;
    }
    const int st = pl.geom == 0 ? (pl.nb == 2 ? conv_launch<2, 16, 2>(stream, pl, input, ul, output, part, bias, residual, B, Cc, H, W, Kk)
                                              : conv_launch<2, 16, 1>(stream, pl, input, ul, output, part, bias, residual, B, Cc, H, W, Kk))
                                : (pl.nb == 2 ? conv_launch<3, 10, 2>(stream, pl, input, ul, output, part, bias, residual, B, Cc, H, W, Kk)
                                              : conv_launch<3, 10, 1>(stream, pl, input, ul, output, part, bias, residual, B, Cc, H, W, Kk));
    if (st != DCD_OK) return st;
    if (pl.ksplit > 1) {
        const size_t n4 = img / 4;                                      // W % 4 == 0
        const int nb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(wino_sum_partials, dim3(nb), dim3(256), 0, stream, output, (const float *)part, n4, pl.ksplit - 1);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3(void *stream_, const float *input, const float *weight, const float *bias, const float *residual, float *output,
                int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace, size_t workspace_bytes)
{
    if (!workspace) return DCD_ERR_BAD_ARG;
    return conv3x3_run((hipStream_t)stream_, input, weight, 0, bias, residual, output, B, Cin, H, W, Cout, backward_data, workspace,
                       workspace_bytes);
}

// floats of the transformed weights of one direction (layout: the kernels' weight slab per (output slice, chunk))
static size_t tw_floats(int Cin, int Cout, int backward_data)
{
    const int Cc = backward_data ? Cout : Cin, Kk = backward_data ? Cin : Cout;
    const int nb = Kk <= 32 ? 1 : 2, ks = 32 * nb;
    return (size_t)((Cc + WN_CH - 1) / WN_CH) * ((Kk + ks - 1) / ks) * 16 * ks * WN_CH;
}

size_t dcd_conv3x3_weights_bytes(int Cin, int Cout, int backward_data)
{
    return Cin > 0 && Cout > 0 ? tw_floats(Cin, Cout, backward_data) * sizeof(float) : 0;
}

int dcd_conv3x3_transform_weights(void *stream_, const float *weight, int Cin, int Cout, float *forward_out, float *backward_out)
{
    (void)hipGetLastError();
    if (!weight || Cin <= 0 || Cout <= 0 || (!forward_out && !backward_out)) return DCD_ERR_BAD_ARG;
    PrepBoth p;
    int nmax = 0;
    for (int mode = 0; mode < 2; ++mode) {
        const int Cc = mode ? Cout : Cin, Kk = mode ? Cin : Cout;
        p.ul[mode] = mode ? backward_out : forward_out;
        p.Cc[mode] = Cc;
        p.Kk[mode] = Kk;
        p.ks[mode] = Kk <= 32 ? 32 : 64;
        p.nchunk[mode] = (Cc + WN_CH - 1) / WN_CH;
        p.nz[mode] = (Kk + p.ks[mode] - 1) / p.ks[mode];
        const int n = p.nz[mode] * p.nchunk[mode] * p.ks[mode] * WN_CH;
        if (p.ul[mode] && n > nmax) nmax = n;
    }
    const int nb = (nmax + 255) / 256 < 4096 ? (nmax + 255) / 256 : 4096;
    hipLaunchKernelGGL(wino_prep_weights_both, dim3(nb, 2), dim3(256), 0, (hipStream_t)stream_, weight, p);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3_transform_weights_table(void *stream_, const long long *table, int entries)
{
    (void)hipGetLastError();
    if (!table || entries <= 0 || entries > 65535) return DCD_ERR_BAD_ARG;
    // 64 blocks x 256 threads per (layer, direction): the largest DGDE layer (512 -> 512) has 262 144 items = 16 per thread
    hipLaunchKernelGGL(wino_prep_weights_table, dim3(64, 2, entries), dim3(256), 0, (hipStream_t)stream_, table);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3_prepared(void *stream_, const float *input, const float *transformed, const float *bias, const float *residual,
                         float *output, int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace,
                         size_t workspace_bytes)
{
    return conv3x3_run((hipStream_t)stream_, input, transformed, 1, bias, residual, output, B, Cin, H, W, Cout, backward_data, workspace,
                       workspace_bytes);
}

size_t dcd_conv3x3_split_weights_bytes(int Cin, int Cout, int backward_data)
{
    return Cin > 0 && Cout > 0 ? tw_dwords_split(Cin, Cout, backward_data) * sizeof(unsigned) : 0;
}

int dcd_conv3x3_split_transform_weights(void *stream_, const float *weight, int Cin, int Cout, void *forward_out, void *backward_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!weight || Cin <= 0 || Cout <= 0 || (!forward_out && !backward_out)) return DCD_ERR_BAD_ARG;
    for (int mode = 0; mode < 2; ++mode) {
        unsigned *dst = (unsigned *)(mode ? backward_out : forward_out);
        if (!dst) continue;
        const int Cc = mode ? Cout : Cin, Kk = mode ? Cin : Cout;
        const int nb = Kk <= 32 ? 1 : 2, nchunk = (Cc + WS_CH - 1) / WS_CH, nz = (Kk + 32 * nb - 1) / (32 * nb);
        const int n = nz * nchunk * 32 * nb * 8;
        const int grid = (n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096;
        if (nb == 2) hipLaunchKernelGGL(wino_prep_weights_split<2>, dim3(grid), dim3(256), 0, stream, weight, dst, Cc, Kk, mode, nchunk, nz);
        else hipLaunchKernelGGL(wino_prep_weights_split<1>, dim3(grid), dim3(256), 0, stream, weight, dst, Cc, Kk, mode, nchunk, nz);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3_split_transform_weights_table(void *stream_, const long long *table, int entries)
{
    (void)hipGetLastError();
    if (!table || entries <= 0 || entries > 65535) return DCD_ERR_BAD_ARG;
    // 64 blocks x 256 threads per (layer, direction): the largest DGDE layer (512 -> 512) has 65 536 items of 32 stores each
    hipLaunchKernelGGL(wino_prep_weights_split_table, dim3(64, 2, entries), dim3(256), 0, (hipStream_t)stream_, table);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

size_t dcd_conv3x3_split_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    const ConvPlan f = conv_plan_split(B, Cin, H, W, Cout), d = conv_plan_split(B, Cout, H, W, Cin);
    const size_t pf = (size_t)(f.ksplit - 1) * B * Cout * H * W, pd = (size_t)(d.ksplit - 1) * B * Cin * H * W;
    return (pf > pd ? pf : pd) * sizeof(float) + 16;
}

int dcd_conv3x3_split_prepared(void *stream_, const float *input, const void *transformed, const float *bias, const float *residual,
                               float *output, int B, int Cin, int H, int W, int Cout, int backward_data, int precision, void *workspace,
                               size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !transformed || !output || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DCD_ERR_BAD_ARG;
    if (precision != DCD_PREC_BF16X3 && precision != DCD_PREC_BF16) return DCD_ERR_BAD_ARG;
    const bool one = precision == DCD_PREC_BF16;
    if ((W & 3) || (H & 1) || (int64_t)(Cin > Cout ? Cin : Cout) * H * W >= (1ll << 31) || (bias && backward_data)) return DCD_ERR_BAD_ARG;
    const int Cc = backward_data ? Cout : Cin, Kk = backward_data ? Cin : Cout;
    const ConvPlan pl = conv_plan_split(B, Cc, H, W, Kk);
    const size_t img = (size_t)B * Kk * H * W;
    const size_t need = (size_t)(pl.ksplit - 1) * img * sizeof(float);
    if (need && (!workspace || workspace_bytes < need)) return DCD_ERR_WORKSPACE;
    float *part = (float *)workspace;
    const unsigned *us = (const unsigned *)transformed;
    const int st = pl.nb == 2 ? (one ? conv_launch_split<2, 1>(stream, pl, input, us, output, part, bias, residual, B, Cc, H, W, Kk)
                                     : conv_launch_split<2, 3>(stream, pl, input, us, output, part, bias, residual, B, Cc, H, W, Kk))
                              : (one ? conv_launch_split<1, 1>(stream, pl, input, us, output, part, bias, residual, B, Cc, H, W, Kk)
                                     : conv_launch_split<1, 3>(stream, pl, input, us, output, part, bias, residual, B, Cc, H, W, Kk));
    if (st != DCD_OK) return st;
    if (pl.ksplit > 1) {
        const size_t n4 = img / 4;
        const int nb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(wino_sum_partials, dim3(nb), dim3(256), 0, stream, output, (const float *)part, n4, pl.ksplit - 1);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

// ---- the one-product form as a direct implicit GEMM (conv_direct_bf16.inc): its own weight layout, same call shape
size_t dcd_conv3x3_bf16_weights_bytes(int Cin, int Cout, int backward_data)
{
    if (Cin <= 0 || Cout <= 0) return 0;
    return 16 * (backward_data ? dc_weight_slots(Cout, Cin) : dc_weight_slots(Cin, Cout));
}

int dcd_conv3x3_bf16_transform_weights(void *stream_, const float *weight, int Cin, int Cout, void *forward_out, void *backward_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!weight || Cin <= 0 || Cout <= 0 || (!forward_out && !backward_out)) return DCD_ERR_BAD_ARG;
    DirectPrep p;
    size_t nmax = 0;
    for (int mode = 0; mode < 2; ++mode) {
        p.ud[mode] = (unsigned *)(mode ? backward_out : forward_out);
        p.Cc[mode] = mode ? Cout : Cin;
        p.Kk[mode] = mode ? Cin : Cout;
        const size_t n = p.ud[mode] ? dc_weight_slots(p.Cc[mode], p.Kk[mode]) : 0;
        if (n > nmax) nmax = n;
    }
    nmax /= 9;                                                             // one thread per nine slots
    const int nb = (int)((nmax + 255) / 256 < 2048 ? (nmax + 255) / 256 : 2048);
    hipLaunchKernelGGL(direct_prep_weights_both, dim3(nb, 2), dim3(256), 0, stream, weight, p);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3_bf16_transform_weights_table(void *stream_, const long long *table, int entries)
{
    (void)hipGetLastError();
    if (!table || entries <= 0 || entries > 65535) return DCD_ERR_BAD_ARG;
    // 32 blocks x 256 threads per (layer, direction): the largest DGDE layer (512 -> 512) has 16 384 items of nine slots = 2 per thread
    hipLaunchKernelGGL(direct_prep_weights_table, dim3(32, 2, entries), dim3(256), 0, (hipStream_t)stream_, table);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

size_t dcd_conv3x3_bf16_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    const size_t pf = (size_t)(conv_plan_direct(B, Cin, H, W, Cout).ksplit - 1) * B * Cout * H * W;
    const size_t pd = (size_t)(conv_plan_direct(B, Cout, H, W, Cin).ksplit - 1) * B * Cin * H * W;
    return (pf > pd ? pf : pd) * sizeof(float) + 16;
}

int dcd_conv3x3_bf16_prepared(void *stream_, const float *input, const void *transformed, const float *bias, const float *residual,
                              float *output, int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace,
                              size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !transformed || !output || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DCD_ERR_BAD_ARG;
    if (bias && backward_data) return DCD_ERR_BAD_ARG;
    const int Cc = backward_data ? Cout : Cin, Kk = backward_data ? Cin : Cout;
    if (!conv_direct_ok(Cc, H, W) || (int64_t)Kk * H * W >= (1ll << 31) || (W & 3)) return DCD_ERR_BAD_ARG;
    const ConvPlan pl = conv_plan_direct(B, Cc, H, W, Kk);
    const size_t img = (size_t)B * Kk * H * W;
    const size_t need = (size_t)(pl.ksplit - 1) * img * sizeof(float);
    if (need && (!workspace || workspace_bytes < need)) return DCD_ERR_WORKSPACE;
    if (pl.ksplit > 1 && (img & 3)) return DCD_ERR_BAD_ARG;               // wino_sum_partials adds float4s
    float *part = (float *)workspace;
    const unsigned *wd = (const unsigned *)transformed;
    // DCD_CONV_DIRECT_MODE: 0 weights resident, 1 window resident, 2 both streamed (default: 2 with 8 x 64 regions, else 1)
    static const int mpin = dcd_env("DCD_CONV_DIRECT_MODE") ? atoi(dcd_env("DCD_CONV_DIRECT_MODE")) : -1;
    const int P = pl.geom & 15, wx = pl.geom >> 4;
    const int mode = P == 4 ? 0 : (mpin >= 0 ? mpin : 1);
#define DCD_DIRECT(NB_, P_, M_, WX_) conv_launch_direct<NB_, P_, M_, WX_>(stream, pl, input, wd, output, part, bias, residual, B, Cc, H, W, Kk)
#define DCD_DIRECT_NB(NB_)                                                                                                              \
    (P == 4 ? DCD_DIRECT(NB_, 4, 0, 1)                                                                                                  \
            : wx == 2 ? (mode == 2 ? DCD_DIRECT(NB_, 2, 2, 2) : DCD_DIRECT(NB_, 2, 1, 2))                                               \
                      : (mode == 2 ? DCD_DIRECT(NB_, 2, 2, 1) : mode == 1 ? DCD_DIRECT(NB_, 2, 1, 1) : DCD_DIRECT(NB_, 2, 0, 1)))
    const int st = pl.nb == 2 ? DCD_DIRECT_NB(2) : DCD_DIRECT_NB(1);
#undef DCD_DIRECT_NB
#undef DCD_DIRECT
    if (st != DCD_OK) return st;
    if (pl.ksplit > 1) {
        const size_t n4 = img / 4;
        const int nb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(wino_sum_partials, dim3(nb), dim3(256), 0, stream, output, (const float *)part, n4, pl.ksplit - 1);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

// ---- 1x1 convolutions in the one-product form (conv1x1_bf16.inc)
int dcd_conv1x1_bf16(void *stream_, const float *weight, int ldw, int transposed, int n_inputs, const float *const *inputs,
                     const int *channels, float *output, int B, int M, long long HW)
{
    (void)hipGetLastError();
    if (!weight || !inputs || !channels || !output || n_inputs < 1 || n_inputs > 4 || B <= 0 || M <= 0 || HW <= 0 || (HW & 3) || ldw <= 0)
        return DCD_ERR_BAD_ARG;
    PwArgs a;
    a.n_in = n_inputs;
    a.K = 0;
    for (int i = 0; i < 4; ++i) {
        a.in[i] = i < n_inputs ? inputs[i] : nullptr;
        a.ch[i] = i < n_inputs ? channels[i] : 0;
        if (i < n_inputs && (!inputs[i] || channels[i] <= 0 || (channels[i] & 15) || (int64_t)channels[i] * HW >= (1ll << 29)))
            return DCD_ERR_BAD_ARG;
        a.K += a.ch[i];
    }
    if ((int64_t)M * HW >= (1ll << 31)) return DCD_ERR_BAD_ARG;
    a.w = weight; a.ldw = ldw; a.wt = transposed ? 1 : 0; a.out = output; a.M = M; a.HW = HW;
    const int nb = M <= 32 ? 1 : 2;
    dim3 grid((unsigned)((HW + 511) / 512), B, (M + 32 * nb - 1) / (32 * nb));
    if (nb == 2) hipLaunchKernelGGL(pw_conv_bf16<2>, grid, dim3(PW_NT), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(pw_conv_bf16<1>, grid, dim3(PW_NT), 0, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static void pw_wrw_partition(int B, int O, int C, long long HW, int &nog, int &ncg, int &S)
{
    nog = (O + 63) / 64;
    ncg = (C + 127) / 128;
    const int64_t T = (int64_t)B * ((HW + 63) / 64);
    int64_t s = 2 * (int64_t)device_cus() / (nog * ncg);           // two workgroups per CU
    if (s > T / 4) s = T / 4;
    if (s < 1) s = 1;
    S = (int)s;
}

size_t dcd_conv1x1_wrw_bf16_workspace_bytes(int B, int O, int C, long long HW)
{
    if (B <= 0 || O <= 0 || C <= 0 || HW <= 0) return 0;
    int nog, ncg, S;
    pw_wrw_partition(B, O, C, HW, nog, ncg, S);
    return (size_t)nog * ncg * S * 64 * 128 * sizeof(float);
}

int dcd_conv1x1_wrw_bf16(void *stream_, const float *grad_output, const float *input, float *grad_weight, int ldw, int B, int O, int C,
                         long long HW, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_output || !input || !grad_weight || !workspace || B <= 0 || O <= 0 || C <= 0 || HW <= 0 || (HW & 3) || ldw < C)
        return DCD_ERR_BAD_ARG;
    if ((int64_t)B * (O > C ? O : C) * HW >= (1ll << 29)) return DCD_ERR_BAD_ARG;
    int nog, ncg, S;
    pw_wrw_partition(B, O, C, HW, nog, ncg, S);
    if (workspace_bytes < (size_t)nog * ncg * S * 64 * 128 * sizeof(float)) return DCD_ERR_WORKSPACE;
    static LdsLimit lds_limit;
    const size_t ldsb = (size_t)PWW_LDS * sizeof(unsigned);
    if (!lds_limit.raise((int)ldsb, pw_wrw_bf16)) return DCD_ERR_LAUNCH;
    hipLaunchKernelGGL(pw_wrw_bf16, dim3(S, ncg, nog), dim3(PWW_NT), ldsb, stream, grad_output, input, (float *)workspace, B, O, C, HW, S);
    hipLaunchKernelGGL(pw_wrw_reduce, dim3(64, 4 * ncg, nog), dim3(256), 0, stream, (const float *)workspace, grad_weight, O, C, ldw, S, ncg);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

// ---- the same layers in exact fp32 (conv1x1_f32.inc)
int dcd_conv1x1_f32(void *stream_, const float *weight, int ldw, int transposed, int n_inputs, const float *const *inputs,
                    const int *channels, float *output, int B, int M, long long HW)
{
    (void)hipGetLastError();
    if (!weight || !inputs || !channels || !output || n_inputs < 1 || n_inputs > 4 || B <= 0 || M <= 0 || HW <= 0 || (HW & 3) || ldw <= 0)
        return DCD_ERR_BAD_ARG;
    PwArgs a;
    a.n_in = n_inputs;
    a.K = 0;
    for (int i = 0; i < 4; ++i) {
        a.in[i] = i < n_inputs ? inputs[i] : nullptr;
        a.ch[i] = i < n_inputs ? channels[i] : 0;
        if (i < n_inputs && (!inputs[i] || channels[i] <= 0 || (channels[i] & 15) || (int64_t)channels[i] * HW >= (1ll << 29)))
            return DCD_ERR_BAD_ARG;
        a.K += a.ch[i];
    }
    if ((int64_t)M * HW >= (1ll << 31)) return DCD_ERR_BAD_ARG;
    a.w = weight; a.ldw = ldw; a.wt = transposed ? 1 : 0; a.out = output; a.M = M; a.HW = HW;
    const int nb = M <= 32 ? 1 : 2;
    dim3 grid((unsigned)((HW + 511) / 512), B, (M + 32 * nb - 1) / (32 * nb));
    if (nb == 2) hipLaunchKernelGGL(pw_conv_f32<2>, grid, dim3(PW_NT), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(pw_conv_f32<1>, grid, dim3(PW_NT), 0, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static void pw_wrw_f32_partition(int B, int O, int C, long long HW, int &nog, int &ncg, int &S)
{
    nog = (O + 63) / 64;
    ncg = (C + 63) / 64;
    const int64_t T = (int64_t)B * ((HW + 63) / 64);
    int64_t s = 2 * (int64_t)device_cus() / (nog * ncg);           // two workgroups per CU
    if (s > T / 4) s = T / 4;
    if (s < 1) s = 1;
    S = (int)s;
}

size_t dcd_conv1x1_wrw_f32_workspace_bytes(int B, int O, int C, long long HW)
{
    if (B <= 0 || O <= 0 || C <= 0 || HW <= 0) return 0;
    int nog, ncg, S;
    pw_wrw_f32_partition(B, O, C, HW, nog, ncg, S);
    return (size_t)nog * ncg * S * 64 * 64 * sizeof(float);
}

int dcd_conv1x1_wrw_f32(void *stream_, const float *grad_output, const float *input, float *grad_weight, int ldw, int B, int O, int C,
                        long long HW, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_output || !input || !grad_weight || !workspace || B <= 0 || O <= 0 || C <= 0 || HW <= 0 || (HW & 3) || ldw < C)
        return DCD_ERR_BAD_ARG;
    if ((int64_t)B * (O > C ? O : C) * HW >= (1ll << 29)) return DCD_ERR_BAD_ARG;
    int nog, ncg, S;
    pw_wrw_f32_partition(B, O, C, HW, nog, ncg, S);
    if (workspace_bytes < (size_t)nog * ncg * S * 64 * 64 * sizeof(float)) return DCD_ERR_WORKSPACE;
    static LdsLimit lds_limit;
    const size_t ldsb = (size_t)PWWF_LDS * sizeof(float);
    if (!lds_limit.raise((int)ldsb, pw_wrw_f32)) return DCD_ERR_LAUNCH;
    hipLaunchKernelGGL(pw_wrw_f32, dim3(S, ncg, nog), dim3(PWW_NT), ldsb, stream, grad_output, input, (float *)workspace, B, O, C, HW, S);
    hipLaunchKernelGGL(pw_wrw_reduce_f32, dim3(64, 4 * ncg, nog), dim3(256), 0, stream, (const float *)workspace, grad_weight, O, C, ldw, S, ncg);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

// ---- 3x3 / stride 2 / pad 1 in exact fp32 (conv_s2_f32.inc)
static bool s2_args_ok(int B, int Cin, int H, int W, int Cout)
{
    return B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 7) == 0 && (Cin & 15) == 0 &&
           (int64_t)Cin * H * W < (1ll << 29) && (int64_t)Cout * (H / 2) * (W / 2) < (1ll << 29);
}

int dcd_conv3x3_s2_f32(void *stream_, const float *input, const float *weight, float *output, int B, int Cin, int H, int W, int Cout)
{
    (void)hipGetLastError();
    if (!input || !weight || !output || !s2_args_ok(B, Cin, H, W, Cout)) return DCD_ERR_BAD_ARG;
    S2Args a;
    a.x = input; a.w = weight; a.y = output; a.B = B; a.C = Cin; a.H = H; a.W = W; a.K = Cout; a.Ho = H / 2; a.Wo = W / 2;
    // four output pixels per lane (512 per workgroup) unless that leaves fewer workgroups than CUs: then two (the small maps)
    const int zb = (Cout + 31) / 32;
    const bool small = (int64_t)((a.Ho * a.Wo + 511) / 512) * B * zb < (int64_t)device_cus();
    if (small) hipLaunchKernelGGL(s2_conv_f32<2>, dim3((unsigned)((a.Ho * a.Wo + 255) / 256), B, zb), dim3(S2_NT), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(s2_conv_f32<4>, dim3((unsigned)((a.Ho * a.Wo + 511) / 512), B, zb), dim3(S2_NT), 0, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_conv3x3_s2_f32_backward_data(void *stream_, const float *grad_output, const float *weight, float *grad_input, int B, int Cin, int H,
                                     int W, int Cout)
{
    (void)hipGetLastError();
    if (!grad_output || !weight || !grad_input || !s2_args_ok(B, Cin, H, W, Cout) || (Cout & 15)) return DCD_ERR_BAD_ARG;
    S2Args a;
    a.x = grad_output; a.w = weight; a.y = grad_input; a.B = B; a.C = Cin; a.H = H; a.W = W; a.K = Cout; a.Ho = H / 2; a.Wo = W / 2;
    const int zb = (Cin + 31) / 32;
    const bool small = (int64_t)((a.Ho * a.Wo + 511) / 512) * B * 2 * zb < (int64_t)device_cus();
    if (small) hipLaunchKernelGGL(s2_dgrad_f32<2>, dim3((unsigned)((a.Ho * a.Wo + 255) / 256), B, 2 * zb), dim3(S2_NT), 0, (hipStream_t)stream_, a);
    else hipLaunchKernelGGL(s2_dgrad_f32<4>, dim3((unsigned)((a.Ho * a.Wo + 511) / 512), B, 2 * zb), dim3(S2_NT), 0, (hipStream_t)stream_, a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static int s2_wrw_splits(int B, int Cin, int H, int W, int Cout)
{
    const int nblk = ((Cout + 31) / 32) * ((Cin + 31) / 32);
    const int64_t groups = (int64_t)B * (H / 2) * (W / 2) / 8;
    int64_t S = (8 * (int64_t)device_cus() + nblk - 1) / nblk;       // two waves per SIMD over the chip
    if (S > groups / 8) S = groups / 8;                               // at least eight iterations per wave
    if (S < 1) S = 1;
    return (int)S;
}

size_t dcd_conv3x3_s2_f32_wrw_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (!s2_args_ok(B, Cin, H, W, Cout)) return 0;
    return (size_t)s2_wrw_splits(B, Cin, H, W, Cout) * 9 * Cout * Cin * sizeof(float);
}

int dcd_conv3x3_s2_f32_wrw(void *stream_, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                           int Cout, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !grad_output || !grad_weight || !workspace || !s2_args_ok(B, Cin, H, W, Cout) || ((H / 2) & 1) || W < 16) return DCD_ERR_BAD_ARG;
    if ((int64_t)B * (Cin > Cout ? (int64_t)Cin * H * W : (int64_t)Cout * (H / 2) * (W / 2)) >= (1ll << 29)) return DCD_ERR_BAD_ARG;
    S2WrwArgs a;
    a.x = input; a.gy = grad_output; a.part = (float *)workspace; a.B = B; a.C = Cin; a.H = H; a.W = W; a.K = Cout; a.Ho = H / 2; a.Wo = W / 2;
    a.S = s2_wrw_splits(B, Cin, H, W, Cout);
    a.ncb = (Cin + 31) / 32;
    if (workspace_bytes < (size_t)a.S * 9 * Cout * Cin * sizeof(float)) return DCD_ERR_WORKSPACE;
    const int items = ((Cout + 31) / 32) * a.ncb * a.S;
    hipLaunchKernelGGL(s2_wrw_f32, dim3((items + 3) / 4), dim3(S2_NT), 0, stream, a);
    const int n = 9 * Cout * Cin;
    hipLaunchKernelGGL(s2_wrw_reduce, dim3((n + 15) / 16), dim3(256), 0, stream, (const float *)workspace, grad_weight, Cout, Cin, a.S);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static void wrw_partition(int B, int Cin, int H, int W, int Cout, int &nog, int &ncg, int &S, int &strips_x, int &KO)
{
    KO = Cout <= 32 ? 32 : 64;
    nog = (Cout + KO - 1) / KO;
    ncg = (Cin + 63) / 64;
    strips_x = (W + 31) / 32;
    const int T = B * (H / 2) * strips_x;
    S = 256 / (nog * ncg);                        // about one workgroup per CU
    // ... but at least four strips per workgroup: every split writes a 16 x 64 x 64 partial (67 MB for 256 of them) that the
    // reduce kernel reads back, which at one image per GPU cost more than the product itself (29 + 17 us at 64 -> 64 @ 96x320)
    if (S > T / 4) S = T / 4;
    if (S >= 8) S &= ~7;
    if (S < 1) S = 1;
    if (S > T) S = T;
}

size_t dcd_conv3x3_wrw_workspace_bytes(int B, int Cin, int H, int W, int Cout)
{
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
    int nog, ncg, S, sx, KO;
    wrw_partition(B, Cin, H, W, Cout, nog, ncg, S, sx, KO);
    return (size_t)nog * ncg * S * 12 * KO * 64 * sizeof(float);
}

int dcd_conv3x3_wrw(void *stream_, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                    int Cout, int precision, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !grad_output || !grad_weight || !workspace || B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return DCD_ERR_BAD_ARG;
    if (precision != DCD_PREC_F32 && precision != DCD_PREC_BF16X3 && precision != DCD_PREC_BF16) return DCD_ERR_BAD_ARG;
    const bool bf = precision == DCD_PREC_BF16;              // the split-bf16 form has no variant of this kernel: it runs exact fp32
    if ((W & 3) || (H & 1) || (int64_t)(Cin > Cout ? Cin : Cout) * H * W >= (1ll << 31)) return DCD_ERR_BAD_ARG;
    int nog, ncg, S, strips_x, KO;
    wrw_partition(B, Cin, H, W, Cout, nog, ncg, S, strips_x, KO);
    const int nblk = nog * ncg;
    if (workspace_bytes < (size_t)nblk * S * 12 * KO * 64 * sizeof(float)) return DCD_ERR_WORKSPACE;
    static const bool wrw_direct = !(dcd_env("DCD_CONV_WRW_DIRECT") && atoi(dcd_env("DCD_CONV_WRW_DIRECT")) == 0);
    if (bf && KO == 64 && wrw_direct && (int64_t)B * (Cin > Cout ? Cin : Cout) * H * W < (1ll << 29)) {
        // one-product form, more than 32 outputs: the direct kernel (conv_direct_bf16.inc), same partition and workspace
        static LdsLimit lds_limit;
        const size_t ldsb = (size_t)DW_LDS * sizeof(unsigned);
        if (!lds_limit.raise((int)ldsb, conv3x3_wrw_direct_bf16)) return DCD_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_wrw_direct_bf16, dim3(nblk * S), dim3(DW_NT), ldsb, stream, input, grad_output, (float *)workspace, Cin,
                           Cout, B, H, W, strips_x, S, ncg, nblk);
        hipLaunchKernelGGL(conv3x3_wrw_direct_reduce, dim3(64, 3, nblk), dim3(256), 0, stream, (const float *)workspace, grad_weight, Cin, Cout,
                           S, ncg);
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    if (KO == 64) {
        static LdsLimit lds_limit;
        const size_t ldsb = (size_t)2 * WrwGeom<2>::BUF * sizeof(float);
        if (!lds_limit.raise((int)ldsb, wino_wrw3x3_f32<2, false>, wino_wrw3x3_f32<2, true>)) return DCD_ERR_LAUNCH;
        if (bf)
            hipLaunchKernelGGL((wino_wrw3x3_f32<2, true>), dim3(nblk * S), dim3(WW_NT), ldsb, stream, input, grad_output, (float *)workspace,
                               Cin, Cout, B, H, W, strips_x, S, ncg, nblk);
        else
            hipLaunchKernelGGL((wino_wrw3x3_f32<2, false>), dim3(nblk * S), dim3(WW_NT), ldsb, stream, input, grad_output, (float *)workspace,
                               Cin, Cout, B, H, W, strips_x, S, ncg, nblk);
    } else {
        static LdsLimit lds_limit;
        const size_t ldsb = (size_t)2 * WrwGeom<1>::BUF * sizeof(float);
        if (!lds_limit.raise((int)ldsb, wino_wrw3x3_f32<1, false>, wino_wrw3x3_f32<1, true>)) return DCD_ERR_LAUNCH;
        if (bf)
            hipLaunchKernelGGL((wino_wrw3x3_f32<1, true>), dim3(nblk * S), dim3(WW_NT), ldsb, stream, input, grad_output, (float *)workspace,
                               Cin, Cout, B, H, W, strips_x, S, ncg, nblk);
        else
            hipLaunchKernelGGL((wino_wrw3x3_f32<1, false>), dim3(nblk * S), dim3(WW_NT), ldsb, stream, input, grad_output, (float *)workspace,
                               Cin, Cout, B, H, W, strips_x, S, ncg, nblk);
    }
    hipLaunchKernelGGL(wino_wrw_reduce, dim3(nblk * KO * 64 / 16), dim3(256), 0, stream, (const float *)workspace, grad_weight, Cin, Cout,
                       S, ncg, KO);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
