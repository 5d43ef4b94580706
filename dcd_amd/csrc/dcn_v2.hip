// dcn_v2.hip -- fused modulated deformable convolution (DCNv2) for gfx950 / MI355X.
//
// Replaces the reference's im2col -> column buffer -> (batched) SGEMM pipeline
// (DGDE/model/backbone/DCNv2/DCN/src/cuda/dcn_v2_cuda.cu:42-341 and
//  cuda/dcn_v2_im2col_cuda.cu:125-327) with three kernels in which the column matrix never
// exists in memory:
//
//   forward      out[o,p]  = bias[o] + sum_k W[o,k] col[k,p]          (gather -> MFMA)
//   backward-data dcol[k,p] = sum_o W[o,k] dY[o,p]  (MFMA) -> grad_offset, grad_mask, grad_input
//   backward-wgt dW[o,k]   = sum_{b,p} dY[o,p] col[k,p]               (gather -> LDS -> MFMA)
//
// Design (wave64, one wave = 32 output pixels):
//  * The K axis of every contraction is re-ordered tap-major: k' = ((group*KK + tap)*cpgp + c).
//    All channels of a tap share the same four bilinear corners, so the corner indices and
//    weights are computed once per (pixel, tap) and live in registers across the channel loop.
//  * v_mfma_f32_32x32x2_f32 takes one f32 VGPR per operand with lane l holding B[k=l>>5][j=l&31].
//    Lane l therefore gathers the sample for pixel (l&31) and channel parity (l>>5): the value it
//    computes IS its MFMA B operand, so the forward needs no LDS and no column buffer at all.
//  * Weights are re-laid out once per call into Wf[k'][o] (forward A operand, coalesced over o)
//    and Wb[o][k'] (backward-data A operand, coalesced over k') in the caller's workspace.
//  * grad_input is a GATHER, not the reference's col2im scatter: `dcn_build_inverse` lists, per input cell and tap, the
//    output pixels whose sample touches the cell, and `dcn_bwd_input_*` contracts W with those dY values on the matrix
//    pipe and stores plainly (atomics remain only as the fallback for samples beyond the list radius / overflowed cells).
//    grad_weight leaves the tiled kernel as per-workgroup partials summed in a fixed order by `dcn_dw_reduce`;
//    grad_bias is its own two-stage reduction.
//
// Sample validity follows the reference exactly: a tap contributes iff -1 < h < H and -1 < w < W
// (cuda/dcn_v2_im2col_cuda.cu:180), each corner iff it lies inside the image (:38-48).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <map>
#include <vector>
#include <mutex>
#include <utility>

#include "../../include/dcd_hip.h"
#include "tuning_env.h"
#include "lds_limit.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

namespace {

template <int N> struct IntC {
    static constexpr int value = N;
};

struct Geom {
    int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, dg;
    int Ho, Wo, HoWo, KK, cpg, cpgp, Kp, Cop;  // cpgp: channels/group padded to 32; Kp = dg*KK*cpgp; Cop: Co padded to 32
};

__host__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

__host__ inline bool make_geom(Geom &g, int B, int C, int H, int W, int Co, int kh, int kw, int sh, int sw, int ph,
                               int pw, int dh, int dw, int dg)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Co <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 ||
        pw < 0 || dh <= 0 || dw <= 0 || dg <= 0 || C % dg)
        return false;
    g.B = B; g.C = C; g.H = H; g.W = W; g.Co = Co; g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw;
    g.ph = ph; g.pw = pw; g.dh = dh; g.dw = dw; g.dg = dg;
    g.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    g.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    if (g.Ho <= 0 || g.Wo <= 0) return false;
    g.HoWo = g.Ho * g.Wo;
    g.KK = kh * kw;
    g.cpg = C / dg;
    g.cpgp = round_up(g.cpg, 32);
    g.Kp = dg * g.KK * g.cpgp;
    g.Cop = round_up(Co, 32);
    // 32-bit index safety for per-image planes
    if ((int64_t)C * H * W >= (1ll << 31) || (int64_t)Co * g.HoWo >= (1ll << 31) ||
        (int64_t)g.Kp * g.Cop >= (1ll << 31) || (int64_t)dg * 2 * g.KK * g.HoWo >= (1ll << 31))
        return false;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Weight re-layout: W[o][c][t] -> Wf[k'][Cop], Wb[Cop][Kp], zero padded.
// ---------------------------------------------------------------------------------------------
// (bid, nb: the block's index and the number of blocks doing this job -- the fused prologue kernels of the one-pass backward give a
// launch's blocks different jobs)
__device__ __forceinline__ void prep_weights_body(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wb, const Geom &g,
                                                  int bid, int nb)
{
    const int n = g.Kp * g.Cop;
    for (int idx = bid * blockDim.x + threadIdx.x; idx < n; idx += nb * blockDim.x) {
        // idx enumerates Wb (o-major) so that reads of w are roughly coalesced over k'
        const int o = idx / g.Kp, kp = idx - o * g.Kp;
        const int seg = kp / g.cpgp, cc = kp - seg * g.cpgp;  // seg = group*KK + tap
        const int grp = seg / g.KK, t = seg - grp * g.KK;
        float v = 0.f;
        if (o < g.Co && cc < g.cpg) v = w[((size_t)o * g.C + grp * g.cpg + cc) * g.KK + t];
        wb[idx] = v;
        wf[(size_t)kp * g.Cop + o] = v;
    }
}
__global__ void dcn_prep_weights(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wb, Geom g)
{
    prep_weights_body(w, wf, wb, g, blockIdx.x, gridDim.x);
}

// Per-(pixel, tap) sampling state shared by all channels of a deformable group.
struct Tap {
    int i1, i2, i3, i4;      // corner element offsets inside one H*W plane (0 when the corner is unused)
    float w1, w2, w3, w4;    // bilinear weights, zero for corners outside the image / invalid samples
    float lh, lw, hh, hw;    // fractional parts (for the coordinate gradient)
    float m;                 // modulation mask (0 when the sample is invalid or the pixel is padding)
    float oh, ow;            // raw offsets of this tap
    // pair form: the two corners of a row are adjacent floats -> ONE dwordx2 load per row.  pt / pb are element offsets of
    // the (x, x+1) pairs of the top / bottom row, clamped into the plane; a0,a1,b0,b1 are the bilinear weights re-mapped
    // onto the loaded pair (at the left / right image edge the valid corner moves to the other element).
    int pt, pb;
    float a0, a1, b0, b1;
    int shift;               // 0: pair = (corner 1|3, corner 2|4); +1: first element is corner 2|4; -1: second element is corner 1|3
    bool c1, c2, c3, c4;     // corner validity
};

struct TapRaw { float oh, ow, m; };

__device__ __forceinline__ TapRaw load_tap_raw(const float *__restrict__ off_b, const float *__restrict__ msk_b, const Geom &g,
                                               int seg, int Pc)
{
    TapRaw r;
    r.oh = off_b[(size_t)(2 * seg) * g.HoWo + Pc];
    r.ow = off_b[(size_t)(2 * seg + 1) * g.HoWo + Pc];
    r.m = msk_b[(size_t)seg * g.HoWo + Pc];
    return r;
}

__device__ __forceinline__ Tap finish_tap(const TapRaw &raw, const Geom &g, int t, int ho, int wo, bool pv)
{
    Tap s;
    const int i = t / g.kw, j = t - i * g.kw;
    const float oh = raw.oh, ow = raw.ow, m = raw.m;
    const float hf = (float)(ho * g.sh - g.ph + i * g.dh) + oh;
    const float wf = (float)(wo * g.sw - g.pw + j * g.dw) + ow;
    const bool sv = pv && hf > -1.f && wf > -1.f && hf < (float)g.H && wf < (float)g.W;
    const float hlf = sv ? floorf(hf) : 0.f, wlf = sv ? floorf(wf) : 0.f;
    const int hl = (int)hlf, wl = (int)wlf, hh_i = hl + 1, wh_i = wl + 1;
    s.lh = sv ? hf - hlf : 0.f;
    s.lw = sv ? wf - wlf : 0.f;
    s.hh = 1.f - s.lh;
    s.hw = 1.f - s.lw;
    s.c1 = sv && hl >= 0 && wl >= 0;
    s.c2 = sv && hl >= 0 && wh_i <= g.W - 1;
    s.c3 = sv && hh_i <= g.H - 1 && wl >= 0;
    s.c4 = sv && hh_i <= g.H - 1 && wh_i <= g.W - 1;
    s.i1 = s.c1 ? hl * g.W + wl : 0;
    s.i2 = s.c2 ? hl * g.W + wh_i : 0;
    s.i3 = s.c3 ? hh_i * g.W + wl : 0;
    s.i4 = s.c4 ? hh_i * g.W + wh_i : 0;
    s.w1 = s.c1 ? s.hh * s.hw : 0.f;
    s.w2 = s.c2 ? s.hh * s.lw : 0.f;
    s.w3 = s.c3 ? s.lh * s.hw : 0.f;
    s.w4 = s.c4 ? s.lh * s.lw : 0.f;
    s.m = sv ? m : 0.f;
    s.oh = oh;
    s.ow = ow;
    {
        const int xb = g.W >= 2 ? min(max(wl, 0), g.W - 2) : 0;
        const int ya = min(max(hl, 0), g.H - 1), yb = min(max(hh_i, 0), g.H - 1);
        const int shift = xb - wl;          // 0 inside, +1 at the left edge (wl == -1), -1 at the right edge (wl == W-1)
        s.pt = ya * g.W + xb;
        s.pb = yb * g.W + xb;
        s.shift = shift;
        s.a0 = shift == 0 ? s.w1 : (shift == 1 ? s.w2 : 0.f);
        s.a1 = shift == 0 ? s.w2 : (shift == -1 ? s.w1 : 0.f);
        s.b0 = shift == 0 ? s.w3 : (shift == 1 ? s.w4 : 0.f);
        s.b1 = shift == 0 ? s.w4 : (shift == -1 ? s.w3 : 0.f);
        if (!sv) { s.pt = 0; s.pb = 0; }
    }
    return s;
}

__device__ __forceinline__ Tap make_tap(const float *__restrict__ off_b, const float *__restrict__ msk_b, const Geom &g,
                                        int seg, int t, int ho, int wo, int Pc, bool pv)
{
    return finish_tap(load_tap_raw(off_b, msk_b, g, seg, Pc), g, t, ho, wo, pv);
}

// Workgroups are dealt to the 8 XCDs round-robin by linear id (observed, speed only).  Re-map (blockIdx.x, blockIdx.y)
// = (tile group, image) so that each XCD walks one CONTIGUOUS run of tiles: neighbouring tiles share their input halo,
// and a run's footprint then fits that XCD's 4 MiB L2 instead of every XCD touching every image row.  Bijective for any grid.
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

// Uniform base + 32-bit per-lane BYTE offset -> `global_load_dword v, v_off, s[base:base+1]`.
__device__ __forceinline__ float ldg(const float *base, unsigned byte_off)
{
    return *(const float *)((const char *)base + byte_off);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// 8-byte load at a 4-byte aligned address (two horizontally adjacent corners)
__device__ __forceinline__ f32x2 ldg2(const float *base, unsigned byte_off)
{
    typedef f32x2 __attribute__((aligned(4))) f32x2_u;
    return *(const f32x2_u *)((const char *)base + byte_off);
}

// ---------------------------------------------------------------------------------------------
// Forward.  grid = (ceil(tiles/4), B, ceil(Cop/32/MB)); block = 256 (4 waves, one 32-pixel tile each).
// ---------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(256) void dcn_fwd_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                   const float *__restrict__ msk, const float *__restrict__ wf,
                                                   const float *__restrict__ bias, float *__restrict__ out, Geom g)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int tile = bx * 4 + wave;
    if (tile * 32 >= g.HoWo) return;  // wave-uniform
    const int ob0 = blockIdx.z * MB;
    const int P = tile * 32 + p;
    const bool pv = P < g.HoWo;
    const int Pc = pv ? P : g.HoWo - 1;
    const int ho = Pc / g.Wo, wo = Pc - ho * g.Wo;
    const unsigned HW4 = (unsigned)(g.H * g.W) * 4u;
    const unsigned Cop4 = (unsigned)g.Cop * 4u;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    const float *in_b = in + (size_t)b * g.C * g.H * g.W;
    const float *off_b = off + (size_t)b * g.dg * 2 * g.KK * g.HoWo;
    const float *msk_b = msk + (size_t)b * g.dg * g.KK * g.HoWo;
    const int npair = g.cpg >> 1;
    const unsigned wlane = ((unsigned)h * (unsigned)g.Cop + (unsigned)(ob0 * 32 + p)) * 4u;

    for (int seg = 0; seg < g.dg * g.KK; ++seg) {
        const int grp = seg / g.KK, t = seg - grp * g.KK;
        const Tap s = make_tap(off_b, msk_b, g, seg, t, ho, wo, Pc, pv);
        const float *ip = in_b + (size_t)grp * g.cpg * g.H * g.W;       // uniform, advanced 2 planes per step
        const float *wp = wf + (size_t)seg * g.cpgp * g.Cop;              // uniform, advanced 2 rows per step
        const unsigned o1 = (unsigned)s.i1 * 4u + h * HW4, o2 = (unsigned)s.i2 * 4u + h * HW4;
        const unsigned o3 = (unsigned)s.i3 * 4u + h * HW4, o4 = (unsigned)s.i4 * 4u + h * HW4;
        const unsigned opt = (unsigned)s.pt * 4u + h * HW4, opb = (unsigned)s.pb * 4u + h * HW4;
        const bool pairs = g.W >= 2;         // uniform
        const float q1 = pairs ? s.a0 : s.w1, q2 = pairs ? s.a1 : s.w2, q3 = pairs ? s.b0 : s.w3, q4 = pairs ? s.b1 : s.w4;
        // Two register stages of UN channel pairs each: the loads of stage B are in flight while stage A feeds the
        // matrix pipe (the gather addresses depend only on the tap, never on loaded data, so they prefetch freely).
#ifndef DCN_FWD_UN
#define DCN_FWD_UN ((MB >= 4) ? 2 : 4)
#endif
        constexpr int UN = DCN_FWD_UN;
        float va[UN][4], vb[UN][4], wa[UN][MB], wb_[UN][MB];
        auto issue = [&](float (&v)[UN][4], float (&a)[UN][MB], int it0) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float *ipc = (const float *)((const char *)ip + (size_t)(it0 + u) * 2 * HW4);
                const float *wpc = (const float *)((const char *)wp + (size_t)(it0 + u) * 2 * Cop4);
#ifdef DCN_ABL_NOGATHER
                v[u][0] = s.w1 + (float)u; v[u][1] = s.w2; v[u][2] = s.w3; v[u][3] = s.w4; (void)ipc;
#else
                if (pairs) {
                    const f32x2 tp = ldg2(ipc, opt), bt = ldg2(ipc, opb);
                    v[u][0] = tp.x; v[u][1] = tp.y; v[u][2] = bt.x; v[u][3] = bt.y;
                } else {
                    v[u][0] = ldg(ipc, o1); v[u][1] = ldg(ipc, o2); v[u][2] = ldg(ipc, o3); v[u][3] = ldg(ipc, o4);
                }
#endif
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#ifdef DCN_ABL_NOWEIGHT
                    a[u][mb] = s.w1 + (float)(mb + u); (void)wpc;
#else
                    a[u][mb] = ldg(wpc, wlane + mb * 128u);
#endif
            }
        };
        auto compute = [&](float (&v)[UN][4], float (&a)[UN][MB]) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const float val = (q1 * v[u][0] + q2 * v[u][1] + q3 * v[u][2] + q4 * v[u][3]) * s.m;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mb], val, acc[mb], 0, 0, 0);
            }
        };
        const int nfull = npair / UN;
        int gq = 0;
        if (nfull > 0) issue(va, wa, 0);
        for (; gq + 1 < nfull; gq += 2) {
            issue(vb, wb_, (gq + 1) * UN);
            compute(va, wa);
            if (gq + 2 < nfull) issue(va, wa, (gq + 2) * UN);
            compute(vb, wb_);
        }
        if (gq < nfull) compute(va, wa);
        for (int it = nfull * UN; it < npair; ++it) {      // remainder pairs (channel counts not divisible by 2*UN)
            const float *ipc = (const float *)((const char *)ip + (size_t)it * 2 * HW4);
            const float *wpc = (const float *)((const char *)wp + (size_t)it * 2 * Cop4);
            const float v1 = ldg(ipc, o1), v2 = ldg(ipc, o2), v3 = ldg(ipc, o3), v4 = ldg(ipc, o4);
            const float val = (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * s.m;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ldg(wpc, wlane + mb * 128u), val, acc[mb], 0, 0, 0);
        }
        ip = (const float *)((const char *)ip + (size_t)npair * 2 * HW4);
        wp = (const float *)((const char *)wp + (size_t)npair * 2 * Cop4);
        if (g.cpg & 1) {  // odd channel count: the h==1 half has no channel left (its weight row is zero padding)
            const float v1 = ldg(ip, (unsigned)s.i1 * 4u), v2 = ldg(ip, (unsigned)s.i2 * 4u);
            const float v3 = ldg(ip, (unsigned)s.i3 * 4u), v4 = ldg(ip, (unsigned)s.i4 * 4u);
            float val = (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * s.m;
            val = h ? 0.f : val;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ldg(wp, wlane + mb * 128u), val, acc[mb], 0, 0, 0);
        }
    }

    float *out_b = out + (size_t)b * g.Co * g.HoWo;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int obase = (ob0 + mb) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = obase + (r & 3) + 8 * (r >> 2);
            if (pv && o < g.Co) out_b[(size_t)o * g.HoWo + P] = acc[mb][r] + bias[o];
        }
    }
}

// ---- constants of the workgroup-tiled forward (defined here: the rescue mode of dcn_fwd9_f32 reads its weight layout)
constexpr int TL_WW = 40;                   // window cols  c0-4 .. c0+35 (16-byte aligned rows)
constexpr int TL_CH = 8;                    // channels per chunk
constexpr int TL_OB = 64;                   // output channels per workgroup (two 32-wide MFMA blocks)
constexpr int TL_W_FLOATS = 9 * TL_OB * TL_CH;            // 4608: [tap][o][h][4 steps]
constexpr int TL_NW = TL_W_FLOATS / 4;                    // 1152 dwordx4 per chunk (weights)
// TR = waves per workgroup = tile rows (8: 16-row window, 1 workgroup / CU; 4: 12-row window, 2 workgroups / CU)
template <int TR> struct TileCfg {
    static constexpr int WH = TR + 8;                             // window rows r0-4 .. r0+TR+3
    static constexpr int PLANE = (WH * TL_WW) % 64 == 32 ? WH * TL_WW : WH * TL_WW + 32;   // odd/even planes 32 banks apart
    static constexpr int IN_FLOATS = TL_CH * PLANE;
    static constexpr int BUF = IN_FLOATS + TL_W_FLOATS;           // floats per buffer
    static constexpr int NIN = TL_CH * WH * (TL_WW / 4);          // dwordx4 per chunk (window)
    static constexpr int NT = TR * 64;                            // threads
    static constexpr int KIN = (NIN + NT - 1) / NT, KW = (TL_NW + NT - 1) / NT;
};


// ---------------------------------------------------------------------------------------------
// Forward, 3x3 fast path: CHANNEL-outer / TAP-inner.
// The generic kernel above walks every channel plane once per tap, so a wave's reuse distance is Cin planes and
// its gathers miss L1 (measured: pipeline depth made no difference, 42 TF).  Here the sampling state of all nine taps
// (4 byte offsets + 4 mask-scaled weights each = 72 VGPRs) stays in registers and the loop runs over channel pairs:
// the 36 gathers of a pair touch the same ~3 rows of one plane, i.e. a handful of L1 lines.  K order is then the
// weight's native k = c*9 + tap, so Wf9[k][Cout_pad] is a plain transpose.
// ---------------------------------------------------------------------------------------------
__global__ void dcn_prep_weights9(const float *__restrict__ w, float *__restrict__ wf9, Geom g)
{
    const int K = g.C * 9, n = K * g.Cop;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int k = idx / g.Cop, o = idx - k * g.Cop;
        wf9[idx] = (o < g.Co) ? w[(size_t)o * K + k] : 0.f;
    }
}

// RESCUE = true: second pass of the tiled forward.  Waves map to (row, 32-column segment) of the tiled kernel's regions and
// run only where that kernel flagged its region as dominated by far samples (rescue_flags, rows_per_region).
template <int MB, bool RESCUE = false>
__global__ __launch_bounds__(256) void dcn_fwd9_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                    const float *__restrict__ msk, const float *__restrict__ wf9,
                                                    const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                    const unsigned char *__restrict__ rescue_flags = nullptr, int tiles_x = 0,
                                                    int rows_per_region = 0, int nchunk = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int tile = bx * 4 + wave;
    const int ob0 = blockIdx.z * MB;
    int P, ho, wo;
    bool pv;
    if (RESCUE) {
        const int row = tile / tiles_x, cseg = tile - row * tiles_x;
        if (row >= g.Ho) return;
        const int regions_y = (g.Ho + rows_per_region - 1) / rows_per_region;
        if (!rescue_flags[((size_t)b * regions_y + row / rows_per_region) * tiles_x + cseg]) return;
        ho = row;
        wo = cseg * 32 + p;
        pv = wo < g.Wo;
        if (!pv) wo = g.Wo - 1;
        P = ho * g.Wo + wo;
    } else {
        if (tile * 32 >= g.HoWo) return;
        P = tile * 32 + p;
        pv = P < g.HoWo;
        if (!pv) P = g.HoWo - 1;
        ho = P / g.Wo;
        wo = P - ho * g.Wo;
    }
    const int Pc = P;
    const unsigned HW4 = (unsigned)(g.H * g.W) * 4u;
    const unsigned Cop4 = (unsigned)g.Cop * 4u;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    const float *in_b = in + (size_t)b * g.C * g.H * g.W;
    const float *off_b = off + (size_t)b * g.dg * 18 * g.HoWo;
    const float *msk_b = msk + (size_t)b * g.dg * 9 * g.HoWo;
    const unsigned wlane = (unsigned)(ob0 * 32 + p) * 4u;
    const int npair = g.cpg >> 1;

    for (int grp = 0; grp < g.dg; ++grp) {
        unsigned to[9][2];      // byte offsets of the top / bottom (x, x+1) pairs
        float tw[9][4];         // pair-mapped bilinear weights x mask
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const Tap s = make_tap(off_b, msk_b, g, grp * 9 + t, t, ho, wo, Pc, pv);
            to[t][0] = (unsigned)s.pt * 4u; to[t][1] = (unsigned)s.pb * 4u;
            tw[t][0] = s.a0 * s.m; tw[t][1] = s.a1 * s.m; tw[t][2] = s.b0 * s.m; tw[t][3] = s.b1 * s.m;
        }
        // lane half h handles channel 2*it + h of the group
        const float *ip = in_b + ((size_t)grp * g.cpg + h) * g.H * g.W;
        const float *wp = wf9 + ((size_t)grp * g.cpg + h) * 9 * g.Cop;
        auto pair_step = [&](const float *ipc, const float *wpc, bool live) {
#pragma unroll
            for (int t0 = 0; t0 < 9; t0 += 3) {      // three taps per stage: 6 pair loads + 3*MB weight loads in flight
                f32x2 vt[3], vb[3];
                float a[3][MB];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    vt[u] = ldg2(ipc, to[t0 + u][0]);
                    vb[u] = ldg2(ipc, to[t0 + u][1]);
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) a[u][mb] = ldg(wpc, wlane + (unsigned)(t0 + u) * Cop4 + mb * 128u);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    float val = tw[t0 + u][0] * vt[u].x + tw[t0 + u][1] * vt[u].y + tw[t0 + u][2] * vb[u].x + tw[t0 + u][3] * vb[u].y;
                    val = live ? val : 0.f;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mb], val, acc[mb], 0, 0, 0);
                }
            }
        };
        for (int it = 0; it < npair; ++it)
            pair_step((const float *)((const char *)ip + (size_t)it * 2 * HW4),
                      (const float *)((const char *)wp + (size_t)it * 18 * Cop4), true);
        if (g.cpg & 1) {      // odd channel count: only the h == 0 half has a channel left
            const size_t back = h ? (size_t)g.H * g.W : 0;          // keep the h == 1 addresses inside the tensor
            const size_t wback = h ? (size_t)9 * g.Cop : 0;
            pair_step((const float *)((const char *)ip + (size_t)npair * 2 * HW4) - back,
                      (const float *)((const char *)wp + (size_t)npair * 18 * Cop4) - wback, h == 0);
        }
    }

    float *out_b = out + (size_t)b * g.Co * g.HoWo;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int obase = (ob0 + mb) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = obase + (r & 3) + 8 * (r >> 2);
            if (pv && o < g.Co) out_b[(size_t)o * g.HoWo + Pc] = acc[mb][r] + bias[o];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward, 3x3 / stride 1 / pad 1 / dilation 1 / one deformable group: workgroup-tiled, LDS-resident gathers.
//
// Measured on MI355X (tools/micro/gather_rate.hip): a vector-memory instruction whose 64 lane addresses are strictly
// consecutive issues in ~4.5 clk per CU; the same instruction with bilinear-corner addresses (consecutive +/- a per-lane
// jitter) takes the 4-lanes-per-clock path: 16.3 clk, dword or dwordx2 alike, and ~36 clk inside the register-gather
// kernels once L1 misses are added.  64->64 @ 96x320 x 8 issues 4.4 M such gathers per forward: that, not MFMA, is the
// 0.26 ms.  LDS has no such path (128 B/clk per CU for any conflict-free pattern), so here:
//   * a workgroup of 8 waves owns an 8 x 32 pixel tile (one row segment per wave);
//   * per chunk of 8 input channels the 16 x 40 window [r0-4, r0+12) x [c0-4, c0+36) is copied to LDS with aligned
//     dwordx4 loads (fast path; cells outside the image become zeros = the reference's zero padding), together with the
//     chunk's weight slab, laid out so that a lane's four A operands of a tap are ONE ds_read_b128;
//   * chunks are double-buffered through registers: loads of chunk k+1 are issued before the MFMA work of chunk k and
//     written to the other LDS buffer after it -- one barrier per chunk;
//   * each bilinear corner pair is one ds_read2_b32.  Samples displaced by 3 px or more fall outside the window and are
//     gathered from global memory by their lane after the chunk loop (correct for any offset, fast for realistic ones).
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wl[z][chunk][tap][o(64)][h(2)][s(4)] = W[z*64+o][chunk*8 + 2s + h][tap]   (zero padded)
__global__ void dcn_prep_weights_tile(const float *__restrict__ w, float *__restrict__ wl, Geom g, int nchunk, int nz,
                                      unsigned *__restrict__ flags, int nflag_words, float *__restrict__ wf9)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nflag_words; i += gridDim.x * blockDim.x) flags[i] = 0u;
    if (wf9) {                                   // weights of the rescue pass (dcn_fwd9_f32 layout [c*9+t][Cout_pad])
        const int K = g.C * 9, n9 = K * g.Cop;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n9; idx += gridDim.x * blockDim.x) {
            const int k = idx / g.Cop, o = idx - k * g.Cop;
            wf9[idx] = (o < g.Co) ? w[(size_t)o * K + k] : 0.f;
        }
    }
    const int n = nz * nchunk * TL_W_FLOATS;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        int r = idx;
        const int s = r & 3; r >>= 2;
        const int h = r & 1; r >>= 1;
        const int o = r % TL_OB; r /= TL_OB;
        const int t = r % 9; r /= 9;
        const int ck = r % nchunk, z = r / nchunk;
        const int oo = z * TL_OB + o, c = ck * TL_CH + 2 * s + h;
        wl[idx] = (oo < g.Co && c < g.C) ? w[((size_t)oo * g.C + c) * 9 + t] : 0.f;
    }
}

// Per-call decision for large learned offsets (forward): counts the offset coordinates displaced by TL_NEAR_F px or more.
// When they exceed 1/32 of all coordinates every workgroup of the tiled kernel hands its region to the rescue kernel at once
// (no per-region mix of per-lane fallbacks and rescues, which is the slow regime: DESIGN.md section 4).
constexpr float TL_NEAR_F = 3.f;
__global__ __launch_bounds__(256) void dcn_fwd_far_count(const float *__restrict__ off, int64_t n, unsigned *__restrict__ counter)
{
    unsigned c = 0;
    const int64_t n4 = ((uintptr_t)off & 15) == 0 ? n >> 2 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(off + 4 * i);
        c += (!(fabsf(v.x) < TL_NEAR_F)) + (!(fabsf(v.y) < TL_NEAR_F)) + (!(fabsf(v.z) < TL_NEAR_F)) + (!(fabsf(v.w) < TL_NEAR_F));
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        c += !(fabsf(off[i]) < TL_NEAR_F);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __shared__ unsigned part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && part[0] + part[1] + part[2] + part[3]) atomicAdd(counter, part[0] + part[1] + part[2] + part[3]);
}

template <int TR>
__global__ __launch_bounds__(TR * 64) void dcn_fwd_tile_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                        const float *__restrict__ msk, const float *__restrict__ wl,
                                                        const float *__restrict__ bias, float *__restrict__ out, Geom g,
                                                        int tiles_x, int nchunk, unsigned char *__restrict__ rescue_flags,
                                                        const float *__restrict__ wf9, int rescue_taps,
                                                        const unsigned *__restrict__ far_count, unsigned far_limit)
{
    typedef TileCfg<TR> T;
    constexpr int TL_ROWS = TR, TL_WH = T::WH, TL_PLANE = T::PLANE, TL_IN_FLOATS = T::IN_FLOATS, TL_BUF = T::BUF,
                  TL_NIN = T::NIN, NT = T::NT, KIN = T::KIN, KW = T::KW;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // 2 x TL_BUF
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    if (*far_count > far_limit) {                              // far samples dominate this call: every region goes to the rescue kernel
        if (tid == 0) rescue_flags[(size_t)b * gridDim.x + bx] = 1;
        return;
    }
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int r0 = ty * TL_ROWS, c0 = tx * 32;
    const int ho = r0 + wave, wo = c0 + p;
    const bool pv = ho < g.Ho && wo < g.Wo;
    const int Pc = pv ? ho * g.Wo + wo : 0;
    const int z = blockIdx.z;
    const int HW = g.H * g.W;
    const int Y0 = r0 - 4, X0 = c0 - 4;

    const float *in_b = in + (size_t)b * g.C * HW;
    const float *off_b = off + (size_t)b * 18 * g.HoWo;
    const float *msk_b = msk + (size_t)b * 9 * g.HoWo;
    const float *wl_z = wl + (size_t)z * nchunk * TL_W_FLOATS;

    // ---- staging map (chunk invariant): three window dwordx4 + three weight dwordx4 per thread
    int sg[KIN], sl[KIN];            // global float offset inside the chunk's 8 planes (-1: zeros), LDS float offset
    bool sv_[KIN];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const int e = tid + NT * k;
        const int ch = e / (TL_WH * 10), rem = e - ch * (TL_WH * 10);
        const int row = rem / 10, q = rem - row * 10;
        const int y = Y0 + row, x = X0 + 4 * q;
        sv_[k] = e < TL_NIN;
        sg[k] = (sv_[k] && y >= 0 && y < g.H && x >= 0 && x < g.W) ? ch * HW + y * g.W + x : -1;
        sl[k] = ch * TL_PLANE + row * TL_WW + 4 * q;
    }
    f32x4 rin[KIN];
    // weight slab of chunk ck -> buffer `buf`: a straight copy, global -> LDS directly (16 bytes per lane, LDS address = wave base
    // + 16 lane: no staging registers, no ds_write); vmcnt(0) before the barrier that publishes the buffer
    auto issue_w = [&](int ck, float *buf) {
        const float *src = wl_z + (size_t)ck * TL_W_FLOATS + 4 * tid;
        float *dst = buf + TL_IN_FLOATS + 4 * 64 * __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
        for (int k = 0; k < KW; ++k)
            if (64 * __builtin_amdgcn_readfirstlane(wave) + NT * k < TL_NW)
                __builtin_amdgcn_global_load_lds(src + 4 * NT * k, (__attribute__((address_space(3))) void *)(dst + 4 * NT * k), 16, 0, 0);
    };
    auto issue = [&](int ck) {
        const float *src = in_b + (size_t)ck * TL_CH * HW;
        const int cleft = g.C - ck * TL_CH;                      // channels left (may be < 8 in the last chunk)
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            rin[k] = zero4;
            if (sv_[k] && sg[k] >= 0 && (tid + NT * k) / (TL_WH * 10) < cleft) rin[k] = *reinterpret_cast<const f32x4 *>(src + sg[k]);
        }
    };
    auto commit = [&](float *buf) {
#pragma unroll
        for (int k = 0; k < KIN; ++k)
            if (sv_[k]) *reinterpret_cast<f32x4 *>(buf + sl[k]) = rin[k];
    };

    issue(0);

    // ---- sampling state of the nine taps of this lane's pixel (while the first chunk is in flight)
    int lo[9];
    float wq[9][4];
    unsigned farbits = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const TapRaw raw = load_tap_raw(off_b, msk_b, g, t, Pc);
        const float hf = (float)(ho - 1 + t / 3) + raw.oh, wf_ = (float)(wo - 1 + t % 3) + raw.ow;
        const bool sv = pv && hf > -1.f && wf_ > -1.f && hf < (float)g.H && wf_ < (float)g.W;
        const float hlf = floorf(hf), wlf = floorf(wf_);
        const int ly = sv ? (int)hlf - Y0 : 0, lx = sv ? (int)wlf - X0 : 0;
        const bool inside = ly >= 0 && lx >= 0 && ly <= TL_WH - 2 && lx <= TL_WW - 2;
        farbits |= (sv && !inside) ? (1u << t) : 0u;
        lo[t] = (inside ? ly * TL_WW + lx : 0) + h * TL_PLANE;
        const float lh = hf - hlf, lw = wf_ - wlf;
        const float m = (sv && inside) ? raw.m : 0.f;
        wq[t][0] = (1.f - lh) * (1.f - lw) * m; wq[t][1] = (1.f - lh) * lw * m;
        wq[t][2] = lh * (1.f - lw) * m;         wq[t][3] = lh * lw * m;
    }

    f32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    // Workgroup-uniform escape for large learned offsets: when samples that left the window are common, the per-lane fallback
    // below would redo most of the work on top of the LDS path -- the region is flagged instead and left to the
    // register-gather kernel that the host launches right after this one (rescue mode).
    int wave_far_taps = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) wave_far_taps += __any((farbits >> t) & 1u) ? 1 : 0;
    // cost model: a far tap adds one serial pass over all channels to ITS wave (~12 % of the wave's main loop each), the
    // rescue kernel costs ~1.7x the main loop -> hand the region over when the worst wave has five or more far taps
    __shared__ int far_worst;
    if (tid == 0) far_worst = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&far_worst, wave_far_taps);
    __syncthreads();
    if (far_worst >= rescue_taps) {
        if (tid == 0) rescue_flags[(size_t)b * gridDim.x + bx] = 1;      // dcn_fwd9_f32<2, true> computes this region
        return;
    }

    issue_w(0, lds);                       // after the early returns above: a workgroup must not end with LDS writes in flight
    commit(lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ck = 0; ck < nchunk; ++ck) {
        const float *buf = lds + (ck & 1) * TL_BUF;
#ifndef TL_ABL_NOSTAGE
        if (ck + 1 < nchunk) {
            issue_w(ck + 1, lds + ((ck + 1) & 1) * TL_BUF);      // released by the barrier that ended chunk ck - 1
            issue(ck + 1);
        }
#endif
        const float *wb_ = buf + TL_IN_FLOATS + (p * 2 + h) * 4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#ifdef TL_ABL_NOW
            const f32x4 a0 = {wq[t][0], wq[t][1], wq[t][2], wq[t][3]}, a1 = {wq[t][1], wq[t][0], wq[t][3], wq[t][2]};
            (void)wb_;
#else
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(wb_ + (t * TL_OB) * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4 *>(wb_ + (t * TL_OB + 32) * 8);
#endif
            const float *cp = buf + lo[t];
            const float q0 = wq[t][0], q1 = wq[t][1], q2 = wq[t][2], q3 = wq[t][3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float *c2 = cp + 2 * s * TL_PLANE;
#ifdef TL_ABL_NOLDS
                const float val = q0 * (float)s + q1 + q2 * (float)ck + q3;
                (void)c2;
#else
                const float val = q0 * c2[0] + q1 * c2[1] + q2 * c2[TL_WW] + q3 * c2[TL_WW + 1];
#endif
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], val, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], val, acc[1], 0, 0, 0);
            }
        }
#ifndef TL_ABL_NOSTAGE
        if (ck + 1 < nchunk) commit(lds + ((ck + 1) & 1) * TL_BUF);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the next chunk's weights have landed in LDS
#ifndef TL_ABL_NOBAR
        __syncthreads();
#endif
    }

    if (__any(farbits != 0u)) {
        // samples whose 2x2 footprint left the staged window: gathered from global memory by their lane
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            if (!__any((farbits >> t) & 1u)) continue;
            const bool mine = (farbits >> t) & 1u;
            const Tap s = make_tap(off_b, msk_b, g, t, t, ho, wo, Pc, pv);
            // weights from the coalesced rescue layout Wf9[c*9+t][Cout_pad]; loads of four channel pairs are issued together
            const float *w9 = wf9 + (size_t)t * g.Cop + z * TL_OB + p;
            const bool second = z * TL_OB + 32 < g.Cop;
            const int cend = (g.C + 1) & ~1;
            for (int c0 = h; c0 < cend; c0 += 8) {
                f32x2 gt[4], gb[4];
                float wa[4], wb2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 2 * u;
                    const int cc = c < g.C ? c : g.C - 1;
                    const float *gp = in_b + (size_t)cc * HW;
                    gt[u] = ldg2(gp, (unsigned)s.pt * 4u);
                    gb[u] = ldg2(gp, (unsigned)s.pb * 4u);
                    const float *wr = w9 + (size_t)cc * 9 * g.Cop;
                    wa[u] = wr[0];
                    wb2[u] = second ? wr[32] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 2 * u;
                    const bool live = mine && c < g.C;
                    const float val = live ? (s.a0 * gt[u].x + s.a1 * gt[u].y + s.b0 * gb[u].x + s.b1 * gb[u].y) * s.m : 0.f;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u], val, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb2[u], val, acc[1], 0, 0, 0);
                }
            }
        }
    }

    float *out_b = out + (size_t)b * g.Co * g.HoWo;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int obase = z * TL_OB + mb * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = obase + (r & 3) + 8 * (r >> 2);
            if (pv && o < g.Co) out_b[(size_t)o * g.HoWo + ho * g.Wo + wo] = acc[mb][r] + bias[o];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Atomic-free grad_input.
//
// Measured on MI355X (tools/micro/atomics.hip): global fp32 atomics retire ~21 G cache-line requests/s
// (305 G lane-atomics/s even when perfectly coalesced) and LDS fp32 atomics ~200 G/s chip-wide; the
// reference's col2im scatter needs 36 atomics per (channel, pixel) = 5.2 G per bs-8 step.  So the scatter
// is turned into a gather: every sampling position is shared by all channels of a deformable group, hence
// the INVERSE map "input cell q, tap t -> {(output pixel p, weight)}" is built once per call
// (dcn_build_inverse, a bounded window search, no atomics, deterministic order) and
//     grad_input[c,q] = sum_{t,o} W[o,t,c] * G_t[o,q],     G_t[o,q] = sum_{(p,w) in list(q,t)} w * dY[o,p]
// is evaluated exactly like the forward: the lane computes G as its MFMA B operand (dcn_bwd_input_f32).
// Samples whose offset exceeds the search radius, and cells that collect more than INV_CAP samples of one
// tap, fall back to atomics inside dcn_bwd_data_f32 -- correct for any input, fast for realistic offsets.
// ---------------------------------------------------------------------------------------------
constexpr float TL_NEAR = 3.f;   // tiled kernels: |offset| below this stays inside every staged window
// Device-side choice between "tiled kernel + far-only pass" and "generic kernel alone", taken identically by every kernel of
// a call from the offset scan's result: when more than half of the 32-pixel tiles hold a far sample (large learned offsets),
// the tiled kernels would mostly produce zeros and the far-only pass would redo nearly everything.
//
// One-pass backward (round 4): its kernel takes the far samples itself, one by one (dcn_bwd_sweep.inc: global loads and atomics
// instead of the LDS window), so what decides is the NUMBER of far samples, not the tiles they sit in: the call's zero-fill
// kernel marks far_scal[3] with FAR_BY_COUNT, the offset scan adds the far coordinates to it, and the generic kernels take over
// when more coordinates are far than far_count_limit() allows (1 in 24 ... 1 in 8 by layer width and batch; a far sample costs
// the one-pass kernel ~8 near ones and four global atomics per channel).
// The limit is the HOST's choice per call (far_count_limit below): the marker is written as FAR_BY_COUNT | (FAR_COUNT_PIVOT - limit)
// and the scan adds to it, so "more far coordinates than the limit" reads "count field above the pivot" in every kernel.
constexpr unsigned FAR_BY_COUNT = 0x80000000u;
constexpr unsigned FAR_COUNT_PIVOT = 0x40000000u;
__device__ __forceinline__ bool far_dominated(const unsigned *far_scal, int total_tiles)
{
    if (!far_scal) return false;
    const unsigned c = far_scal[3];
    if (c & FAR_BY_COUNT) return (c & ~FAR_BY_COUNT) > FAR_COUNT_PIVOT;
    return (int64_t)far_scal[1] * 2 > (int64_t)total_tiles;
}

// Far coordinates (of B * 18 * HoWo) up to which the one-pass backward keeps the call.  Measured (tools/probes/far_div_sweep.py,
// profiles/r04_far_div_b8.txt / _b1.txt): the one-pass kernel's time grows linearly with the far samples (64 -> 64 @ 96x320 x 8:
// 0.91 ms + 41 ms per unit of far fraction), the generic kernels cost 2.6 - 3.4 ms whatever the fraction, so they take over
// above ~1 in 24 coordinates; the wide-output layers' generic path is relatively dearer (1 in 16), and with few images the
// generic kernels under-fill the chip (0.65 ms against 0.18 at one image: 1 in 12 / 1 in 8).  Round 4 started with 1 in 64,
// which handed calls over at a third of the break-even density.  DCD_FAR_DIV / DCD_FAR_DIV_WIDE override (A/B timing).
inline unsigned far_count_limit(const Geom &g, bool wide)
{
    static const int div = dcd_env("DCD_FAR_DIV") ? atoi(dcd_env("DCD_FAR_DIV")) : 0;
    static const int div_wide = dcd_env("DCD_FAR_DIV_WIDE") ? atoi(dcd_env("DCD_FAR_DIV_WIDE")) : 0;
    int d = wide ? div_wide : div;
    if (d <= 0) d = wide ? (g.B >= 4 ? 16 : 8) : (g.B >= 4 ? 24 : 12);
    const int64_t total = (int64_t)g.B * 18 * g.HoWo;
    int64_t lim = total / d;
    if (lim > (int64_t)FAR_COUNT_PIVOT - 1) lim = FAR_COUNT_PIVOT - 1;
    return (unsigned)lim;
}

// ---- Who takes the far samples of a one-pass backward call: decided per LAYER on the host, from the far count of its previous call.
// The device-side hand-over above needs the generic kernels behind every call -- five launches that return at once in the
// common case (70 launches and 0.25 - 0.3 ms per train step, 2 % of the one-image step).  The one-pass kernel is correct for ANY
// share of far samples (only slower than the generic kernels beyond the limit), so the decision may lag by a call: the call's
// epilogue kernel stores its far count into a word of mapped pinned host memory (one plain store, no copy, no extra launch, works
// inside a replayed graph too), and the layer's next call reads that word on the host.  Below half the limit the call runs
// with the limit "never": no generic launches, no generic weight layouts.  Unknown (first call) or above: hand-over armed, as
// before.  A sudden change of the offset statistics costs one slow call per layer.  State is keyed by (device, weight pointer):
// the owner of a weight tensor says when that key dies -- dcd_dcn_v2_forget(weight) (dcd_amd's DCNv2 module calls it when it is
// destroyed or its parameters move) drops the entry, so another tensor the allocator later puts at the same address starts as
// "unknown" instead of inheriting the old layer's history.  A dropped entry's report word is reset and RECYCLED for the next new
// layer, never unmapped: a HIP graph captured earlier may still hold its device address (its replays then write a far count into a
// word some other layer reads -- that can cost that layer one slow or one armed call, never a wrong result); dcd_dcn_v2_policy_free
// really frees everything, for a process that replays no such graph any more.
// DCD_DCN_HANDOVER = always | never | auto (default).
struct HandoverState {
    unsigned *host = nullptr;     // far coordinates of the layer's last finished call; 0xffffffff = none yet
    unsigned *dev = nullptr;      // the same word as the device sees it
};
static std::mutex g_handover_mu;
static std::map<std::pair<int, const void *>, HandoverState> g_handover;
static std::map<int, std::vector<HandoverState>> g_handover_free;     // per device: report words of forgotten layers
constexpr unsigned HANDOVER_UNKNOWN = 0xffffffffu;

static std::atomic<int> g_handover_pin{-1};                            // dcd_dcn_v2_set_handover: >= 0 overrides the environment default
static int handover_mode()
{
    const int pin = g_handover_pin.load(std::memory_order_relaxed);
    if (pin >= 0) return pin;
    static const int m = [] {
        const char *e = dcd_env("DCD_DCN_HANDOVER");
        if (e && !strcmp(e, "always")) return 1;
        if (e && !strcmp(e, "never")) return 0;
        return 2;
    }();
    return m;
}

static bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// -> 1: device-side hand-over armed (generic launches follow the sweep), 0: the sweep keeps the call whatever the offsets are.
// *report: device address the call's epilogue stores its far count to (null: no report).
static int handover_decide(hipStream_t stream, const void *weight, unsigned real_limit, unsigned **report)
{
    *report = nullptr;
    const int mode = handover_mode();
    if (mode != 2) return mode;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_handover_mu);
    HandoverState &h = g_handover[std::make_pair(dev, weight)];
    if (!h.host) {
        std::vector<HandoverState> &fl = g_handover_free[dev];
        if (!fl.empty()) {                                                // a forgotten layer's word (already reset to "unknown")
            h = fl.back();
            fl.pop_back();
            *report = h.dev;
            return 1;
        }
        if (stream_is_capturing(stream)) return 1;                        // no allocation inside a capture
        unsigned *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc((void **)&hp, sizeof(unsigned), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&dp, hp, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
            return 1;
        }
        *hp = HANDOVER_UNKNOWN;
        h.host = hp;
        h.dev = dp;
    }
    *report = h.dev;
    const unsigned far = *(volatile unsigned *)h.host;
    return (far == HANDOVER_UNKNOWN || (uint64_t)far * 2 > real_limit) ? 1 : 0;
}

// Round 6: a far-DOMINATED call of a layer the column-buffer path also takes (dense_ok: Cin >= 256, or Cin, Cout >= 128) goes to that
// path instead of the one-pass kernel + device-side hand-over to the generic three-pass kernels: at 2 px offsets (a quarter of the
// samples far) the column-buffer backward takes 0.75 / 1.49 / 0.70 ms on 256->128 @ 24x80 / 128->128 @ 48x160 / 256->64 @ 24x80
// where the handed-over call took 1.26 / 2.00 / 0.78 (gpurun_out/route_probe.txt, tools/route_probe.sh).  Host-side like the
// policy above, from the layer's previous report; the column-buffer call reports its far count too, so the layer returns to the
// one-pass kernel when its offsets shrink.  -> true: take the column-buffer path and store the far count to *report.
static bool handover_far_dominated(const void *weight, unsigned real_limit, unsigned **report)
{
    *report = nullptr;
    if (handover_mode() != 2) return false;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_handover_mu);
    auto it = g_handover.find(std::make_pair(dev, weight));
    if (it == g_handover.end() || !it->second.host) return false;
    const unsigned far = *(volatile unsigned *)it->second.host;
    if (far == HANDOVER_UNKNOWN || far <= real_limit) return false;
    *report = it->second.dev;
    return true;
}

__global__ void dcn_far_report(const unsigned *__restrict__ scal, unsigned *__restrict__ report)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *report = scal[3] & ~FAR_BY_COUNT;
}

// Forward side of the same policy.  The tiled forward kernel takes far samples itself (per lane, from global memory); the call-wide
// far count and the rescue launch behind it only pay when far samples are common.  A layer whose last BACKWARD reported fewer far
// coordinates than 1 in 64 (half the forward's own limit) runs without both: two launches less per call.  No report yet (first
// step, inference, the dense backward of the Cout 256 layers): as before.
static bool forward_keeps_far_samples(const void *weight, int64_t ncoord)
{
    const int mode = handover_mode();
    if (mode != 2) return mode == 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_handover_mu);
    auto it = g_handover.find(std::make_pair(dev, weight));
    if (it == g_handover.end() || !it->second.host) return false;
    const unsigned far = *(volatile unsigned *)it->second.host;
    return far != HANDOVER_UNKNOWN && (int64_t)far * 64 <= ncoord;
}

constexpr int INV_CAP = 10;     // list capacity per (cell, tap): offsets below 1 px give at most 9, typically 4
constexpr int INV_RCAP = 8;     // offsets up to this many pixels are inverted; beyond -> atomic fallback
constexpr int INV_RTILE = 3;    // the tiled grad_input kernel's dY window covers lists built with a radius up to this
constexpr int INV_OVERFLOW = 255;

struct InvLists {
    const unsigned *absmax_bits;   // float bits of max |offset| over the call (device scalar)
    unsigned char *cnt;            // [B][S][HW]
    int *idx;                      // [B][S][INV_CAP][HW]
    float *w;                      // [B][S][INV_CAP][HW]
    int packed;                    // idx holds (py << 16 | px) instead of the flat pixel index (tiled grad_input kernel)
    const unsigned char *blockmax; // [B][S][ceil(Ho/8)][ceil(Wo/8)]: ceil(max |offset|) over an 8x8 block of output pixels (<= 255)
    unsigned char *tileflag;       // [B][ceil(H/4)][ceil(W/32)] or null: 1 = some cell of the 4 x 32 cell tile searched wider than
                                   // INV_RTILE, i.e. its lists may leave the tiled grad_input kernel's dY window -> that tile is
                                   // the register-gather kernel's
    int flag_tx, flag_ty;          // tiles per row / column of that table
};

__device__ __forceinline__ int inv_radius(const unsigned *absmax_bits)
{
    const float m = __uint_as_float(*absmax_bits);
    int r = (m == m) ? (int)ceilf(fminf(m, 1e6f)) : INV_RCAP;
    return r > INV_RCAP ? INV_RCAP : r;
}

// Also lists the 32-pixel tiles (image-major tile ids, the generic kernels' tiling) that hold at least one sample displaced
// by TL_NEAR px or more: the workgroup-tiled kernels leave exactly those samples to the generic kernels' far-only mode,
// which then visits the listed tiles only (usually none).  scal[0] = max bits, scal[1] = number of listed tiles.
__device__ __forceinline__ void offset_absmax_body(const float *__restrict__ off, int64_t n, unsigned *__restrict__ scal, int HoWo,
                                                   int ch_per_img, int tiles_per_img, unsigned char *__restrict__ far_flag,
                                                   int *__restrict__ far_list, int bid, int nb)
{
    float m = 0.f;
    unsigned nfar = 0u;                                            // coordinates displaced by TL_NEAR px or more -> scal[3]
    auto visit = [&](int64_t i, float raw) {
        const float v = fabsf(raw);
        m = fmaxf(m, v);
        if (!(v < TL_NEAR)) {                                      // rare
            ++nfar;
            if (!far_flag) return;                                 // one-pass backward: its kernel owns the far samples, no tile list
            const int64_t plane = i / HoWo;
            const int P = (int)(i - plane * HoWo), b = (int)(plane / ch_per_img);
            const int tid_ = b * tiles_per_img + (P >> 5);
            // plain read first: with large learned offsets 10-25 % of all coordinates land here and nearly all of them find
            // their tile already listed -- the atomics alone made this scan 0.2 ms (2 px) to 0.74 ms (4 px) on a 64-channel layer
            if (__builtin_nontemporal_load(far_flag + tid_) == 0) {
                unsigned *word = (unsigned *)(far_flag + (tid_ & ~3));
                const unsigned bit = 1u << (8 * (tid_ & 3));
                if ((atomicOr(word, bit) & bit) == 0u) far_list[atomicAdd(scal + 1, 1u)] = tid_;
            }
        }
    };
    const int64_t n4 = ((uintptr_t)off & 15) == 0 ? n >> 2 : 0;     // 16-byte loads when the tensor allows it
    for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < n4; i += (int64_t)nb * blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(off + 4 * i);
        visit(4 * i, v.x); visit(4 * i + 1, v.y); visit(4 * i + 2, v.z); visit(4 * i + 3, v.w);
    }
    for (int64_t i = 4 * n4 + (int64_t)bid * blockDim.x + threadIdx.x; i < n; i += (int64_t)nb * blockDim.x)
        visit(i, off[i]);
    // one atomic per block: thousands of same-address atomics serialise in L2 (the earlier per-wave version spent most of
    // its 40 us there)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); nfar += __shfl_xor(nfar, o); }
    __shared__ float part[4];
    __shared__ unsigned partn[4];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = m; partn[threadIdx.x >> 6] = nfar; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(scal, __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));   // non-negative floats order like their bits
        const unsigned nf = partn[0] + partn[1] + partn[2] + partn[3];
        if (nf) atomicAdd(scal + 3, nf);
    }
}
__global__ void dcn_offset_absmax(const float *__restrict__ off, int64_t n, unsigned *__restrict__ scal, int HoWo,
                                  int ch_per_img, int tiles_per_img, unsigned char *__restrict__ far_flag,
                                  int *__restrict__ far_list)
{
    offset_absmax_body(off, n, scal, HoWo, ch_per_img, tiles_per_img, far_flag, far_list, blockIdx.x, gridDim.x);
}

// ceil(max(|dh|, |dw|)) over each 8x8 block of output pixels, per (image, tap segment): lets dcn_build_inverse search a
// window as wide as the offsets NEAR a cell require instead of as wide as the call's largest offset (learned offset fields
// are smooth: the local bound is typically 1-2 px where the global one is 3-8, and the search costs (2R+1)^2 reads).
// grid = (blocks_x * blocks_y, S, B), one wave per block.
__global__ __launch_bounds__(64) void dcn_offset_blockmax(const float *__restrict__ off, Geom g, unsigned char *__restrict__ blockmax,
                                                          const unsigned *__restrict__ only_if_far = nullptr)
{
    // one-pass backward (dcn_bwd_sweep.inc): the inverse lists are needed only when the generic kernels take over the call
    if (only_if_far && !far_dominated(only_if_far, g.B * ((g.HoWo + 31) / 32))) return;
    const int nbx = (g.Wo + 7) >> 3, nby = (g.Ho + 7) >> 3;
    const int blk = blockIdx.x, seg = blockIdx.y, b = blockIdx.z;
    const int by = blk / nbx, bx = blk - by * nbx;
    const int S = g.dg * g.KK;
    const int py = by * 8 + (threadIdx.x >> 3), px = bx * 8 + (threadIdx.x & 7);
    float m = 0.f;
    if (py < g.Ho && px < g.Wo) {
        const float *oh_p = off + ((size_t)b * S + seg) * 2 * g.HoWo;
        const int P = py * g.Wo + px;
        m = fmaxf(fabsf(oh_p[P]), fabsf(oh_p[g.HoWo + P]));
        if (!(m == m)) m = 1e6f;                                   // NaN offsets: widest
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (threadIdx.x == 0) blockmax[(((size_t)b * S + seg) * nby + by) * nbx + bx] = (unsigned char)fminf(ceilf(m), 255.f);
}

// one thread per (input cell q, tap segment, image)
__global__ __launch_bounds__(256) void dcn_build_inverse(const float *__restrict__ off, const float *__restrict__ msk,
                                                         InvLists inv, Geom g, const unsigned *__restrict__ only_if_far = nullptr)
{
    if (only_if_far && !far_dominated(only_if_far, g.B * ((g.HoWo + 31) / 32))) return;
    const int HW = g.H * g.W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW) return;
    const int seg = blockIdx.y, b = blockIdx.z;
    const int S = g.dg * g.KK;
    const int t = seg % g.KK, i = t / g.kw, j = t - i * g.kw;
    const int qy = q / g.W, qx = q - qy * g.W;
    const int rg = inv_radius(inv.absmax_bits);                   // call-wide bound: samples beyond it are "far" (atomic fallback)
    // output pixels whose un-deformed tap position lies within R of this cell
    const int cy = qy + g.ph - i * g.dh, cx = qx + g.pw - j * g.dw;
    int rm1 = rg;
    if (inv.blockmax) {
        // Local bound: the 3x3 neighbourhood of 8x8 output blocks around the cell's own output position reaches >= 8 output
        // pixels (>= 8 input pixels for any stride) in every direction.  If the offsets in it are all <= r0 <= 7, only pixels
        // within r0 + 1 <= 8 can put a non-far sample next to this cell, and they all lie inside the neighbourhood.
        const int nbx = (g.Wo + 7) >> 3, nby = (g.Ho + 7) >> 3;
        const int pcy = min(max(cy / g.sh, 0), g.Ho - 1) >> 3, pcx = min(max(cx / g.sw, 0), g.Wo - 1) >> 3;
        const unsigned char *bm = inv.blockmax + ((size_t)b * S + seg) * nby * nbx;
        int r0 = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = pcy + dy, xx = pcx + dx;
                if (yy >= 0 && yy < nby && xx >= 0 && xx < nbx) r0 = max(r0, (int)bm[yy * nbx + xx]);
            }
        if (r0 <= 7 && r0 < rg) rm1 = r0;
    }
    if (inv.tileflag && rm1 > INV_RTILE) inv.tileflag[((size_t)b * inv.flag_ty + (qy >> 2)) * inv.flag_tx + (qx >> 5)] = 1;
    const int R = rm1 + 1;
    const float rlim = (float)rg;
    const float *oh_p = off + ((size_t)b * S + seg) * 2 * g.HoWo;
    const float *ow_p = oh_p + g.HoWo;
    const float *m_p = msk + ((size_t)b * S + seg) * g.HoWo;
    const size_t base = ((size_t)b * S + seg) * INV_CAP * HW + q;
    int py0 = (cy - R + g.sh - 1) / g.sh, py1 = (cy + R) / g.sh;
    int px0 = (cx - R + g.sw - 1) / g.sw, px1 = (cx + R) / g.sw;
    if (cy - R < 0) py0 = 0;
    if (cx - R < 0) px0 = 0;
    if (py1 > g.Ho - 1) py1 = g.Ho - 1;
    if (px1 > g.Wo - 1) px1 = g.Wo - 1;
    int cnt = 0;
    for (int py = py0; py <= py1; ++py)
        for (int px = px0; px <= px1; ++px) {
            const int P = py * g.Wo + px;
            const float oh = oh_p[P], ow = ow_p[P];
            if (!(fabsf(oh) <= rlim && fabsf(ow) <= rlim)) continue;      // far sample: atomic fallback owns it
            const float dh_ = ((float)(py * g.sh - g.ph + i * g.dh) + oh) - (float)qy;
            const float dw_ = ((float)(px * g.sw - g.pw + j * g.dw) + ow) - (float)qx;
            if (fabsf(dh_) < 1.f && fabsf(dw_) < 1.f) {
                if (cnt < INV_CAP) {
                    inv.idx[base + (size_t)cnt * HW] = inv.packed ? ((py << 16) | px) : P;
                    inv.w[base + (size_t)cnt * HW] = (1.f - fabsf(dh_)) * (1.f - fabsf(dw_)) * m_p[P];
                }
                ++cnt;
            }
        }
    inv.cnt[((size_t)b * S + seg) * HW + q] = (unsigned char)(cnt > INV_CAP ? INV_OVERFLOW : cnt);
    // scal[2] = number of overflowed (cell, tap) lists of the call.  Normally zero: the data kernels then skip their per-corner
    // look-ups of `cnt` altogether (four dependent byte loads per pixel and tap, each waited for before the MFMA block)
    if (cnt > INV_CAP) atomicAdd(const_cast<unsigned *>(inv.absmax_bits) + 2, 1u);
}

// grid = (ceil(in_tiles/4), B, ceil(total_channel_blocks/MB)); lane = (input cell l&31, output-channel parity l>>5)
template <int MB>
__global__ __launch_bounds__(256) void dcn_bwd_input_f32(const float *__restrict__ gy, const float *__restrict__ wb,
                                                         InvLists inv, float *__restrict__ gin, Geom g, int partner_of_tiled = 0,
                                                         const unsigned *__restrict__ only_if_far = nullptr)
{
    if (only_if_far && !far_dominated(only_if_far, g.B * ((g.HoWo + 31) / 32))) return;
    // launched next to the tiled kernel: each 4 x 32 cell tile is done by exactly one of the two, decided by the flag
    // dcn_build_inverse set for it (or, without the table, by the call-wide radius)
    if (partner_of_tiled && !inv.tileflag && inv_radius(inv.absmax_bits) <= INV_RTILE) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int HW = g.H * g.W;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int tile = bx * 4 + wave;
    if (tile * 32 >= HW) return;
    const int S = g.dg * g.KK;
    const int nblk = g.cpgp / 32;                 // channel blocks per deformable group
    const int gb0 = blockIdx.z * MB;              // first global channel block of this wave
    const int Q = tile * 32 + p;
    bool qv = Q < HW;
    const int Qc = qv ? Q : HW - 1;
    if (partner_of_tiled && inv.tileflag) {
        const int qy_ = Qc / g.W, qx_ = Qc - qy_ * g.W;
        qv = qv && inv.tileflag[((size_t)b * inv.flag_ty + (qy_ >> 2)) * inv.flag_tx + (qx_ >> 5)] != 0;
        if (!__any(qv)) return;                   // wave-uniform: none of these 32 cells lies in a flagged tile
    }
    const int nsteps = g.Cop / 2;
    const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    for (int seg = 0; seg < S; ++seg) {
        const int grp = seg / g.KK;
        // skip taps of groups this wave holds no channels of (wave-uniform)
        if ((gb0 + MB - 1) / nblk < grp || gb0 / nblk > grp) continue;
        int cnt = qv ? (int)inv.cnt[((size_t)b * S + seg) * HW + Qc] : 0;
        if (cnt == INV_OVERFLOW) cnt = 0;
        int maxc = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
        if (maxc == 0) continue;
        unsigned eoff[INV_CAP];
        float ew[INV_CAP];
        const size_t base = ((size_t)b * S + seg) * INV_CAP * HW + Qc;
        const int mu = __builtin_amdgcn_readfirstlane(maxc);      // wave-uniform: the loads below sit behind SCALAR branches
#pragma unroll
        for (int e = 0; e < INV_CAP; ++e) {
            int ri = 0;
            float rw_ = 0.f;
            if (e < mu) {                                          // slots past this cell's cnt hold stale words: masked below
                ri = inv.idx[base + (size_t)e * HW];
                rw_ = inv.w[base + (size_t)e * HW];
            }
            if (inv.packed) ri = (ri >> 16) * g.Wo + (ri & 0xffff);     // (row, column) as the tiled kernel wants them -> flat pixel
            eoff[e] = e < cnt ? (unsigned)ri * 4u : 0u;
            ew[e] = e < cnt ? rw_ : 0.f;
        }
        const unsigned wlane = ((unsigned)h * (unsigned)g.Kp + (unsigned)(seg * g.cpgp) + (unsigned)p) * 4u;
        const unsigned Kp8 = (unsigned)g.Kp * 8u;           // two weight rows per step
        const size_t row2 = (size_t)2 * g.HoWo;              // two dY rows per step
        const int last_o = g.Co - 1;
        // stage = the raw dY samples and weight operands of ONE output-channel pair; two stages alternate so that the
        // next pair's loads are in flight while the current pair runs on the matrix pipe
        float ga[INV_CAP], gb_[INV_CAP], wa[MB], wbx[MB];
        // NE: entries gathered per cell, a compile-time count picked from the wave-uniform maximum; unused slots read element
        // 0 of the row with weight 0 (no per-entry branches: those serialised the gathers)
        auto run = [&](auto ne_tag) {
            constexpr int NE = decltype(ne_tag)::value;
            auto issue = [&](float (&gv)[INV_CAP], float (&wv)[MB], int st) {
                const int o = 2 * st + h;
                const float *gp = gy_b + (size_t)(o < g.Co ? o : last_o) * g.HoWo;
#pragma unroll
                for (int e = 0; e < NE; ++e)
#ifdef K3_ABL_NOGATHER
                    gv[e] = ew[e] + (float)st;
#else
                    gv[e] = ldg(gp, eoff[e]);
#endif
                const float *wp = (const float *)((const char *)wb + (size_t)st * Kp8);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int gbk = gb0 + mb;
                    if (gbk / nblk == grp) wv[mb] = ldg(wp, wlane + (unsigned)((gbk - grp * nblk) * 32) * 4u);
                }
            };
            auto compute = [&](float (&gv)[INV_CAP], float (&wv)[MB], int st) {
                float val = 0.f;
#pragma unroll
                for (int e = 0; e < NE; ++e) val += ew[e] * gv[e];
                val = (2 * st + h < g.Co) ? val : 0.f;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    if ((gb0 + mb) / nblk == grp) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[mb], val, acc[mb], 0, 0, 0);
            };
            issue(ga, wa, 0);
            int st = 0;
            for (; st + 1 < nsteps; st += 2) {
                issue(gb_, wbx, st + 1);
                compute(ga, wa, st);
                if (st + 2 < nsteps) issue(ga, wa, st + 2);
                compute(gb_, wbx, st + 1);
            }
            if (st < nsteps) compute(ga, wa, st);
        };
        (void)row2;
        if (mu <= 2) run(IntC<2>{});
        else if (mu <= 4) run(IntC<4>{});
        else if (mu <= 6) run(IntC<6>{});
        else if (mu <= 8) run(IntC<8>{});
        else run(IntC<INV_CAP>{});
    }

    float *gin_b = gin + (size_t)b * g.C * HW;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int gb = gb0 + mb;
        const int grp = gb / nblk, blk = gb - grp * nblk;
        if (grp >= g.dg) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cc = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (qv && cc < g.cpg) gin_b[((size_t)grp * g.cpg + cc) * HW + Q] = acc[mb][r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// grad_input from the inverse lists, workgroup-tiled (3x3 / stride 1 / pad 1 / dil 1 / one group).
// dcn_bwd_input_f32 spends its time on ~5 jittered global gathers of dY per (tap, output-channel pair) -- the 16-clk
// texture-addresser path again.  Every listed pixel lies within 5 px of its cell (|offset| <= 3 by construction of the
// lists), so a workgroup that owns a 4 x 32 cell tile stages the 14 x 48 dY window of 16 output channels at a time in LDS
// and the gathers become ds_read_b32.  Loop order: channel chunk (barrier) -> tap (list entries re-read, coalesced) ->
// 8 output-channel pairs.  No fallback path is needed: what the lists do not hold is added by the data kernels' atomics.
// grid = (cell tiles, B, ceil(Cin/32/MB)); block = 256 (wave = tile row).
// ---------------------------------------------------------------------------------------------
constexpr int BI_TR = 4;
constexpr int BI_WH = BI_TR + 10;                     // 14 window rows  r0-5 .. r0+8
constexpr int BI_WW = 48;                             // window cols  c0-8 .. c0+39
constexpr int BI_PLANE = BI_WH * BI_WW;               // 672 = 10*64 + 32: the two lane halves (o, o+1) are 32 banks apart
constexpr int BI_OC = 16;                             // output channels per chunk

template <int MB>
__global__ __launch_bounds__(BI_TR * 64) void dcn_bwd_input_tile_f32(const float *__restrict__ gy, const float *__restrict__ wb,
                                                                    InvLists inv, float *__restrict__ gin, Geom g, int tiles_x)
{
    static_assert(BI_TR == 4, "dcn_build_inverse flags 4 x 32 cell tiles");
    if (!inv.tileflag && inv_radius(inv.absmax_bits) > INV_RTILE) return;   // listed pixels may lie outside the window
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [BI_OC][BI_PLANE] window | [8 pairs][9][MB][2][32] weights
    float *wsl = lds + BI_OC * BI_PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    // some cell of this tile searched wider than the window allows: the register-gather kernel (launched next) owns the tile
    if (inv.tileflag && inv.tileflag[((size_t)b * inv.flag_ty + ty) * inv.flag_tx + tx]) return;
    const int r0 = ty * BI_TR, c0 = tx * 32;
    const int qy = r0 + wave, qx = c0 + p;
    const bool qv = qy < g.H && qx < g.W;
    const int HW = g.H * g.W;
    const int Qc = qv ? qy * g.W + qx : 0;
    const int Y0 = r0 - 5, X0 = c0 - 8;
    const int gb0 = blockIdx.z * MB;
    const int nchunk = g.Cop / BI_OC;
    const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    constexpr int NIN = BI_OC * BI_WH * (BI_WW / 4);          // 2688 dwordx4 per chunk
    constexpr int KIN = (NIN + BI_TR * 64 - 1) / (BI_TR * 64);   // 11 (last partial)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // list entries (raw, undecoded) of the next tap to run: cnt is capped at 6 loads in flight per array to bound registers
    // The two lane halves serve the same cell, so each loads HALF of the list (half 0 the even slots, half 1 the odd ones) and
    // one v_permlane32_swap per register hands both halves the pair: 10 loads per lane and tap instead of 20 (the list loads
    // -- duplicate addresses in the two halves, i.e. the slow address path -- were the largest piece of the kernel's non-MFMA time)
    static_assert(INV_CAP % 2 == 0, "list slots are loaded in (even, odd) pairs");
    int ncnt = 0, neo[INV_CAP / 2];
    float new_[INV_CAP / 2];
    auto fetch_list = [&](int t) {
        int c = qv ? (int)inv.cnt[((size_t)b * 9 + t) * HW + Qc] : 0;
        ncnt = c == INV_OVERFLOW ? 0 : c;
        const size_t base = ((size_t)b * 9 + t) * INV_CAP * HW + Qc;
#pragma unroll
        for (int e2 = 0; e2 < INV_CAP / 2; ++e2) {       // unconditional: slots past cnt hold stale data of the workspace, masked on use
            neo[e2] = inv.idx[base + (size_t)(2 * e2 + h) * HW];
            new_[e2] = inv.w[base + (size_t)(2 * e2 + h) * HW];
        }
    };
    fetch_list(0);

    for (int ck = 0; ck < nchunk; ++ck) {
        const float *src = gy_b + (size_t)ck * BI_OC * g.HoWo;
        const int oleft = g.Co - ck * BI_OC;
        __syncthreads();                                       // previous chunk fully consumed
#ifdef BIT_ABL_NOSTAGE
        if (ck < 0)
#endif
#pragma unroll 1
        for (int k0 = 0; k0 < KIN; k0 += 4) {
            f32x4 rin[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + BI_TR * 64 * (k0 + k);
                const int o = e / (BI_WH * 12), rm = e - o * (BI_WH * 12);
                const int wr = rm / 12, q = rm - wr * 12;
                const int y = Y0 + wr, x = X0 + 4 * q;
                rin[k] = zero4;
                if (e < NIN && o < oleft && y >= 0 && y < g.Ho && x >= 0 && x < g.Wo)
                    rin[k] = *reinterpret_cast<const f32x4 *>(src + (size_t)o * g.HoWo + y * g.Wo + x);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = tid + BI_TR * 64 * (k0 + k);
                if (e < NIN) *reinterpret_cast<f32x4 *>(lds + e * 4) = rin[k];      // [o][row][col] is exactly item order
            }
        }
        {
            // A operands of this chunk: Wb rows of its 16 output channels, the 64 channels of this workgroup, all 9 taps.
            // Slab index = ((pair*9 + tap)*MB + mb)*2 + h groups of 32 floats: the two lane halves sit 32 banks apart.
            constexpr int NWQ = (BI_OC / 2) * 9 * MB * 2 * 8;                 // dwordx4 items
            f32x4 rw[(NWQ + BI_TR * 64 - 1) / (BI_TR * 64)];
#pragma unroll
            for (int k = 0; k < (NWQ + BI_TR * 64 - 1) / (BI_TR * 64); ++k) {
                const int e = tid + BI_TR * 64 * k;
                const int gq = e >> 3, q = e & 7;
                const int hh = gq & 1, mbq = (gq >> 1) % MB, st = (gq >> 1) / MB;
                const int t = st % 9, sp = st / 9;
                rw[k] = zero4;
                if (e < NWQ && (gb0 + mbq) * 32 < g.cpgp)
                    rw[k] = *reinterpret_cast<const f32x4 *>(wb + (size_t)(ck * BI_OC + 2 * sp + hh) * g.Kp + (size_t)t * g.cpgp +
                                                             (gb0 + mbq) * 32 + 4 * q);
            }
#pragma unroll
            for (int k = 0; k < (NWQ + BI_TR * 64 - 1) / (BI_TR * 64); ++k) {
                const int e = tid + BI_TR * 64 * k;
                if (e < NWQ) *reinterpret_cast<f32x4 *>(wsl + e * 4) = rw[k];
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            // entries of this tap were fetched while the previous tap ran; request the next tap's now
            int cnt = ncnt;
            int eo[INV_CAP];
            float ew[INV_CAP];
#pragma unroll
            for (int e2 = 0; e2 < INV_CAP / 2; ++e2) {
                // V_PERMLANE32_SWAP: lanes 32-63 of the first operand <-> lanes 0-31 of the second: [0] = half 0's value in
                // every lane (the even slot), [1] = half 1's (the odd slot)
                const auto pi = __builtin_amdgcn_permlane32_swap((unsigned)neo[e2], (unsigned)neo[e2], false, false);
                const auto pw = __builtin_amdgcn_permlane32_swap(__float_as_uint(new_[e2]), __float_as_uint(new_[e2]), false, false);
                eo[2 * e2] = (int)pi[0]; eo[2 * e2 + 1] = (int)pi[1];
                ew[2 * e2] = __uint_as_float(pw[0]); ew[2 * e2 + 1] = __uint_as_float(pw[1]);
            }
            fetch_list(t == 8 ? 0 : t + 1);
            int maxc = cnt;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
            if (maxc == 0) continue;
#pragma unroll
            for (int e = 0; e < INV_CAP; ++e) {
                const int P = eo[e];                                   // packed (py << 16 | px), 0 weight when unused
                eo[e] = e < cnt ? ((P >> 16) - Y0) * BI_WW + ((P & 0xffff) - X0) + h * BI_PLANE : 0;
                ew[e] = e < cnt ? ew[e] : 0.f;
            }
            const float *wrow = wsl + (t * MB * 2 + h) * 32 + p;
            // NE = entries evaluated per cell: a compile-time count picked from the wave-uniform maximum, every read
            // unconditional (unused slots point at window cell 0 with weight 0).  With a run-time bound the compiler guarded each
            // ds_read with its own exec branch and waited for it at once -- ten serialised LDS latencies per output pair.
            auto run = [&](auto ne_tag) {
                constexpr int NE = decltype(ne_tag)::value;
#pragma unroll
                for (int s = 0; s < BI_OC / 2; ++s) {
                    const float *pl = lds + 2 * s * BI_PLANE;
                    float val = 0.f;
#pragma unroll
                    for (int e = 0; e < NE; ++e)
#ifdef BIT_ABL_NOLDS
                        val += ew[e] * (float)eo[e];
#else
                        val += ew[e] * pl[eo[e]];
#endif
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
#ifdef BIT_ABL_NOMFMA
                        acc[mb][s] += val;
#else
                        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[(s * 9 * MB + mb) * 64], val, acc[mb], 0, 0, 0);
#endif
                }
            };
            const int mu = __builtin_amdgcn_readfirstlane(maxc);
            if (mu <= 2) run(IntC<2>{});
            else if (mu <= 4) run(IntC<4>{});
            else if (mu <= 6) run(IntC<6>{});
            else if (mu <= 8) run(IntC<8>{});
            else run(IntC<INV_CAP>{});
        }
    }

    float *gin_b = gin + (size_t)b * g.C * HW;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int blk = gb0 + mb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cc = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (qv && cc < g.C) gin_b[(size_t)cc * HW + Qc] = acc[mb][r];
        }
    }
}

// grad_bias[o] = sum_{b,p} dY[b,o,p].  grid = (Cout, splits); one float atomic per block (same-address atomics from
// every wave of the data kernel serialised in L2 and cost more than the whole MFMA work).
// (o, sp, nsp: the block's output channel, its slice and the number of slices)
__device__ __forceinline__ void bias_grad_body(const float *__restrict__ gy, float *__restrict__ gbias, int B, int Co, int HoWo,
                                               const unsigned *__restrict__ only_if_far, int o, int sp, int nsp)
{
    // one-pass backward: its chunk-0 waves sum dY themselves (dcn_sweep_reduce_dw adds the partials); this kernel then runs only
    // when the generic kernels take the call over
    if (only_if_far && !far_dominated(only_if_far, B * ((HoWo + 31) / 32))) return;
    // block (o, split): split s walks its slice of every image's plane with 16-byte loads, no per-element index division
    const int nq = HoWo >> 2;                                  // float4 per plane (planes are 16-byte aligned when HoWo % 4 == 0)
    const bool vec = (HoWo & 3) == 0;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *p = gy + ((size_t)b * Co + o) * HoWo;
        if (vec) {
            for (int i = sp * 256 + threadIdx.x; i < nq; i += nsp * 256) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 4 * i);
                acc += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int i = sp * 256 + threadIdx.x; i < HoWo; i += nsp * 256) acc += p[i];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(gbias + o, part[0] + part[1] + part[2] + part[3]);
}
__global__ __launch_bounds__(256) void dcn_bias_grad(const float *__restrict__ gy, float *__restrict__ gbias, int B, int Co, int HoWo,
                                                     const unsigned *__restrict__ only_if_far = nullptr)
{
    bias_grad_body(gy, gbias, B, Co, HoWo, only_if_far, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---------------------------------------------------------------------------------------------
// Backward w.r.t. input, offset, mask (grad_bias: dcn_bias_grad).
// grid = (ceil(tiles/4), B, nsplit); each z handles a contiguous range of 32-channel blocks.
// NS = Cop/2 register-cached dY values per lane (0: stream dY from memory, any Cout).
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void dcn_bwd_data_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                        const float *__restrict__ msk, const float *__restrict__ wb,
                                                        const float *__restrict__ gy, float *__restrict__ gin,
                                                        float *__restrict__ goff, float *__restrict__ gmsk,
                                                        float *__restrict__ gbias, Geom g, int nsplit, InvLists inv,
                                                        const unsigned *__restrict__ far_scal, const int *__restrict__ far_list,
                                                        int no_lists = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    int tile = bx * 4 + wave;
    // far-only mode (far_scal != nullptr): the tiled kernel produced grad_offset / grad_mask of every sample displaced by
    // less than TL_NEAR; this launch visits the listed tiles and OVERWRITES the entries of the remaining samples only.
    const bool far_mode = far_scal != nullptr && !far_dominated(far_scal, g.B * ((g.HoWo + 31) / 32));   // else: everything
    int tap_only = -1;
    if (far_mode) {
        const int tiles_per_img = (g.HoWo + 31) / 32;
        int L = (b * (int)gridDim.x + bx) * 4 + wave;
        // one-pass backward: a listed tile's taps go to different waves when the launch has the waves for it.  A listed tile was
        // ONE wave's serial walk over 9 taps x all channel blocks -- 100-230 us per layer in the train step for a handful of far
        // samples (profiles/step_r03_v1_sequence.txt), whatever the size of the rest of the launch.
        const int nfar = (int)far_scal[1], waves = (int)(gridDim.x * gridDim.y) * 4;
        if (no_lists && (int64_t)nfar * g.KK <= (int64_t)waves) {
            tap_only = L % g.KK;
            L /= g.KK;
        }
        if (L >= nfar) return;
        const int tid_ = far_list[L];
        b = tid_ / tiles_per_img;
        tile = tid_ - b * tiles_per_img;
    }
    if (tile * 32 >= g.HoWo) return;
    const int z = blockIdx.z;
    const float rlim = (float)inv_radius(inv.absmax_bits);
    const bool any_overflow = inv.absmax_bits[2] != 0u;
    const int S_all = g.dg * g.KK;
    const int P = tile * 32 + p;
    const bool pv = P < g.HoWo;
    const int Pc = pv ? P : g.HoWo - 1;
    const int ho = Pc / g.Wo, wo = Pc - ho * g.Wo;
    const int HW = g.H * g.W;
    const int nsteps = g.Cop / 2;

    const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;
    float dy[NS > 0 ? NS : 1];
    if (NS > 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int o = 2 * s + h;
            dy[s] = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
        }
    }

    const float *in_b = in + (size_t)b * g.C * HW;
    float *gin_b = gin + (size_t)b * g.C * HW;
    const float *off_b = off + (size_t)b * g.dg * 2 * g.KK * g.HoWo;
    const float *msk_b = msk + (size_t)b * g.dg * g.KK * g.HoWo;
    float *goff_b = goff + (size_t)b * g.dg * 2 * g.KK * g.HoWo;
    float *gmsk_b = gmsk + (size_t)b * g.dg * g.KK * g.HoWo;

    const int nblk = g.cpgp / 32;
    const int blk0 = (int)((int64_t)z * nblk / nsplit), blk1 = (int)((int64_t)(z + 1) * nblk / nsplit);

    for (int grp = 0; grp < g.dg; ++grp) {
        const float *in_g = in_b + (size_t)grp * g.cpg * HW;
        float *gin_g = gin_b + (size_t)grp * g.cpg * HW;
        for (int t = 0; t < g.KK; ++t) {
            if (tap_only >= 0 && t != tap_only) continue;
            const int seg = grp * g.KK + t;
            const Tap s = make_tap(off_b, msk_b, g, seg, t, ho, wo, Pc, pv);
            float s_m = 0.f, s_h = 0.f, s_w = 0.f;
            // grad_input normally comes from dcn_bwd_input_f32; this kernel scatters only what the inverse lists do
            // not cover: samples farther than the search radius, and corners that fell into an overflowed cell.
            // no_lists (one-pass backward, dcn_bwd_sweep.inc): in far-only mode there are no inverse lists at all, every sample
            // this launch owns scatters its grad_input here
            const bool far = (far_mode && no_lists) || !(fabsf(s.oh) <= rlim && fabsf(s.ow) <= rlim);
            const bool mine = !far_mode || !(fabsf(s.oh) < TL_NEAR && fabsf(s.ow) < TL_NEAR);   // sample owned by this launch
            if (far_mode && !__any(mine)) continue;
            const unsigned char *cnt_p = inv.cnt + ((size_t)b * S_all + seg) * HW;
            bool o1 = false, o2 = false, o3 = false, o4 = false;
            if (any_overflow) {                                  // uniform, normally false: no look-ups at all
                o1 = cnt_p[s.i1] == INV_OVERFLOW; o2 = cnt_p[s.i2] == INV_OVERFLOW;
                o3 = cnt_p[s.i3] == INV_OVERFLOW; o4 = cnt_p[s.i4] == INV_OVERFLOW;
            }
            const bool a1 = mine && s.c1 && (far || o1), a2 = mine && s.c2 && (far || o2);
            const bool a3 = mine && s.c3 && (far || o3), a4 = mine && s.c4 && (far || o4);
            for (int blk = blk0; blk < blk1; ++blk) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float *wp = wb + (size_t)seg * g.cpgp + blk * 32 + p + (size_t)h * g.Kp;
                if (NS > 0) {
#pragma unroll
                    for (int k = 0; k < NS; ++k)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[(size_t)(2 * k) * g.Kp], dy[k], acc, 0, 0, 0);
                } else {
                    for (int k = 0; k < nsteps; ++k) {
                        const int o = 2 * k + h;
                        const float d = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wp[(size_t)(2 * k) * g.Kp], d, acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    // rows r = 4*r4 .. 4*r4+3 are 4 consecutive channels: issue their 16 gathers, then do the arithmetic
                    const int cc0 = blk * 32 + 8 * r4 + 4 * h;
                    float v[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cc = cc0 + u;
                        const float *ip = in_g + (size_t)(cc < g.cpg ? cc : 0) * HW;
                        v[u][0] = ip[s.i1]; v[u][1] = ip[s.i2]; v[u][2] = ip[s.i3]; v[u][3] = ip[s.i4];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cc = cc0 + u;
                        if (cc < g.cpg) {
                            const float v1 = s.c1 ? v[u][0] : 0.f, v2 = s.c2 ? v[u][1] : 0.f;
                            const float v3 = s.c3 ? v[u][2] : 0.f, v4 = s.c4 ? v[u][3] : 0.f;
                            const float d = acc[4 * r4 + u];
                            s_m += d * (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4);
                            const float dm = d * s.m;
                            // d/dh and d/dw of the bilinear sample (cuda/dcn_v2_im2col_cuda.cu:82-123)
                            s_h += dm * (s.hw * (v3 - v1) + s.lw * (v4 - v2));
                            s_w += dm * (s.hh * (v2 - v1) + s.lh * (v4 - v3));
                        }
                    }
                }
                if (__any(a1 | a2 | a3 | a4)) {     // rare: far samples / overflowed cells of this tap (wave-uniform test)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cc = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (cc < g.cpg) {
                            const float dm = acc[r] * s.m;
                            float *gp = gin_g + (size_t)cc * HW;
                            if (a1) atomicAdd(gp + s.i1, dm * s.w1);
                            if (a2) atomicAdd(gp + s.i2, dm * s.w2);
                            if (a3) atomicAdd(gp + s.i3, dm * s.w3);
                            if (a4) atomicAdd(gp + s.i4, dm * s.w4);
                        }
                    }
                }
            }
            s_m += __shfl_xor(s_m, 32);
            s_h += __shfl_xor(s_h, 32);
            s_w += __shfl_xor(s_w, 32);
            if (h == 0 && pv && mine) {
                if (nsplit == 1) {
                    goff_b[(size_t)(2 * seg) * g.HoWo + P] = s_h;
                    goff_b[(size_t)(2 * seg + 1) * g.HoWo + P] = s_w;
                    gmsk_b[(size_t)seg * g.HoWo + P] = s_m;
                } else {
                    atomicAdd(goff_b + (size_t)(2 * seg) * g.HoWo + P, s_h);
                    atomicAdd(goff_b + (size_t)(2 * seg + 1) * g.HoWo + P, s_w);
                    atomicAdd(gmsk_b + (size_t)seg * g.HoWo + P, s_m);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward w.r.t. offset / mask, workgroup-tiled (3x3 / stride 1 / pad 1 / dil 1 / one group).
// Same contraction as dcn_bwd_data_f32 (dcol = W^T dY on the matrix pipe, dY held in registers), but the 4 x 16 bilinear
// corner values each lane needs per (tap, 32-channel block) come from the LDS window of its workgroup's 4 x 32-pixel tile
// (two ds_read2_b32 per channel) instead of 64 jittered global gathers.  Near samples only (|offset| < TL_NEAR): the generic
// kernel's far-only mode overwrites grad_offset / grad_mask of the far ones in the listed tiles.  The rare grad_input
// atomics for cells whose inverse list overflowed are issued from here exactly as in the generic kernel.
// grid = (tiles, B, nsplit); block = 256 (wave = tile row).
// ---------------------------------------------------------------------------------------------
constexpr int BD_TR = 4;
constexpr int BD_WH = BD_TR + 8;                      // 12 window rows
constexpr int BD_PLANE = BD_WH * TL_WW + 8;           // 488: 4 planes apart (the two lane halves) = 32 banks apart
constexpr int BD_CB = 32;

template <int NS>
__global__ __launch_bounds__(BD_TR * 64, 2) void dcn_bwd_data_tile_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                                   const float *__restrict__ msk, const float *__restrict__ wb,
                                                                   const float *__restrict__ gy, float *__restrict__ gin,
                                                                   float *__restrict__ goff, float *__restrict__ gmsk, Geom g,
                                                                   int tiles_x, int nsplit, InvLists inv,
                                                                   const unsigned *__restrict__ far_scal)
{
    if (far_dominated(far_scal, g.B * ((g.HoWo + 31) / 32))) return;      // the generic kernel does the whole job instead
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [32][BD_PLANE] window | 2 x [2 NS][32] weights of a tap
    float *wsl = lds + BD_CB * BD_PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    int bx = blockIdx.x, b = blockIdx.y;
    xcd_remap(bx, b);
    const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int r0 = ty * BD_TR, c0 = tx * 32;
    const int ho = r0 + wave, wo = c0 + p;
    const bool pv = ho < g.Ho && wo < g.Wo;
    const int P = pv ? ho * g.Wo + wo : 0;
    const int Y0 = r0 - 4, X0 = c0 - 4;
    const int HW = g.H * g.W;
    const int z = blockIdx.z;

    const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;
    float dy[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int o = 2 * s + h;
        dy[s] = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
    }
    const float *in_b = in + (size_t)b * g.C * HW;
    float *gin_b = gin + (size_t)b * g.C * HW;
    const float *off_b = off + (size_t)b * 18 * g.HoWo;
    const float *msk_b = msk + (size_t)b * 9 * g.HoWo;
    float *goff_b = goff + (size_t)b * 18 * g.HoWo;
    float *gmsk_b = gmsk + (size_t)b * 9 * g.HoWo;

    // one 32-channel block per workgroup (grid.z = blocks): the tap loop stays rolled (bounded registers) and each tap's
    // sums leave through atomics on the zero-filled gradients when there is more than one block
    const int blk0 = z, blk1 = z + 1;

    constexpr int NIN = BD_CB * BD_WH * 10;                   // window dwordx4 per block (3840)
    constexpr int KIN = NIN / (BD_TR * 64);                   // 15 per thread
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int blk = blk0; blk < blk1; ++blk) {
        const float *in_c = in_b + (size_t)blk * BD_CB * HW;
        const int cleft = g.cpg - blk * BD_CB;
        __syncthreads();                                       // previous block fully consumed
#ifdef BDT_ABL_NOSTAGE
        if (blk < 0)
#endif
#pragma unroll 1
        for (int k0 = 0; k0 < KIN; k0 += 5) {
            f32x4 rin[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int e = tid + BD_TR * 64 * (k0 + k);
                const int ch = e / (BD_WH * 10), rm = e - ch * (BD_WH * 10);
                const int wr = rm / 10, q = rm - wr * 10;
                const int y = Y0 + wr, x = X0 + 4 * q;
                rin[k] = zero4;
                if (ch < cleft && y >= 0 && y < g.H && x >= 0 && x < g.W)
                    rin[k] = *reinterpret_cast<const f32x4 *>(in_c + (size_t)ch * HW + y * g.W + x);
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int e = tid + BD_TR * 64 * (k0 + k);
                const int ch = e / (BD_WH * 10), rm = e - ch * (BD_WH * 10);
                *reinterpret_cast<f32x4 *>(lds + ch * BD_PLANE + rm * 4) = rin[k];
            }
        }
        __syncthreads();

        // A operands (weights of one tap: [2 NS outputs][32 channels] = NS x 256 B) travel through LDS, double-buffered per tap:
        // fetched as coalesced dwordx4 (2 NS / 64 per thread) instead of NS dword loads per LANE and tap -- with eight waves per
        // CU those loads (two 128-byte runs each, the slow address path) kept the CU's vector-memory unit busy for ~4 k cycles
        // per tap, twice the MFMA block they feed (ablation: r02_abl_64.txt).
        constexpr int NWQ = (2 * NS * 8 + BD_TR * 64 - 1) / (BD_TR * 64);          // dwordx4 per thread and tap
        f32x4 wreg[NWQ];
        auto w_issue = [&](int t) {
#pragma unroll
            for (int k = 0; k < NWQ; ++k) {
                const int e = tid + BD_TR * 64 * k;
                const int o = e >> 3, q = e & 7;
                if (e < 2 * NS * 8) wreg[k] = *reinterpret_cast<const f32x4 *>(wb + (size_t)o * g.Kp + (size_t)t * g.cpgp + blk * 32 + 4 * q);
            }
        };
        auto w_commit = [&](int buf) {
#pragma unroll
            for (int k = 0; k < NWQ; ++k) {
                const int e = tid + BD_TR * 64 * k;
                if (e < 2 * NS * 8) *reinterpret_cast<f32x4 *>(wsl + buf * (2 * NS * 32) + e * 4) = wreg[k];
            }
        };
        w_issue(0);
        w_commit(0);
        __syncthreads();
        const bool any_overflow = far_scal[2] != 0u;           // uniform, normally false
        TapRaw raw_next = load_tap_raw(off_b, msk_b, g, 0, P);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            // sampling state of (pixel, tap): recomputed per channel block (3 loads + ~40 VALU against 2*NS MFMAs); the raw
            // offsets / mask of the NEXT tap are requested here so their latency hides behind this tap's MFMA block
            const TapRaw raw = raw_next;
            if (t + 1 < 9) raw_next = load_tap_raw(off_b, msk_b, g, t + 1, P);
            const int ky = t / 3, kx = t - ky * 3;
            const float hf = (float)(ho - 1 + ky) + raw.oh, wf_ = (float)(wo - 1 + kx) + raw.ow;
            const bool sv = pv && hf > -1.f && wf_ > -1.f && hf < (float)g.H && wf_ < (float)g.W &&
                            fabsf(raw.oh) < TL_NEAR && fabsf(raw.ow) < TL_NEAR;
            const float hlf = floorf(hf), wlf = floorf(wf_);
            const int hl = (int)hlf, wl = (int)wlf;
            const int pos = sv ? (hl - Y0) * TL_WW + (wl - X0) : 0;
            const float lh = sv ? hf - hlf : 0.f, lw = sv ? wf_ - wlf : 0.f;
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float m = sv ? raw.m : 0.f;
            const float w1 = sv ? hh * hw : 0.f, w2 = sv ? hh * lw : 0.f, w3 = sv ? lh * hw : 0.f, w4 = sv ? lh * lw : 0.f;
            // corners whose cell list overflowed get their grad_input by atomics here (the list kernel skips them)
            const bool c1 = sv && hl >= 0 && wl >= 0, c2 = sv && hl >= 0 && wl + 1 <= g.W - 1;
            const bool c3 = sv && hl + 1 <= g.H - 1 && wl >= 0, c4 = sv && hl + 1 <= g.H - 1 && wl + 1 <= g.W - 1;
            const int i1 = c1 ? hl * g.W + wl : 0, i2 = c2 ? hl * g.W + wl + 1 : 0;
            const int i3 = c3 ? (hl + 1) * g.W + wl : 0, i4 = c4 ? (hl + 1) * g.W + wl + 1 : 0;
            const unsigned char *cnt_p = inv.cnt + ((size_t)b * 9 + t) * HW;
#ifdef BDT_ABL_NOCNT
            const bool a1 = false, a2 = false, a3 = false, a4 = false; (void)cnt_p;
#else
            bool a1 = false, a2 = false, a3 = false, a4 = false;
            if (any_overflow) {                                  // rare: some (cell, tap) list of this call overflowed
                a1 = c1 && cnt_p[i1] == INV_OVERFLOW; a2 = c2 && cnt_p[i2] == INV_OVERFLOW;
                a3 = c3 && cnt_p[i3] == INV_OVERFLOW; a4 = c4 && cnt_p[i4] == INV_OVERFLOW;
            }
#endif

            // dcol block on the matrix pipe: the A operands (weights of this tap) were fetched during the previous tap, the
            // next tap's are requested now; two accumulator chains keep dependent MFMAs from serialising the wave
            if (t + 1 < 9) w_issue(t + 1);
            const float *wa = wsl + (t & 1) * (2 * NS * 32) + h * 32 + p;      // A of k-step k: wa[k * 64] = Wb[2k + h][tap, channel p]
            f32x16 acc, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#ifndef BDT_ABL_NOMFMA
#pragma unroll
            for (int k = 0; k < NS; k += 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[k * 64], dy[k], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[(k + 1) * 64], dy[k + 1], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
#else
            acc[0] = wa[0] + dy[t]; acc[5] = dy[t + 1];
#endif

            const float *cp0 = lds + pos + 4 * h * BD_PLANE;
            float s_m = 0.f, s_h = 0.f, s_w = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *cp = cp0 + ((r & 3) + 8 * (r >> 2)) * BD_PLANE;
#ifdef BDT_ABL_NOCORNER
                const float v1 = w1 + (float)r, v2 = w2, v3 = w3, v4 = w4; (void)cp;
#else
                const float v1 = cp[0], v2 = cp[1], v3 = cp[TL_WW], v4 = cp[TL_WW + 1];
#endif
                const float d = acc[r];
                s_m += d * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                const float dm = d * m;
                s_h += dm * (hw * (v3 - v1) + lw * (v4 - v2));
                s_w += dm * (hh * (v2 - v1) + lh * (v4 - v3));
            }
            s_m += __shfl_xor(s_m, 32);
            s_h += __shfl_xor(s_h, 32);
            s_w += __shfl_xor(s_w, 32);
#ifdef BDT_ABL_NOATOM
            if (h == 0 && pv && s_m == 123.456f) {
#else
            if (h == 0 && pv) {
#endif
                if (nsplit == 1) {
                    goff_b[(size_t)(2 * t) * g.HoWo + P] = s_h;
                    goff_b[(size_t)(2 * t + 1) * g.HoWo + P] = s_w;
                    gmsk_b[(size_t)t * g.HoWo + P] = s_m;
                } else {
                    atomicAdd(goff_b + (size_t)(2 * t) * g.HoWo + P, s_h);
                    atomicAdd(goff_b + (size_t)(2 * t + 1) * g.HoWo + P, s_w);
                    atomicAdd(gmsk_b + (size_t)t * g.HoWo + P, s_m);
                }
            }
            if (__any(a1 | a2 | a3 | a4)) {
                float *gin_g = gin_b + (size_t)blk * BD_CB * HW;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cc = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (cc < cleft) {
                        const float dm = acc[r] * m;
                        float *gp = gin_g + (size_t)cc * HW;
                        if (a1) atomicAdd(gp + i1, dm * w1);
                        if (a2) atomicAdd(gp + i2, dm * w2);
                        if (a3) atomicAdd(gp + i3, dm * w3);
                        if (a4) atomicAdd(gp + i4, dm * w4);
                    }
                }
            }
            if (t + 1 < 9) w_commit((t + 1) & 1);     // that buffer was last read in tap t-1, which every wave has left
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward w.r.t. weight.  D[k', o] = sum_p col[k', p] dY[o, p].
// grid = (ceil(RB/4), S, ceil(Cop/32/MB)); block = 256: wave w owns row-block 4*blockIdx.x + w
// (32 k' rows = 32 channels of one (group, tap)) and MB 32-wide output-channel blocks.
// Each block walks its share of (image, 32-pixel tile) pairs; per tile the sampled columns and
// the dY tile are transposed through LDS (row stride 33 -> conflict-free ds_read_b32).
// ---------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(256) void dcn_bwd_weight_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                          const float *__restrict__ msk, const float *__restrict__ gy,
                                                          float *__restrict__ gw, Geom g, int tiles_per_img, int nsplit,
                                                          const unsigned *__restrict__ far_scal, const int *__restrict__ far_list)
{
    // far-only mode (far_scal != nullptr): the tiled kernel took every sample with |offset| < TL_NEAR; this kernel adds the
    // rest, walking only the tiles dcn_offset_absmax listed (far_scal[1] of them; usually zero -> uniform early exit).
    const bool far_only_absmax = far_scal != nullptr && !far_dominated(far_scal, g.B * tiles_per_img);          // else: everything
    if (far_only_absmax && far_scal[1] == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *colT = smem;                       // [4][32][33]
    float *dyT = smem + 4 * 32 * 33;          // [MB*32][33]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int HW = g.H * g.W;
    const int nblk = g.cpgp / 32;
    const int RB = g.dg * g.KK * nblk;
    const int rb = blockIdx.x * 4 + wave;
    const bool rbv = rb < RB;
    const int rbc = rbv ? rb : RB - 1;
    // row-blocks are numbered channel-block major: the 4 waves of a workgroup take 4 consecutive TAPS of the same 32
    // channels, so their gathers (same planes, neighbouring positions) share L1 lines while the barriers keep them in step
    const int S_ = g.dg * g.KK;
    const int blk = rbc / S_, seg = rbc - blk * S_;
    const int grp = seg / g.KK, t = seg - grp * g.KK;
    const int ob0 = blockIdx.z * MB;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    const int total = far_only_absmax ? (int)far_scal[1] : g.B * tiles_per_img;
    const int t0 = (int)((int64_t)blockIdx.y * total / nsplit), t1 = (int)((int64_t)(blockIdx.y + 1) * total / nsplit);
    if (t0 >= t1) return;                     // fewer listed tiles than splits: nothing to add, nothing to flush
    float *myT = colT + wave * 32 * 33;
    bool any_act = false;                     // did this wave (tap) accumulate anything?  (wave-uniform)

    // Three-deep software pipeline over the (image, tile) sequence of this block:
    //   stage R: raw offset/mask loads of tile ti+2      (feeds address generation)
    //   stage G: 64 bilinear-corner gathers + dY rows of tile ti+1 into registers
    //   stage M: tile ti -- registers -> LDS (transposed), barrier, 16 x MB MFMAs from LDS
    // so the two dependent global-memory latencies are covered by the matrix work of earlier tiles.
    auto tile_coords = [&](int ti, int &b, int &P, bool &pv, int &Pc, int &ho, int &wo) {
        if (far_only_absmax) ti = far_list[ti];
        b = ti / tiles_per_img;
        const int tile = ti - b * tiles_per_img;
        P = tile * 32 + p;
        pv = P < g.HoWo;
        Pc = pv ? P : g.HoWo - 1;
        ho = Pc / g.Wo;
        wo = Pc - ho * g.Wo;
    };
    auto stage_raw = [&](int ti) {
        int b, P, Pc, ho, wo; bool pv;
        tile_coords(ti, b, P, pv, Pc, ho, wo);
        return load_tap_raw(off + (size_t)b * g.dg * 2 * g.KK * g.HoWo, msk + (size_t)b * g.dg * g.KK * g.HoWo, g, seg, Pc);
    };
    float v[16][4], dyr[MB * 4];
    Tap s;
    bool act_cur = true, act_next = true;      // does this wave contribute to tile ti / ti+1 (always, outside far-only mode)
    auto stage_gather = [&](int ti, const TapRaw &raw) {
        int b, P, Pc, ho, wo; bool pv;
        tile_coords(ti, b, P, pv, Pc, ho, wo);
        s = finish_tap(raw, g, t, ho, wo, pv && rbv);
        if (far_only_absmax && fabsf(raw.oh) < TL_NEAR && fabsf(raw.ow) < TL_NEAR) s.m = 0.f;     // near sample: already counted
        // far-only mode: a listed tile usually has ONE far sample, i.e. eight of the nine taps (waves) have nothing to add:
        // they skip their 64 gathers and 16*MB MFMAs and only keep the barriers and their share of the dY tile
        act_next = !far_only_absmax || __any(s.m != 0.f);
        const float *in_g = in + ((size_t)b * g.C + (size_t)grp * g.cpg) * HW;
        const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;
        if (act_next)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cc = blk * 32 + 2 * q + h;
            const float *ip = in_g + (size_t)(cc < g.cpg ? cc : 0) * HW;
#ifdef DW_ABL_NOGATHER
            if (true) { v[q][0] = s.a0; v[q][1] = s.a1 + (float)q; v[q][2] = s.b0; v[q][3] = s.b1; (void)ip; } else
#endif
            if (g.W >= 2) {
                const f32x2 tp = ldg2(ip, (unsigned)s.pt * 4u), bt = ldg2(ip, (unsigned)s.pb * 4u);
                v[q][0] = tp.x; v[q][1] = tp.y; v[q][2] = bt.x; v[q][3] = bt.y;
            } else {
                v[q][0] = ip[s.i1]; v[q][1] = ip[s.i2]; v[q][2] = ip[s.i3]; v[q][3] = ip[s.i4];
            }
        }
#pragma unroll
        for (int q = 0; q < MB * 4; ++q) {
            const int o = ob0 * 32 + q * 8 + wave * 2 + h;
            dyr[q] = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
        }
    };
    TapRaw raw_next = {0.f, 0.f, 0.f};
    if (t0 < t1) {
        stage_gather(t0, stage_raw(t0));
        act_cur = act_next;
        if (t0 + 1 < t1) raw_next = stage_raw(t0 + 1);
    }
    for (int ti = t0; ti < t1; ++ti) {
        // registers of tile ti -> LDS: sampled columns colT[row][pixel], dY tile dyT[o][pixel]
        const bool act = act_cur;
        any_act |= act;
        if (act)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = 2 * q + h;
            const bool pr = g.W >= 2;
            const float val = ((pr ? s.a0 : s.w1) * v[q][0] + (pr ? s.a1 : s.w2) * v[q][1] + (pr ? s.b0 : s.w3) * v[q][2] +
                               (pr ? s.b1 : s.w4) * v[q][3]) * s.m;
            myT[row * 33 + p] = (blk * 32 + row < g.cpg) ? val : 0.f;
        }
#pragma unroll
        for (int q = 0; q < MB * 4; ++q) dyT[(q * 8 + wave * 2 + h) * 33 + p] = dyr[q];
        __syncthreads();
        if (ti + 1 < t1) {
            const TapRaw raw = raw_next;
            if (ti + 2 < t1) raw_next = stage_raw(ti + 2);
            stage_gather(ti + 1, raw);
            act_cur = act_next;
        }
        // contraction over the 32 pixels of tile ti
#ifdef DW_ABL_NOMFMA
        if (ti < 0)
#endif
        if (act)
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const float a = myT[p * 33 + 2 * k + h];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dyT[(mb * 32 + p) * 33 + 2 * k + h], acc[mb], 0, 0, 0);
        }
        __syncthreads();
    }

    // lane holds D[row = (r&3)+8*(r>>2)+4h][o = ob*32 + p].  Transposed through this wave's LDS region so that the flush
    // runs with lanes along the channel axis (36-byte stride, 9 lines per half-wave) instead of along o (C*KK*4-byte
    // stride: one cache-line request per lane, ~21 G/s -- that flush alone cost 0.7 ms on the 512->256 layer).
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) myT[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + p] = acc[mb][r];
        __syncthreads();
        const int cc = blk * 32 + p;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int ol = 2 * it + h;
            const int o = (ob0 + mb) * 32 + ol;
            if (any_act && rbv && o < g.Co && cc < g.cpg)
                atomicAdd(gw + ((size_t)o * g.C + grp * g.cpg + cc) * g.KK + t, myT[p * 33 + ol]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward w.r.t. weight, workgroup-tiled (3x3 / stride 1 / pad 1 / dil 1 / one group): no transposes, no global gathers.
//
// dW[o][c][t] = sum_p dY[o][p] col[c,t][p] contracts over PIXELS, so the MFMA wants A[i = c][k = pixel]: lane = channel.
// From global memory that layout is uncoalesced; from an LDS window it is just another conflict-free pattern (plane stride
// 2*odd -> 32 channels hit 32 distinct even banks, the x+1 corner the odd ones).  So per 2 x 32-pixel tile the workgroup
// stages the 10 x 40 window of a 32-channel block and the dY tile (stride 65: lane = o conflict-free), and each of its six
// waves = (tile row, kernel row ky) runs 16 steps x 3 taps of: sampling state of pixel 2s+h broadcast from the lane that
// owns it (ds_bpermute), two ds_read2_b32 corner pairs for ITS channel, bilinear, MFMA against the dY operand.
// Accumulators (3 taps x 64 outputs) persist over the workgroup's share of tiles; one atomic flush at the end.
// Samples displaced by >= 3 px (outside any window) are zero-weighted here and taken by dcn_bwd_weight_f32 in its
// far-only mode, which exits immediately when max|offset| < 3 (device scalar from dcn_offset_absmax).
// ---------------------------------------------------------------------------------------------
constexpr int DW_TR = 2;
constexpr int DW_WH = DW_TR + 8;                      // 10 window rows
constexpr int DW_PLANE = DW_WH * TL_WW + 2;           // 402 = 2 * 201
constexpr int DW_CB = 32;                             // channels per block
constexpr int DW_IN_FLOATS = DW_CB * DW_PLANE;        // 12 864
constexpr int DW_DYS = DW_TR * 32 + 1;                // dY row stride 65
constexpr int DW_DY_FLOATS = TL_OB * DW_DYS;          // 4 160
constexpr int DW_NT = DW_TR * 3 * 64;                 // 384 threads

__global__ __launch_bounds__(DW_NT, 3) void dcn_bwd_weight_tile_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                                const float *__restrict__ msk, const float *__restrict__ gy,
                                                                float *__restrict__ part, Geom g, int tiles_x, int tiles_y,
                                                                int nsplit, const unsigned *__restrict__ far_scal)
{
    if (far_dominated(far_scal, g.B * ((g.HoWo + 31) / 32))) return;      // the generic kernel does the whole job instead
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [DW_IN_FLOATS | DW_DY_FLOATS]
    float *win = lds, *dyt = lds + DW_IN_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int row = wave & 1, ky = wave >> 1;
    const int cb = blockIdx.x, zo = blockIdx.z;
    const int HW = g.H * g.W;
    const int total = g.B * tiles_x * tiles_y;
    const int t0 = (int)((int64_t)blockIdx.y * total / nsplit), t1 = (int)((int64_t)(blockIdx.y + 1) * total / nsplit);

    f32x16 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][mb][r] = 0.f;

    constexpr int KIN = (DW_CB * DW_WH * 10 + DW_NT - 1) / DW_NT;      // 9 window dwordx4 per thread
    constexpr int KDY = (TL_OB * DW_TR * 8 + DW_NT - 1) / DW_NT;       // 3 dY dwordx4 per thread

    for (int ti = t0; ti < t1; ++ti) {
        const int b = ti / (tiles_x * tiles_y);
        const int rem = ti - b * (tiles_x * tiles_y);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int r0 = ty * DW_TR, c0 = tx * 32;
        const int Y0 = r0 - 4, X0 = c0 - 4;
        const float *in_b = in + ((size_t)b * g.C + (size_t)cb * DW_CB) * HW;
        const float *gy_b = gy + ((size_t)b * g.Co + (size_t)zo * TL_OB) * g.HoWo;
        const int cleft = g.C - cb * DW_CB, oleft = g.Co - zo * TL_OB;

        // ---- sampling state of this wave's (row, ky), lane = pixel
        const int ho = r0 + row, wo = c0 + i;
        const bool pv = ho < g.Ho && wo < g.Wo;
        const int Pc = pv ? ho * g.Wo + wo : 0;
        const float *off_b = off + (size_t)b * 18 * g.HoWo;
        const float *msk_b = msk + (size_t)b * 9 * g.HoWo;
        int posv[3];
        float wv[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int t = ky * 3 + kx;
#ifdef DWT_ABL_NOTAP
            const TapRaw raw = {0.1f * (float)t, 0.2f, 0.5f};
            (void)off_b; (void)msk_b;
#else
            const TapRaw raw = load_tap_raw(off_b, msk_b, g, t, Pc);
#endif
            const float hf = (float)(ho - 1 + ky) + raw.oh, wf_ = (float)(wo - 1 + kx) + raw.ow;
            const bool sv = pv && hf > -1.f && wf_ > -1.f && hf < (float)g.H && wf_ < (float)g.W &&
                            fabsf(raw.oh) < TL_NEAR && fabsf(raw.ow) < TL_NEAR;
            const float hlf = floorf(hf), wlf = floorf(wf_);
            const int ly = sv ? (int)hlf - Y0 : 0, lx = sv ? (int)wlf - X0 : 0;
            posv[kx] = ly * TL_WW + lx;
            const float lh = hf - hlf, lw = wf_ - wlf;
            const float m = sv ? raw.m : 0.f;
            wv[kx][0] = (1.f - lh) * (1.f - lw) * m; wv[kx][1] = (1.f - lh) * lw * m;
            wv[kx][2] = lh * (1.f - lw) * m;         wv[kx][3] = lh * lw * m;
        }

#ifndef DWT_ABL_NOBAR
        __syncthreads();                                   // previous tile fully consumed
#endif
#ifdef DWT_ABL_NOSTAGE
        if (ti == t0)
#endif
        {
        // ---- stage the window (three batches of three dwordx4 per thread: bounded register use) and the dY tile
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        typedef float f32x2_ __attribute__((ext_vector_type(2)));
#pragma unroll 1
        for (int k0 = 0; k0 < KIN; k0 += 3) {
            f32x4 rin[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = tid + DW_NT * (k0 + k);
                const int ch = e / (DW_WH * 10), rm = e - ch * (DW_WH * 10);
                const int wr = rm / 10, q = rm - wr * 10;
                const int y = Y0 + wr, x = X0 + 4 * q;
                rin[k] = zero4;
                if (e < DW_CB * DW_WH * 10 && ch < cleft && y >= 0 && y < g.H && x >= 0 && x < g.W)
                    rin[k] = *reinterpret_cast<const f32x4 *>(in_b + (size_t)ch * HW + y * g.W + x);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = tid + DW_NT * (k0 + k);
                const int ch = e / (DW_WH * 10), rm = e - ch * (DW_WH * 10);
                if (e < DW_CB * DW_WH * 10) {
                    float *d = win + ch * DW_PLANE + rm * 4;
                    *reinterpret_cast<f32x2_ *>(d) = f32x2_{rin[k].x, rin[k].y};
                    *reinterpret_cast<f32x2_ *>(d + 2) = f32x2_{rin[k].z, rin[k].w};
                }
            }
        }
        {
            f32x4 rdy[KDY];
#pragma unroll
            for (int k = 0; k < KDY; ++k) {
                const int e = tid + DW_NT * k;
                const int o = e / (DW_TR * 8), rm = e - o * (DW_TR * 8);
                const int wr = rm / 8, q = rm - wr * 8;
                const int y = r0 + wr, x = c0 + 4 * q;
                rdy[k] = zero4;
                if (e < TL_OB * DW_TR * 8 && o < oleft && y < g.Ho && x < g.Wo)
                    rdy[k] = *reinterpret_cast<const f32x4 *>(gy_b + (size_t)o * g.HoWo + y * g.Wo + x);
            }
#pragma unroll
            for (int k = 0; k < KDY; ++k) {
                const int e = tid + DW_NT * k;
                const int o = e / (DW_TR * 8), rm = e - o * (DW_TR * 8);
                if (e < TL_OB * DW_TR * 8) {
                    float *d = dyt + o * DW_DYS + rm * 4;
                    d[0] = rdy[k].x; d[1] = rdy[k].y; d[2] = rdy[k].z; d[3] = rdy[k].w;
                }
            }
        }
        }
#ifndef DWT_ABL_NOBAR
        __syncthreads();
#endif

        const float *wl_ = win + i * DW_PLANE;                 // this lane's channel plane
        const float *dl0 = dyt + i * DW_DYS + row * 32 + h, *dl1 = dl0 + 32 * DW_DYS;
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const int src = 2 * s + h;                          // lane that owns pixel 2s+h of this row
            const float d0 = dl0[2 * s], d1 = dl1[2 * s];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#ifdef DWT_ABL_NOSHFL
                const int pos = posv[kx] + (src & 1);
                const float q0 = wv[kx][0], q1 = wv[kx][1], q2 = wv[kx][2], q3 = wv[kx][3];
#else
                const int pos = __shfl(posv[kx], src);
                const float q0 = __shfl(wv[kx][0], src), q1 = __shfl(wv[kx][1], src);
                const float q2 = __shfl(wv[kx][2], src), q3 = __shfl(wv[kx][3], src);
#endif
                const float *cp = wl_ + pos;
#ifdef DWT_ABL_NOGATHER
                const float val = q0 + q1 * (float)pos + q2 + q3 * (float)s;
                (void)cp;
#else
                const float val = q0 * cp[0] + q1 * cp[1] + q2 * cp[TL_WW] + q3 * cp[TL_WW + 1];
#endif
#ifdef DWT_ABL_NOMFMA
                acc[kx][0][s] += val * d0;
                acc[kx][1][s] += val * d1;
#else
                acc[kx][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(val, d0, acc[kx][0], 0, 0, 0);
                acc[kx][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(val, d1, acc[kx][1], 0, 0, 0);
#endif
            }
        }
    }

    // lane holds, per tap, D[c = (r&3)+8*(r>>2)+4h][o = 32 mb + i] in acc[kx][mb][r].  The two tile rows are summed in LDS,
    // laid out like the weight tensor ([o][c][t]: 288 contiguous floats per output channel), and written as this workgroup's
    // partial result with plain coalesced stores; dcn_dw_reduce sums the partials.  (Atomics straight into grad_weight cost
    // 0.11 ms on 64->64: 256 splits contend for the same 36 864 addresses; strided ones one cache-line request per lane.)
    constexpr int STG = DW_CB * 9 + 1;                     // 289: lanes (o) hit distinct banks
    float *stage = lds;                                    // 32 x 289 floats (reuses the window area)
    float *mine = part + ((size_t)(cb * gridDim.z + zo) * nsplit + blockIdx.y) * (2 * 32 * DW_CB * 9);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        __syncthreads();
        if (row == 0) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[i * STG + ((r & 3) + 8 * (r >> 2) + 4 * h) * 9 + ky * 3 + kx] = acc[kx][mb][r];
        }
        __syncthreads();
        if (row == 1) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    stage[i * STG + ((r & 3) + 8 * (r >> 2) + 4 * h) * 9 + ky * 3 + kx] += acc[kx][mb][r];
        }
        __syncthreads();
        for (int e = tid; e < 32 * DW_CB * 9; e += DW_NT) {
            const int ol = e / (DW_CB * 9), j = e - ol * (DW_CB * 9);
            mine[mb * (32 * DW_CB * 9) + e] = stage[ol * STG + j];
        }
    }
}

// grad_weight[o][c][t] += sum over the pixel splits of the tiled kernel's partials.  grid.y split-groups each sum a slice
// of the splits (coalesced over the 288-float rows) and add it to the zero-filled gradient: <= 16 atomics per address.
__global__ void dcn_dw_reduce(const float *__restrict__ part, float *__restrict__ gw, Geom g, int ncb, int nzo, int nsplit,
                              const unsigned *__restrict__ far_scal)
{
    if (far_dominated(far_scal, g.B * ((g.HoWo + 31) / 32))) return;      // no partials were written
    constexpr int PER = 2 * 32 * DW_CB * 9;                // floats per partial
    const int n = ncb * nzo * PER;
    const int s0 = (int)((int64_t)blockIdx.y * nsplit / gridDim.y), s1 = (int)((int64_t)(blockIdx.y + 1) * nsplit / gridDim.y);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int blk = idx / PER, e = idx - blk * PER;
        const int cb = blk / nzo, zo = blk - cb * nzo;
        const int mb = e / (32 * DW_CB * 9), e2 = e - mb * (32 * DW_CB * 9);
        const int ol = e2 / (DW_CB * 9), j = e2 - ol * (DW_CB * 9);
        const int o = zo * TL_OB + mb * 32 + ol, c = cb * DW_CB + j / 9;
        if (o >= g.Co || c >= g.C) continue;
        const float *src = part + (size_t)blk * nsplit * PER + e;
        float sum = 0.f;
#pragma unroll 8
        for (int sp = s0; sp < s1; ++sp) sum += src[(size_t)sp * PER];
        atomicAdd(gw + ((size_t)o * g.C + (size_t)cb * DW_CB) * 9 + j, sum);
    }
}

// One launch instead of up to six hipMemsetAsync calls (each is its own ~5 us kernel on the stream).
struct ZeroRanges {
    unsigned *p[8];
    unsigned n[8];        // dwords
    unsigned *mark_p = nullptr;   // one word set to mark_v (must lie outside the ranges)
    unsigned mark_v = 0u;
};
__device__ __forceinline__ void zero_ranges_body(const ZeroRanges &z, unsigned bid, unsigned nb)
{
    if (z.mark_p && bid == 0 && threadIdx.x == 0) *z.mark_p = z.mark_v;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        unsigned *p = z.p[r];
        const unsigned n = z.n[r];
        for (unsigned i = bid * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) p[i] = 0u;
    }
}
__global__ void dcn_zero_ranges(ZeroRanges z) { zero_ranges_body(z, blockIdx.x, gridDim.x); }

inline int pick_mb(int nb, int tiles_total)
{
    // largest MB in {8,4,2,1} (32-wide Cout blocks per wave) that still leaves >= 1024 waves
    for (int mb = 8; mb > 1; mb >>= 1) {
        if (mb > nb) continue;
        if ((int64_t)tiles_total * ((nb + mb - 1) / mb) >= 1024) return mb;
    }
    return 1;
}

#include "dcn_v2_bf16x3.inc"
#include "dcn_bwd_sweep.inc"

}  // namespace

namespace {

// DCN.forward's glue (dcn_v2.py:118-123): out (B, 3T, HW) of conv_offset_mask -> offset = its first 2T channels (contiguous copy),
// mask = sigmoid of the last T; and the adjoint.  One launch each instead of slice copy + sigmoid (forward) and two zero-fills,
// two slice copies, sigmoid backward and an accumulation (backward).  grid.x covers B * 3T * HW / 4 (HW % 4 == 0) or scalars.
template <int V>
__global__ void dcn_offset_mask_split(const float *__restrict__ out, float *__restrict__ offset, float *__restrict__ mask, int B, int T,
                                      int64_t HW)
{
    const int64_t per_img = (int64_t)3 * T * HW / V, n = (int64_t)B * per_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_img, r = (i - b * per_img) * V;       // r = c * HW + p
        const bool is_off = r < (int64_t)2 * T * HW;
        float *dst = is_off ? offset + b * 2 * T * HW + r : mask + b * T * HW + (r - (int64_t)2 * T * HW);
        if (V == 4) {
            f32x4 v = *reinterpret_cast<const f32x4 *>(out + b * 3 * T * HW + r);
            if (!is_off) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = 1.f / (1.f + __expf(-v[j]));
            }
            *reinterpret_cast<f32x4 *>(dst) = v;
        } else {
            const float v = out[b * 3 * T * HW + r];
            *dst = is_off ? v : 1.f / (1.f + __expf(-v));
        }
    }
}

template <int V>
__global__ void dcn_offset_mask_merge(const float *__restrict__ goff, const float *__restrict__ gmask, const float *__restrict__ mask,
                                      float *__restrict__ gout, int B, int T, int64_t HW)
{
    const int64_t per_img = (int64_t)3 * T * HW / V, n = (int64_t)B * per_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_img, r = (i - b * per_img) * V;
        const bool is_off = r < (int64_t)2 * T * HW;
        const int64_t rm = r - (int64_t)2 * T * HW;
        if (V == 4) {
            f32x4 v;
            if (is_off) v = *reinterpret_cast<const f32x4 *>(goff + b * 2 * T * HW + r);
            else {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(gmask + b * T * HW + rm);
                const f32x4 m = *reinterpret_cast<const f32x4 *>(mask + b * T * HW + rm);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = g[j] * m[j] * (1.f - m[j]);
            }
            *reinterpret_cast<f32x4 *>(gout + b * 3 * T * HW + r) = v;
        } else {
            float v;
            if (is_off) v = goff[b * 2 * T * HW + r];
            else {
                const float m = mask[b * T * HW + rm];
                v = gmask[b * T * HW + rm] * m * (1.f - m);
            }
            gout[b * 3 * T * HW + r] = v;
        }
    }
}

}  // namespace

#include "dcn_dense.inc"

extern "C" {

// partial grad_weight blocks of the tiled kernel: (channel blocks x output slices x pixel splits) x [64 o][32 c][9]
static size_t dw_partial_floats(int Cin, int Cout)
{
    const size_t nb = (size_t)((Cin + DW_CB - 1) / DW_CB) * ((Cout + TL_OB - 1) / TL_OB);
    return (nb > 512 ? nb : 512) * (size_t)(2 * 32 * DW_CB * 9);
}

// per (image, tap segment, 8x8 block of output pixels) one byte, after the grad_weight partials
static size_t tileflag_bytes(const Geom &g) { return ((size_t)g.B * ((g.H + 3) / 4) * ((g.W + 31) / 32) + 255) / 256 * 256; }

// [block maxima | tile flags of the tiled grad_input kernel]
static size_t blockmax_bytes(const Geom &g)
{
    return ((size_t)g.B * g.dg * g.KK * ((g.Ho + 7) / 8) * ((g.Wo + 7) / 8) + 255) / 256 * 256 + tileflag_bytes(g);
}

static size_t base_workspace_bytes(const Geom &g)
{
    // [Wf | Wb | scalars (256 B) | inverse lists: cnt, idx, w | far-tile flags | far-tile list | grad_weight partials]
    const size_t cells = (size_t)g.B * g.dg * g.KK * g.H * g.W;
    const size_t ntile = (size_t)g.B * ((g.HoWo + 31) / 32);
    const size_t n = (size_t)g.Kp * g.Cop * sizeof(float) * 2 + 256 + ((cells + 255) / 256 * 256) + cells * INV_CAP * 8 + 256 +
                     ((ntile + 255) / 256 * 256) + (ntile + 63) / 64 * 64 * sizeof(int) + dw_partial_floats(g.C, g.Co) * sizeof(float);
    return (n + 255) / 256 * 256 + blockmax_bytes(g);
}

size_t dcd_dcn_v2_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                  int pw, int dh, int dw, int dg)
{
    Geom g;
    if (!make_geom(g, B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return 0;
    // dense path (dcn_dense.inc): [column buffer | split-K partials] after the lists; then the one-pass backward's
    // [prepared weights | grad_offset / grad_mask partial planes | grad_weight partials] (dcn_bwd_sweep.inc)
    return base_workspace_bytes(g) + dense_workspace_bytes(g) + sweep_workspace_bytes(g);
}

int dcd_dcn_v2_forward(void *stream_, const float *input, const float *weight, const float *bias,
                       const float *offset, const float *mask, float *output, int B, int Cin, int H, int W,
                       int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                       int precision, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    Geom g;
    if (!input || !weight || !bias || !offset || !mask || !output || !workspace) return DCD_ERR_BAD_ARG;
    if (!make_geom(g, B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return DCD_ERR_BAD_ARG;
    // DCD_PREC_BF16X3 permits (does not oblige) the split-bf16 contraction: the workgroup-tiled kernels have it, every other
    // geometry runs the exact fp32 kernels, which are inside any tolerance the split form is
    // DCD_PREC_BF16 (one product of bf16-rounded operands): the same kernels and prepared weights, their low halves unread
    if (precision != DCD_PREC_F32 && precision != DCD_PREC_BF16X3 && precision != DCD_PREC_BF16) return DCD_ERR_BAD_ARG;
    const bool split = precision != DCD_PREC_F32;
    const bool one = precision == DCD_PREC_BF16;
    const size_t nw = (size_t)g.Kp * g.Cop;
    if (workspace_bytes < nw * sizeof(float) * 2) return DCD_ERR_WORKSPACE;
    if (dense_ok(g, false)) {                                  // wide input, small map: column buffer + GEMM
        if (workspace_bytes < base_workspace_bytes(g) + dense_workspace_bytes(g)) return DCD_ERR_WORKSPACE;
        dense_forward(stream, input, weight, bias, offset, mask, output, g, (float *)((char *)workspace + base_workspace_bytes(g)),
                      precision);
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    float *wf = (float *)workspace, *wb = wf + nw;

    const int tiles = (g.HoWo + 31) / 32;
    const int nb = g.Cop / 32;
    const int mb = pick_mb(nb, tiles * B);
    dim3 grid((tiles + 3) / 4, B, (nb + mb - 1) / mb), block(256);
#ifndef DCN_NO_FWD_TILE
    // workgroup-tiled LDS kernel: the DLA-34 shape (3x3, stride 1, pad 1, dil 1, dg 1); maps of at least 16 rows
    if (kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && dg == 1 && (W & 3) == 0 &&
        H >= 8 && W >= 32 && dcd_env("DCD_NO_TILE") == nullptr) {
        const int nz = (Cout + TL_OB - 1) / TL_OB;
        const int nchunk = split ? (Cin + TB_CH - 1) / TB_CH : (Cin + TL_CH - 1) / TL_CH;
        const size_t nwl = (size_t)nz * nchunk * (split ? TB_W_FLOATS : TL_W_FLOATS);
        if (nwl <= 2 * nw) {                                          // Wl lives in the [Wf | Wb] area
            static LdsLimit lds_limit8, lds_limit4, lds_limit8b, lds_limit4b;
            static int tile_rows = 0;
            if (tile_rows == 0) {
                const char *e = dcd_env("DCD_TILE_ROWS");
                tile_rows = (e && atoi(e) == 4) ? 4 : 8;
            }
            if (!lds_limit8.raise((int)(2 * TileCfg<8>::BUF * sizeof(float)), dcn_fwd_tile_f32<8>) ||
                !lds_limit4.raise((int)(2 * TileCfg<4>::BUF * sizeof(float)), dcn_fwd_tile_f32<4>))
                return DCD_ERR_LAUNCH;
            if (split && (!lds_limit8b.raise((int)(TileCfgB<8>::NBUF * TileCfgB<8>::BUF * sizeof(float)), dcn_fwd_tile_bf16x3<8, 3>,
                                             dcn_fwd_tile_bf16x3<8, 1>) ||
                          !lds_limit4b.raise((int)(TileCfgB<4>::NBUF * TileCfgB<4>::BUF * sizeof(float)), dcn_fwd_tile_bf16x3<4, 3>,
                                             dcn_fwd_tile_bf16x3<4, 1>)))
                return DCD_ERR_LAUNCH;
            static int rescue_taps = 0;
            if (rescue_taps == 0) {
                const char *e = dcd_env("DCD_FWD_RESCUE_TAPS");       // A/B: far taps in the worst wave that hand a region to the rescue kernel
                rescue_taps = (e && atoi(e) >= 1 && atoi(e) <= 10) ? atoi(e) : 5;
            }
            const int tiles_x = (g.Wo + 31) / 32;
            const bool rows8 = tile_rows == 8 && (int64_t)tiles_x * ((g.Ho + 7) / 8) * B * nz >= 512;
            const int TRr = rows8 ? 8 : 4;
            const int regions = tiles_x * ((g.Ho + TRr - 1) / TRr);
            // region flags ("far samples dominate here: left to the rescue pass") live behind the weight area
            const int nflag_words = (B * regions + 3) / 4 + 1;        // + the far-coordinate counter, zeroed with the flags
            unsigned char *flags = (unsigned char *)(wf + 2 * nw);
            unsigned *far_count = (unsigned *)flags + (nflag_words - 1);
            const int64_t ncoord = (int64_t)B * 18 * g.HoWo;
            const bool keeps = forward_keeps_far_samples(weight, ncoord);     // no far count, no rescue launch (policy above)
            const unsigned far_limit = keeps ? 0xffffffffu : (unsigned)(ncoord / 32 < 0xffffffffll ? ncoord / 32 : 0xffffffffll);
            const int rescue_taps_call = keeps ? 10 : rescue_taps;            // 10: no region is ever handed over (nine taps)
            const size_t n9 = (size_t)Cin * 9 * g.Cop;
            if (nwl <= nw && n9 <= nw && workspace_bytes >= 2 * nw * sizeof(float) + (size_t)nflag_words * 4) {
                float *wf9 = wf + nw;                         // [Wl | Wf9 | flags]
                if (split)
                    hipLaunchKernelGGL(dcn_prep_weights_tile_bf16, dim3((unsigned)((nwl + 255) / 256 < 2048 ? (nwl + 255) / 256 : 2048)),
                                       dim3(256), 0, stream, weight, (unsigned short *)wf, g, nchunk, nz, (unsigned *)flags, nflag_words, wf9);
                else
                    hipLaunchKernelGGL(dcn_prep_weights_tile, dim3((unsigned)((nwl + 255) / 256 < 2048 ? (nwl + 255) / 256 : 2048)),
                                       dim3(256), 0, stream, weight, wf, g, nchunk, nz, (unsigned *)flags, nflag_words, wf9);
                if (!keeps) {
                    int gsz = (int)((ncoord + 4095) / 4096);
                    if (gsz > 512) gsz = 512;
                    hipLaunchKernelGGL(dcn_fwd_far_count, dim3(gsz), dim3(256), 0, stream, offset, ncoord, far_count);
                }
#define DCD_FWD_TILE_B(TRV, NPV)                                                                                                      \
    hipLaunchKernelGGL((dcn_fwd_tile_bf16x3<TRV, NPV>), dim3(regions, B, nz), dim3(TRV * 64),                                         \
                       TileCfgB<TRV>::NBUF * TileCfgB<TRV>::BUF * sizeof(float), stream, input, offset, mask, (const float *)wf, bias, \
                       output, g, tiles_x, nchunk, flags, (const float *)wf9, rescue_taps_call, (const unsigned *)far_count, far_limit)
                if (split && rows8) {
                    if (one) DCD_FWD_TILE_B(8, 1); else DCD_FWD_TILE_B(8, 3);
                } else if (split) {
                    if (one) DCD_FWD_TILE_B(4, 1); else DCD_FWD_TILE_B(4, 3);
                }
#undef DCD_FWD_TILE_B
                else if (rows8)
                    hipLaunchKernelGGL(dcn_fwd_tile_f32<8>, dim3(regions, B, nz), dim3(512), 2 * TileCfg<8>::BUF * sizeof(float), stream,
                                       input, offset, mask, wf, bias, output, g, tiles_x, nchunk, flags, (const float *)wf9, rescue_taps_call,
                                       (const unsigned *)far_count, far_limit);
                else
                    hipLaunchKernelGGL(dcn_fwd_tile_f32<4>, dim3(regions, B, nz), dim3(256), 2 * TileCfg<4>::BUF * sizeof(float), stream,
                                       input, offset, mask, wf, bias, output, g, tiles_x, nchunk, flags, (const float *)wf9, rescue_taps_call,
                                       (const unsigned *)far_count, far_limit);
                if (keeps) return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
                // rescue pass: register-gather kernel over the flagged regions only (usually none: its waves exit at once)
                const int nb9 = g.Cop / 32, mb9 = nb9 >= 4 ? 4 : nb9 >= 2 ? 2 : 1;
                dim3 gridr((tiles_x * g.Ho + 3) / 4, B, (nb9 + mb9 - 1) / mb9);
                if (mb9 == 4)
                    hipLaunchKernelGGL((dcn_fwd9_f32<4, true>), gridr, dim3(256), 0, stream, input, offset, mask, wf9, bias, output, g,
                                       (const unsigned char *)flags, tiles_x, TRr, nchunk);
                else if (mb9 == 2)
                    hipLaunchKernelGGL((dcn_fwd9_f32<2, true>), gridr, dim3(256), 0, stream, input, offset, mask, wf9, bias, output, g,
                                       (const unsigned char *)flags, tiles_x, TRr, nchunk);
                else
                    hipLaunchKernelGGL((dcn_fwd9_f32<1, true>), gridr, dim3(256), 0, stream, input, offset, mask, wf9, bias, output, g,
                                       (const unsigned char *)flags, tiles_x, TRr, nchunk);
                return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
            }
        }
    }
#endif
#ifndef DCN_NO_FWD9
    if (g.KK == 9 && W >= 2 && (size_t)Cin * 9 * g.Cop <= nw) {      // 3x3 fast path (the only shape DGDE uses); reuses the Wf area
        const size_t n9 = (size_t)Cin * 9 * g.Cop;
        hipLaunchKernelGGL(dcn_prep_weights9, dim3((unsigned)((n9 + 255) / 256 < 2048 ? (n9 + 255) / 256 : 2048)), dim3(256), 0,
                           stream, weight, wf, g);
        const int mb9 = mb > 4 ? 4 : mb;                   // 72 VGPRs of tap state: keep <= 64 accumulator registers
        dim3 grid9((tiles + 3) / 4, B, (nb + mb9 - 1) / mb9);
        switch (mb9) {
            case 4: hipLaunchKernelGGL(dcn_fwd9_f32<4>, grid9, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
            case 2: hipLaunchKernelGGL(dcn_fwd9_f32<2>, grid9, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
            default: hipLaunchKernelGGL(dcn_fwd9_f32<1>, grid9, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
        }
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
#endif
    hipLaunchKernelGGL(dcn_prep_weights, dim3((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048)), dim3(256),
                       0, stream, weight, wf, wb, g);
    switch (mb) {
        case 8: hipLaunchKernelGGL(dcn_fwd_f32<8>, grid, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
        case 4: hipLaunchKernelGGL(dcn_fwd_f32<4>, grid, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
        case 2: hipLaunchKernelGGL(dcn_fwd_f32<2>, grid, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
        default: hipLaunchKernelGGL(dcn_fwd_f32<1>, grid, block, 0, stream, input, offset, mask, wf, bias, output, g); break;
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_dcn_v2_backward(void *stream_, const float *input, const float *weight, const float *bias,
                        const float *offset, const float *mask, const float *grad_output, float *grad_input,
                        float *grad_offset, float *grad_mask, float *grad_weight, float *grad_bias, int B,
                        int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                        int dw, int dg, int precision, void *workspace, size_t workspace_bytes)
{
    (void)bias;
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    Geom g;
    if (!input || !weight || !offset || !mask || !grad_output || !grad_input || !grad_offset || !grad_mask ||
        !grad_weight || !grad_bias || !workspace)
        return DCD_ERR_BAD_ARG;
    if (!make_geom(g, B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return DCD_ERR_BAD_ARG;
    if (precision != DCD_PREC_F32 && precision != DCD_PREC_BF16X3 && precision != DCD_PREC_BF16) return DCD_ERR_BAD_ARG;
    const bool split = precision != DCD_PREC_F32;              // the generic kernels: DCD_PREC_BF16 runs their split form
    const bool one = precision == DCD_PREC_BF16;
    const size_t nw = (size_t)g.Kp * g.Cop;
    if (workspace_bytes + 256 < dcd_dcn_v2_workspace_bytes(B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg))
        return DCD_ERR_WORKSPACE;
    float *wf = (float *)workspace, *wb = wf + nw;
    const size_t cells = (size_t)B * dg * g.KK * H * W;
    InvLists inv;
    unsigned *absmax = (unsigned *)(wb + nw);
    inv.absmax_bits = absmax;
    inv.cnt = (unsigned char *)absmax + 256;
    inv.idx = (int *)(inv.cnt + (cells + 255) / 256 * 256);
    inv.w = (float *)(inv.idx + cells * INV_CAP);
    inv.packed = 0;
    const size_t ntile = (size_t)B * ((g.HoWo + 31) / 32);
    unsigned char *far_flag = (unsigned char *)(((uintptr_t)(inv.w + cells * INV_CAP) + 255) / 256 * 256);
    int *far_list = (int *)(far_flag + (ntile + 255) / 256 * 256);
    float *dw_part = (float *)(far_list + (ntile + 63) / 64 * 64);
    unsigned char *blockmax = (unsigned char *)workspace + (base_workspace_bytes(g) - blockmax_bytes(g));
    inv.blockmax = blockmax;
    unsigned char *tileflag = (unsigned char *)workspace + (base_workspace_bytes(g) - tileflag_bytes(g));
    inv.tileflag = nullptr;                                    // set below when the tiled grad_input kernel is in play
    inv.flag_tx = (W + 31) / 32;
    inv.flag_ty = (H + 3) / 4;
    const dim3 bm_grid(((g.Ho + 7) / 8) * ((g.Wo + 7) / 8), dg * g.KK, B);

    // one-pass backward wherever it applies (3x3 / stride 1 / pad 1, Cout <= 256; round 3: Cout <= 64 only), the dense path's
    // layers included: 256->64 @ 24x80 0.36 vs 0.50 ms.  DCD_BWD_SWEEP=2 keeps the dense path for the layers it takes,
    // DCD_SWEEP_WIDE=0 keeps round 3's limit of 64 outputs (A/B)
    const char *sw_e = dcd_env("DCD_SWEEP_WIDE");              // read per call: the tests switch it inside one process
    const bool sweep_wide = !(sw_e && atoi(sw_e) == 0);
    // Cout 256 (the two deepest DGDE layers, Cin 512 / 256 on 12x40 / 24x80 maps) stays on the dense path when it applies: the
    // one-pass kernel gains little there (0.68 vs 0.73 ms, 0.46 vs 0.44) and those layers' offsets (fan-in 9 Cin) are large enough in
    // the train step to hand the call to the generic kernels (1.9 vs 0.8 ms), which the dense path does not care about
    const bool use_sweep = sweep_ok(g) && (g.Cop <= 64 || (sweep_wide && (g.Cop <= 128 || !dense_ok(g, true)))) && dense_mode() != 1 &&
                           (sweep_mode() == 1 || !dense_ok(g, true));
    unsigned *dense_report = nullptr;
    const bool far_to_dense = use_sweep && dense_ok(g, true) &&
                              handover_far_dominated(weight, far_count_limit(g, g.Co > 64), &dense_report);
    if ((!use_sweep || far_to_dense) && dense_ok(g, true)) {
        ZeroRanges z;
        for (int r = 0; r < 8; ++r) { z.p[r] = nullptr; z.n[r] = 0; }
        z.p[0] = absmax; z.n[0] = 4;
        z.p[1] = (unsigned *)far_flag; z.n[1] = (unsigned)((ntile + 3) / 4);
        z.p[3] = (unsigned *)grad_bias; z.n[3] = (unsigned)Cout;
        hipLaunchKernelGGL(dcn_zero_ranges, dim3(8), dim3(256), 0, stream, z);
        hipLaunchKernelGGL(dcn_prep_weights, dim3((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048)), dim3(256),
                           0, stream, weight, wf, wb, g);
        const int64_t noff = (int64_t)B * dg * 2 * g.KK * g.HoWo;
        int gsz = (int)((noff + 4095) / 4096);
        if (gsz > 512) gsz = 512;
        hipLaunchKernelGGL(dcn_offset_absmax, dim3(gsz), dim3(256), 0, stream, offset, noff, absmax, g.HoWo, dg * 2 * g.KK,
                           (g.HoWo + 31) / 32, far_flag, far_list);
        if (dense_report) hipLaunchKernelGGL(dcn_far_report, dim3(1), dim3(64), 0, stream, (const unsigned *)absmax, dense_report);
        hipLaunchKernelGGL(dcn_offset_blockmax, bm_grid, dim3(64), 0, stream, offset, g, blockmax);
        hipLaunchKernelGGL(dcn_build_inverse, dim3((H * W + 255) / 256, dg * g.KK, B), dim3(256), 0, stream, offset, mask, inv, g);
        int splits = (int)(((int64_t)g.HoWo + 4095) / 4096);
        if (splits > 32) splits = 32;
        hipLaunchKernelGGL(dcn_bias_grad, dim3(Cout, splits), dim3(256), 0, stream, grad_output, grad_bias, B, Cout, g.HoWo);
        dense_backward(stream, input, wb, offset, mask, grad_output, grad_input, grad_offset, grad_mask, grad_weight, g, inv,
                       (float *)((char *)workspace + base_workspace_bytes(g)), precision);
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    const int tiles = (g.HoWo + 31) / 32;
    const int nblk = g.cpgp / 32;
    if (!use_sweep)
        hipLaunchKernelGGL(dcn_prep_weights, dim3((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048)), dim3(256),
                           0, stream, weight, wf, wb, g);
    if (use_sweep) {
        // ---- one-pass backward (dcn_bwd_sweep.inc).  Near samples AND the (few) far ones: dcn_bwd_sweep.  Far samples dominating the
        // call (device-side decision from the offset scan's count): the sweep returns at once, the inverse lists are built after all
        // and the generic kernels do everything, as in the three-pass path (their launches below find nothing to do otherwise).
        // Launches: prologue A (weight layouts, scalars), prologue B (zero fill, offset scan), sweep, [product], epilogue (three
        // reductions), five guarded generic kernels = 9 (round 3: 13).
        const SweepPlan sp = sweep_plan(g);
        float *swp = (float *)((char *)workspace + base_workspace_bytes(g) + dense_workspace_bytes(g));
        float *cpart = swp + sp.wp_floats, *dwpart = cpart + sp.cpart_floats;
        const unsigned *fs = (const unsigned *)absmax;
        // the generic data kernel's channel blocks over grid.z (when it runs it does the whole call); with more than
        // one block it accumulates grad_offset / grad_mask with atomics onto what the epilogue wrote (zeros then)
        int nsplit = 1;
        while (nsplit * 2 <= nblk) nsplit *= 2;
        unsigned *far_report = nullptr;
        const unsigned real_limit = far_count_limit(g, sp.nob > 1);
        const int handover = handover_decide(stream, weight, real_limit, &far_report);
        const unsigned far_limit = handover ? real_limit : FAR_COUNT_PIVOT - 1;      // "never": no kernel sees the call as far-dominated
        {
            // the generic kernels' weight layouts are only read after a hand-over
            const int nb_gen = handover ? (int)((nw + 255) / 256 < 1024 ? (nw + 255) / 256 : 1024) : 0;
            const int nb_sw = (int)((sp.wp_floats + 255) / 256 < 2048 ? (sp.wp_floats + 255) / 256 : 2048);
            hipLaunchKernelGGL(dcn_sweep_prologue_a, dim3(nb_gen + nb_sw), dim3(256), 0, stream, weight, wf, wb, swp, absmax, g, sp.nck, sp.nob,
                               split ? 1 : 0, nb_gen, far_limit);
        }
        {
            ZeroRanges z;
            for (int r = 0; r < 8; ++r) { z.p[r] = nullptr; z.n[r] = 0; }
            z.p[2] = (unsigned *)grad_weight; z.n[2] = (unsigned)((size_t)Cout * Cin * g.KK);
            z.p[3] = (unsigned *)grad_bias; z.n[3] = (unsigned)Cout;
            z.p[4] = (unsigned *)grad_input; z.n[4] = (unsigned)((size_t)B * Cin * H * W);
            const int nb_zero = (int)(z.n[4] / 1024 + 1 < 4096 ? z.n[4] / 1024 + 1 : 4096);
            const int64_t noff = (int64_t)B * 18 * g.HoWo;
            int nb_scan = (int)((noff + 4095) / 4096);
            if (nb_scan > 512) nb_scan = 512;
            // no tile list (far_flag = null inside): the sweep takes its far samples itself; the far-only launches below then find
            // nothing listed and return at once unless the call is handed to the generic kernels altogether
            hipLaunchKernelGGL(dcn_sweep_prologue_b, dim3(nb_zero + nb_scan), dim3(256), 0, stream, z, nb_zero, offset, noff, absmax, g.HoWo,
                               (g.HoWo + 31) / 32);
        }
        {
            SweepArgs a;
            a.in = input; a.off = offset; a.msk = mask; a.wp = swp; a.gy = grad_output; a.gin = grad_input; a.cpart = cpart;
            a.dwpart = dwpart; a.far_scal = fs; a.g = g; a.nstrip = sp.nstrip; a.nseg = sp.nseg; a.seg_rows = sp.seg_rows;
            a.nck = sp.nck; a.nv = sp.nv; a.nslot = sp.nslot;
            a.col = dwpart + sp.dw_floats;                   // nob > 1: [grad_weight product partials | col]
            static LdsLimit sw_lds_limit;
            const int ldsb = SW_WAVES * SW_LDS_FLOATS * (int)sizeof(float);
            if (!sw_lds_limit.raise(ldsb, dcn_bwd_sweep<DCD_PREC_F32, 1, true>, dcn_bwd_sweep<DCD_PREC_BF16X3, 1, true>,
                                    dcn_bwd_sweep<DCD_PREC_F32, 2, false>, dcn_bwd_sweep<DCD_PREC_BF16X3, 2, false>,
                                    dcn_bwd_sweep<DCD_PREC_F32, 4, false>, dcn_bwd_sweep<DCD_PREC_BF16X3, 4, false>,
                                    dcn_bwd_sweep<DCD_PREC_BF16, 1, true>, dcn_bwd_sweep<DCD_PREC_BF16, 2, false>,
                                    dcn_bwd_sweep<DCD_PREC_BF16, 4, false>))
                return DCD_ERR_LAUNCH;
            const dim3 sgrid(sp.nslot / SW_WAVES), sblock(64 * SW_WAVES);
#define DCD_LAUNCH_SWEEP(NOBV, DWKV)                                                                                     \
    do {                                                                                                                \
        if (one) hipLaunchKernelGGL((dcn_bwd_sweep<DCD_PREC_BF16, NOBV, DWKV>), sgrid, sblock, ldsb, stream, a);        \
        else if (split) hipLaunchKernelGGL((dcn_bwd_sweep<DCD_PREC_BF16X3, NOBV, DWKV>), sgrid, sblock, ldsb, stream, a); \
        else hipLaunchKernelGGL((dcn_bwd_sweep<DCD_PREC_F32, NOBV, DWKV>), sgrid, sblock, ldsb, stream, a);             \
    } while (0)
            if (sp.nob == 1) DCD_LAUNCH_SWEEP(1, true);
            else if (sp.nob == 2) DCD_LAUNCH_SWEEP(2, false);
            else DCD_LAUNCH_SWEEP(4, false);
#undef DCD_LAUNCH_SWEEP
        }
        {
            SweepEpilogueArgs e;
            e.cpart = cpart; e.gy = grad_output; e.dwpart = dwpart; e.goff = grad_offset; e.gmsk = grad_mask; e.gw = grad_weight;
            e.gbias = grad_bias; e.far_scal = fs; e.g = g; e.nck = sp.nck; e.total_tiles = B * tiles;
            e.far_report = far_report; e.far_base = FAR_COUNT_PIVOT - far_limit;
            // nob == 1: the sweep's chunk-0 waves sum dY themselves (dcn_bias_grad's blocks then only act when the generic kernels
            // take the call over)
            e.bias_only_if_far = sp.nob == 1 ? fs : (const unsigned *)nullptr;
            const int64_t n4 = (int64_t)B * 27 * g.HoWo / 4;
            e.nb_coord = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
            int splits = (int)(((int64_t)g.HoWo + 4095) / 4096);
            if (splits > 32) splits = 32;
            if (splits < 1) splits = 1;
            e.bias_splits = splits;
            int nb_dw;
            if (sp.nob == 1) {
                // grad_weight and grad_bias (the sweep's chunk-0 waves summed dY over their pixels) from the per-wave partials
                const int nred = sp.nck * SW_DW_FLOATS + 64;
                e.nvp = sp.nslot / sp.nck;
                e.dw_bx = (nred + 255) / 256;
                e.dw_by = e.nvp >= 64 ? 16 : (e.nvp >= 8 ? 4 : 1);
                e.gemm_n = 0; e.gemm_S = 0;
                nb_dw = e.dw_bx * e.dw_by;
            } else {
                // grad_weight = sum_b dY[b] (Cout x HoWo) col[b]^T (HoWo x 9 Cin): ONE batched product, both operands pixel-contiguous,
                // partials per (image, pixel chunk) summed in a fixed order -- the near samples' part (the sweep's far loop wrote the
                // far samples' masked values into col as well).  Its row index c * 9 + t IS grad_weight's layout.
                static_assert(SW_GEMM_T == SG_T && SW_GEMM_K == SG_K, "sweep_plan sizes the product's partials with these");
                const int K9 = Cin * 9, n = Cout * K9;
                SgemmArgs ga;
                ga.A = grad_output; ga.B = dwpart + sp.dw_floats; ga.C = dwpart; ga.bias = nullptr; ga.M = Cout; ga.N = K9; ga.K = g.HoWo;
                ga.lda = g.HoWo; ga.ldb = g.HoWo; ga.ldc = K9;
                ga.strideA = (long long)Cout * g.HoWo; ga.strideB = (long long)K9 * g.HoWo;
                ga.strideC = (long long)sp.dw_split * n; ga.strideCs = n;
                ga.nsplit = sp.dw_split; ga.kchunk = sp.dw_kchunk; ga.ct = 0; ga.b_off = nullptr;
                // far samples dominate (far_dominated, count form): the sweep wrote no columns, the generic kernels produce grad_weight
                ga.skip_count = fs + 3; ga.skip_above = FAR_COUNT_PIVOT;
                if (split) sgemm_bf16x3(stream, true, true, ga, B, one);
                else sgemm_f32(stream, true, true, ga, B);
                e.nvp = 0; e.dw_bx = 0; e.dw_by = 0;
                e.gemm_n = n; e.gemm_S = B * sp.dw_split;
                nb_dw = (n / 4 + 255) / 256 < 512 ? (n / 4 + 255) / 256 : 512;
            }
            hipLaunchKernelGGL(dcn_sweep_epilogue, dim3(e.nb_coord + Cout * splits + nb_dw), dim3(256), 0, stream, e);
        }
        if (!handover) return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
        // lists + gather grad_input: only when the far samples dominate (each kernel checks the same device scalar)
        inv.packed = 0;
        inv.tileflag = nullptr;
        hipLaunchKernelGGL(dcn_offset_blockmax, bm_grid, dim3(64), 0, stream, offset, g, blockmax, fs);
        hipLaunchKernelGGL(dcn_build_inverse, dim3((H * W + 255) / 256, 9, B), dim3(256), 0, stream, offset, mask, inv, g, fs);
        {
            const int in_tiles = (H * W + 31) / 32;
            int mbi = nblk >= 8 ? 8 : nblk >= 4 ? 4 : nblk >= 2 ? 2 : 1;
            while (mbi > 1 && (int64_t)in_tiles * B * ((nblk + mbi - 1) / mbi) < 1024) mbi >>= 1;
            dim3 grid((in_tiles + 3) / 4, B, (nblk + mbi - 1) / mbi), block(256);
            if (mbi == 8) hipLaunchKernelGGL(dcn_bwd_input_f32<8>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, 0, fs);
            else if (mbi == 4) hipLaunchKernelGGL(dcn_bwd_input_f32<4>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, 0, fs);
            else if (mbi == 2) hipLaunchKernelGGL(dcn_bwd_input_f32<2>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, 0, fs);
            else hipLaunchKernelGGL(dcn_bwd_input_f32<1>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, 0, fs);
        }
        {
            dim3 grid((tiles + 3) / 4, B, nsplit), block(256);
#define DCD_LAUNCH_BD_FAR(NS)                                                                                            \
    hipLaunchKernelGGL(dcn_bwd_data_f32<NS>, grid, block, 0, stream, input, offset, mask, wb, grad_output, grad_input,  \
                       grad_offset, grad_mask, grad_bias, g, nsplit, inv, fs, (const int *)far_list, 1)
            if (g.Cop == 32) DCD_LAUNCH_BD_FAR(16);
            else if (g.Cop == 64) DCD_LAUNCH_BD_FAR(32);
            else if (g.Cop == 128) DCD_LAUNCH_BD_FAR(64);
            else if (g.Cop == 256) DCD_LAUNCH_BD_FAR(128);
            else DCD_LAUNCH_BD_FAR(0);
#undef DCD_LAUNCH_BD_FAR
        }
        {
            const int RB = 9 * nblk;
            const int nb = g.Cop / 32;
            const int mb = nb >= 8 ? 8 : nb >= 4 ? 4 : nb >= 2 ? 2 : 1;
            const int gx = (RB + 3) / 4, gz = (nb + mb - 1) / mb;
            const int total = B * tiles;
            int S = 512 / (gx * gz);
            if (S < 1) S = 1;
            if (S > total) S = total;
            dim3 grid(gx, S, gz), block(256);
            const size_t lds = (size_t)(4 * 32 * 33 + mb * 32 * 33) * sizeof(float);
#define DCD_LAUNCH_BW_FAR(MBV)                                                                                           \
    hipLaunchKernelGGL(dcn_bwd_weight_f32<MBV>, grid, block, lds, stream, input, offset, mask, grad_output, grad_weight, g, \
                       tiles, S, fs, (const int *)far_list)
            if (mb == 8) DCD_LAUNCH_BW_FAR(8);
            else if (mb == 4) DCD_LAUNCH_BW_FAR(4);
            else if (mb == 2) DCD_LAUNCH_BW_FAR(2);
            else DCD_LAUNCH_BW_FAR(1);
#undef DCD_LAUNCH_BW_FAR
        }
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    // split the channel blocks over grid.z when there are too few pixel tiles to fill the chip
    int nsplit = 1;
    while ((int64_t)tiles * B * nsplit < 1536 && nsplit * 2 <= nblk) nsplit *= 2;

    const bool tile_shape = kh == 3 && kw == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 && dg == 1 &&
                            (W & 3) == 0 && H >= 8 && W >= 32 && dcd_env("DCD_NO_TILE") == nullptr;
    // Cout <= 64: always.  Cout 128 (64 dY registers per lane): possible since the tap weights travel through LDS (no spills,
    // 191 VGPRs) but LDS then allows one workgroup per CU; A/B switch DCD_BD_TILE128=1.
    static int bd128 = -1;
    if (bd128 < 0) {
        const char *e = dcd_env("DCD_BD_TILE128");
        bd128 = (e && atoi(e) == 1) ? 1 : 0;
    }
    const bool bd_tile_ok = tile_shape && (g.Cop == 64 || (g.Cop == 128 && bd128));
    {
        ZeroRanges z;
        for (int r = 0; r < 8; ++r) { z.p[r] = nullptr; z.n[r] = 0; }
        z.p[0] = absmax; z.n[0] = 4;                         // max |offset| bits, far-tile count, overflowed-list count
        z.p[1] = (unsigned *)far_flag; z.n[1] = (unsigned)((ntile + 3) / 4);
        z.p[2] = (unsigned *)grad_weight; z.n[2] = (unsigned)((size_t)Cout * Cin * g.KK);
        z.p[3] = (unsigned *)grad_bias; z.n[3] = (unsigned)Cout;
        if (nsplit > 1 || (bd_tile_ok && nblk > 1)) {      // these accumulate with atomics when the channel blocks are split
            z.p[4] = (unsigned *)grad_offset; z.n[4] = (unsigned)((size_t)B * dg * 2 * g.KK * g.HoWo);
            z.p[5] = (unsigned *)grad_mask; z.n[5] = (unsigned)((size_t)B * dg * g.KK * g.HoWo);
        }
        z.p[6] = (unsigned *)tileflag; z.n[6] = (unsigned)(tileflag_bytes(g) / 4);
        const size_t most = (size_t)z.n[2] > (size_t)z.n[4] ? z.n[2] : z.n[4];
        hipLaunchKernelGGL(dcn_zero_ranges, dim3((unsigned)(most / 1024 + 1 < 2048 ? most / 1024 + 1 : 2048)), dim3(256), 0, stream, z);
    }

    // (1) search radius from max |offset|, (2) inverse sample lists, (3) grad_input by gather + MFMA (plain stores)
    {
        const int64_t noff = (int64_t)B * dg * 2 * g.KK * g.HoWo;
        int gsz = (int)((noff + 4095) / 4096);
        if (gsz > 512) gsz = 512;
        hipLaunchKernelGGL(dcn_offset_absmax, dim3(gsz), dim3(256), 0, stream, offset, noff, absmax, g.HoWo, dg * 2 * g.KK,
                           (g.HoWo + 31) / 32, far_flag, far_list);
        const int HWin = H * W;
#ifndef DCN_NO_BWD_TILE
        // tiled grad_input: one 64-channel slice per workgroup; wider inputs re-stage the dY window and re-read the lists per
        // slice and lose to the register-gather kernel (measured: 128->128 1.56 vs 1.48 ms, 256->256 2.25 vs 1.64 ms per backward)
        static int bi_nblk = 0;                                  // A/B: DCD_BI_TILE_NBLK=4 lets the tiled kernel take 128-channel inputs
        if (bi_nblk == 0) {
            const char *e = dcd_env("DCD_BI_TILE_NBLK");
            bi_nblk = e ? atoi(e) : 2;
            if (bi_nblk < 1) bi_nblk = 2;
        }
        const bool bi_tile_ok = tile_shape && nblk <= bi_nblk && (nblk <= 2 || g.Cop <= 64) && H < 65536 && W < 65536 &&
                                dcd_env("DCD_NO_BI_TILE") == nullptr;
#else
        const bool bi_tile_ok = false;
#endif
        inv.packed = bi_tile_ok ? 1 : 0;
        static int bi_hybrid = -1;                               // A/B: DCD_BI_HYBRID=0 -> one kernel per call, chosen by the call-wide radius
        if (bi_hybrid < 0) {
            const char *e = dcd_env("DCD_BI_HYBRID");
            bi_hybrid = (e && atoi(e) == 0) ? 0 : 1;
        }
        inv.tileflag = (bi_tile_ok && bi_hybrid) ? tileflag : nullptr;
        hipLaunchKernelGGL(dcn_offset_blockmax, bm_grid, dim3(64), 0, stream, offset, g, blockmax);
        hipLaunchKernelGGL(dcn_build_inverse, dim3((HWin + 255) / 256, dg * g.KK, B), dim3(256), 0, stream, offset, mask, inv, g);
        bool bi_tiled = false;
#ifndef DCN_NO_BWD_TILE
        if (bi_tile_ok) {
            static LdsLimit lds_limit;
            const size_t ldsb = (size_t)(BI_OC * BI_PLANE + (BI_OC / 2) * 9 * 2 * 2 * 32) * sizeof(float);
            if (!lds_limit.raise((int)ldsb, dcn_bwd_input_tile_f32<2>)) return DCD_ERR_LAUNCH;
            const int tiles_x = (W + 31) / 32, tiles_y = (H + BI_TR - 1) / BI_TR;
            hipLaunchKernelGGL(dcn_bwd_input_tile_f32<2>, dim3(tiles_x * tiles_y, B, (nblk + 1) / 2), dim3(BI_TR * 64), ldsb, stream,
                               grad_output, wb, inv, grad_input, g, tiles_x);
            bi_tiled = true;
        }
#endif
        const int in_tiles = (HWin + 31) / 32;
        const int total_blocks = dg * nblk;
        int mbi = total_blocks >= 8 ? 8 : total_blocks >= 4 ? 4 : total_blocks >= 2 ? 2 : 1;
        while (mbi > 1 && (int64_t)in_tiles * B * ((total_blocks + mbi - 1) / mbi) < 1024) mbi >>= 1;
        if (const char *e = dcd_env("DCD_BI_MB")) {              // A/B timing of the channel blocks per workgroup
            const int v = atoi(e);
            if ((v == 1 || v == 2 || v == 4 || v == 8) && v <= total_blocks) mbi = v;
        }
        dim3 grid((in_tiles + 3) / 4, B, (total_blocks + mbi - 1) / mbi), block(256);
        const int partner = bi_tiled ? 1 : 0;     // with the tiled kernel launched, this one only runs when the lists are too wide for it
        if (mbi == 8) hipLaunchKernelGGL(dcn_bwd_input_f32<8>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, partner);
        else if (mbi == 4) hipLaunchKernelGGL(dcn_bwd_input_f32<4>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, partner);
        else if (mbi == 2) hipLaunchKernelGGL(dcn_bwd_input_f32<2>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, partner);
        else hipLaunchKernelGGL(dcn_bwd_input_f32<1>, grid, block, 0, stream, grad_output, wb, inv, grad_input, g, partner);
    }
    {
        int splits = (int)(((int64_t)g.HoWo + 4095) / 4096);      // >= 4 float4 per thread and image
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
        hipLaunchKernelGGL(dcn_bias_grad, dim3(Cout, splits), dim3(256), 0, stream, grad_output, grad_bias, B, Cout, g.HoWo);
    }
    // (4) grad_offset / grad_mask (+ atomic fallback for what the lists do not cover)
    bool bd_tiled = false;
#ifndef DCN_NO_BWD_TILE
    if (bd_tile_ok) {
        static LdsLimit lds_limit;
        const size_t ldsb = (size_t)(BD_CB * BD_PLANE + 2 * g.Cop * 32) * sizeof(float);
        if (!lds_limit.raise((int)((BD_CB * BD_PLANE + 2 * 128 * 32) * sizeof(float)), dcn_bwd_data_tile_f32<32>, dcn_bwd_data_tile_f32<64>))
            return DCD_ERR_LAUNCH;
        const int tiles_x = (g.Wo + 31) / 32, tiles_y = (g.Ho + BD_TR - 1) / BD_TR;
        const int nsp = nblk;                 // one 32-channel block per workgroup
        dim3 gridt(tiles_x * tiles_y, B, nsp), blockt(BD_TR * 64);
        if (g.Cop == 64)
            hipLaunchKernelGGL(dcn_bwd_data_tile_f32<32>, gridt, blockt, ldsb, stream, input, offset, mask, wb, grad_output, grad_input,
                               grad_offset, grad_mask, g, tiles_x, nsp, inv, (const unsigned *)absmax);
        else
            hipLaunchKernelGGL(dcn_bwd_data_tile_f32<64>, gridt, blockt, ldsb, stream, input, offset, mask, wb, grad_output, grad_input,
                               grad_offset, grad_mask, g, tiles_x, nsp, inv, (const unsigned *)absmax);
        bd_tiled = true;
    }
#endif
    {
        dim3 grid((tiles + 3) / 4, B, nsplit), block(256);
        const int ns = g.Cop / 2;
#define DCD_LAUNCH_BD(NS)                                                                                          \
    hipLaunchKernelGGL(dcn_bwd_data_f32<NS>, grid, block, 0, stream, input, offset, mask, wb, grad_output, grad_input, \
                       grad_offset, grad_mask, grad_bias, g, nsplit, inv, bd_tiled ? (const unsigned *)absmax : (const unsigned *)nullptr, \
                       (const int *)far_list)
        if (ns == 16) DCD_LAUNCH_BD(16);
        else if (ns == 32) DCD_LAUNCH_BD(32);
        else if (ns == 64) DCD_LAUNCH_BD(64);
        else if (ns == 128) DCD_LAUNCH_BD(128);
        else DCD_LAUNCH_BD(0);
#undef DCD_LAUNCH_BD
    }
    bool dw_tiled = false;
#ifndef DCN_NO_BWD_TILE
    if (tile_shape) {
        static LdsLimit lds_limit, lds_limit2, lds_limit2b;
        const size_t ldsb = (size_t)(DW_IN_FLOATS + DW_DY_FLOATS) * sizeof(float);
        if (!lds_limit.raise((int)ldsb, dcn_bwd_weight_tile_f32) ||
            !lds_limit2.raise((int)(Dw2Cfg<DCD_PREC_F32>::FLOATS * sizeof(float)), dcn_bwd_weight_tile_v2<DCD_PREC_F32>) ||
            !lds_limit2b.raise((int)(Dw2Cfg<DCD_PREC_BF16X3>::FLOATS * sizeof(float)), dcn_bwd_weight_tile_v2<DCD_PREC_BF16X3>))
            return DCD_ERR_LAUNCH;
        static int dw_gen = 0;                                 // A/B: DCD_DW_GEN=1 keeps the first-generation f32 kernel
        if (dw_gen == 0) {
            const char *e = dcd_env("DCD_DW_GEN");
            dw_gen = (e && atoi(e) == 1) ? 1 : 2;
        }
        static int dw_colmajor = -1;                           // A/B: DCD_DW_ORDER=0 walks the tiles row by row
        if (dw_colmajor < 0) {
            const char *e = dcd_env("DCD_DW_ORDER");
            dw_colmajor = (e && atoi(e) == 0) ? 0 : 1;
        }
        const int tiles_x = (g.Wo + 31) / 32, tiles_y = (g.Ho + DW_TR - 1) / DW_TR;
        const int ncb = (Cin + DW_CB - 1) / DW_CB, nzo = (Cout + TL_OB - 1) / TL_OB;
        const int total = B * tiles_x * tiles_y;
        int S = 512 / (ncb * nzo);
        if (S < 1) S = 1;
        if (S > total) S = total;
        if (split)
            hipLaunchKernelGGL(dcn_bwd_weight_tile_v2<DCD_PREC_BF16X3>, dim3(ncb, S, nzo), dim3(DW_NT),
                               Dw2Cfg<DCD_PREC_BF16X3>::FLOATS * sizeof(float), stream, input, offset, mask, grad_output, dw_part, g,
                               tiles_x, tiles_y, S, (const unsigned *)absmax, dw_colmajor);
        else if (dw_gen == 2)
            hipLaunchKernelGGL(dcn_bwd_weight_tile_v2<DCD_PREC_F32>, dim3(ncb, S, nzo), dim3(DW_NT),
                               Dw2Cfg<DCD_PREC_F32>::FLOATS * sizeof(float), stream, input, offset, mask, grad_output, dw_part, g,
                               tiles_x, tiles_y, S, (const unsigned *)absmax, dw_colmajor);
        else
            hipLaunchKernelGGL(dcn_bwd_weight_tile_f32, dim3(ncb, S, nzo), dim3(DW_NT), ldsb, stream, input, offset, mask, grad_output,
                               dw_part, g, tiles_x, tiles_y, S, (const unsigned *)absmax);
        const int nred = ncb * nzo * 2 * 32 * DW_CB * 9;
        const int rgroups = S >= 16 ? 16 : S;
        hipLaunchKernelGGL(dcn_dw_reduce, dim3((nred + 255) / 256, rgroups), dim3(256), 0, stream, dw_part, grad_weight, g, ncb, nzo, S,
                           (const unsigned *)absmax);
        dw_tiled = true;
    }
#endif
    {
        const int RB = dg * g.KK * nblk;
        const int nb = g.Cop / 32;
        const int mb = nb >= 8 ? 8 : nb >= 4 ? 4 : nb >= 2 ? 2 : 1;
        const int gx = (RB + 3) / 4, gz = (nb + mb - 1) / mb;
        const int total = B * tiles;
        int S = 512 / (gx * gz);
        if (S < 1) S = 1;
        if (S > total) S = total;
        dim3 grid(gx, S, gz), block(256);
        const size_t lds = (size_t)(4 * 32 * 33 + mb * 32 * 33) * sizeof(float);
#define DCD_LAUNCH_BW(MBV)                                                                                        \
    hipLaunchKernelGGL(dcn_bwd_weight_f32<MBV>, grid, block, lds, stream, input, offset, mask, grad_output, grad_weight, g, \
                       tiles, S, dw_tiled ? (const unsigned *)absmax : (const unsigned *)nullptr, (const int *)far_list)
        if (mb == 8) DCD_LAUNCH_BW(8);
        else if (mb == 4) DCD_LAUNCH_BW(4);
        else if (mb == 2) DCD_LAUNCH_BW(2);
        else DCD_LAUNCH_BW(1);
#undef DCD_LAUNCH_BW
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

#ifdef SW_PROFILE
int dcd_debug_sweep_profile(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(sw_prof_buf), sizeof(unsigned long long) * 2048) == hipSuccess ? 0 : 1;
}
#endif

int dcd_dcn_v2_forget(const float *weight)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DCD_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_handover_mu);
    int n = 0;
    for (auto it = g_handover.begin(); it != g_handover.end();) {
        if (it->first.first == dev && (!weight || it->first.second == (const void *)weight)) {
            if (it->second.host) {
                *(volatile unsigned *)it->second.host = HANDOVER_UNKNOWN;
                g_handover_free[dev].push_back(it->second);
            }
            it = g_handover.erase(it);
            ++n;
        } else {
            ++it;
        }
    }
    (void)n;
    return DCD_OK;
}

int dcd_dcn_v2_policy_state(const float *weight, unsigned *far_count)
{
    int dev = 0;
    if (!weight || hipGetDevice(&dev) != hipSuccess) return -1;
    std::lock_guard<std::mutex> lock(g_handover_mu);
    auto it = g_handover.find(std::make_pair(dev, (const void *)weight));
    if (it == g_handover.end() || !it->second.host) return 0;
    const unsigned far = *(volatile unsigned *)it->second.host;
    if (far == HANDOVER_UNKNOWN) return 1;
    if (far_count) *far_count = far;
    return 2;
}

int dcd_dcn_v2_set_handover(int mode)
{
    if (mode < -1 || mode > 2) return DCD_ERR_BAD_ARG;
    g_handover_pin.store(mode, std::memory_order_relaxed);
    return DCD_OK;
}

int dcd_dcn_v2_policy_free(void)
{
    std::lock_guard<std::mutex> lock(g_handover_mu);
    for (auto &kv : g_handover)
        if (kv.second.host) (void)hipHostFree(kv.second.host);
    g_handover.clear();
    for (auto &kv : g_handover_free)
        for (HandoverState &h : kv.second)
            if (h.host) (void)hipHostFree(h.host);
    g_handover_free.clear();
    (void)hipGetLastError();
    return DCD_OK;
}

int dcd_dcn_offset_mask_split(void *stream_, const float *out, float *offset, float *mask, int B, int taps, int64_t HW)
{
    (void)hipGetLastError();
    if (!out || !offset || !mask || B <= 0 || taps <= 0 || HW <= 0) return DCD_ERR_BAD_ARG;
    const bool v4 = (HW & 3) == 0 && (((uintptr_t)out | (uintptr_t)offset | (uintptr_t)mask) & 15) == 0;
    const int64_t n = (int64_t)B * 3 * taps * HW / (v4 ? 4 : 1);
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (v4) hipLaunchKernelGGL(dcn_offset_mask_split<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, out, offset, mask, B, taps, HW);
    else hipLaunchKernelGGL(dcn_offset_mask_split<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, out, offset, mask, B, taps, HW);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_dcn_offset_mask_merge(void *stream_, const float *grad_offset, const float *grad_mask, const float *mask, float *grad_out, int B,
                              int taps, int64_t HW)
{
    (void)hipGetLastError();
    if (!grad_offset || !grad_mask || !mask || !grad_out || B <= 0 || taps <= 0 || HW <= 0) return DCD_ERR_BAD_ARG;
    const bool v4 = (HW & 3) == 0 &&
                    (((uintptr_t)grad_offset | (uintptr_t)grad_mask | (uintptr_t)mask | (uintptr_t)grad_out) & 15) == 0;
    const int64_t n = (int64_t)B * 3 * taps * HW / (v4 ? 4 : 1);
    const unsigned grid = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (v4)
        hipLaunchKernelGGL(dcn_offset_mask_merge<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, grad_offset, grad_mask, mask, grad_out,
                           B, taps, HW);
    else
        hipLaunchKernelGGL(dcn_offset_mask_merge<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream_, grad_offset, grad_mask, mask, grad_out,
                           B, taps, HW);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_sgemm_shifted(void *stream_, const float *A, int lda, long long strideA, const float *Bbase, const long long *b_off,
                      long long strideB, int b_kcontig, const float *bias, float *C, int ldc, long long strideC,
                      long long strideCs, int M, int N, int K, int Z, int nsplit)
{
    (void)hipGetLastError();
    if (!A || !Bbase || !b_off || !C || M <= 0 || N <= 0 || K <= 0 || Z <= 0 || nsplit <= 0 || (lda & 3) || ((uintptr_t)A & 15) ||
        (strideA & 3) || (bias && nsplit != 1))
        return DCD_ERR_BAD_ARG;
    SgemmArgs a;
    a.A = A; a.B = Bbase; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = 0; a.ldc = ldc;
    a.strideA = strideA; a.strideB = strideB; a.strideC = strideC; a.strideCs = strideCs;
    a.nsplit = nsplit; a.kchunk = ((K + nsplit - 1) / nsplit + SG_K - 1) / SG_K * SG_K; a.ct = 0; a.b_off = b_off;
    if ((long long)a.kchunk * (nsplit - 1) >= K) return DCD_ERR_BAD_ARG;      // an empty split would leave its partial unwritten
    sgemm_f32_rows64((hipStream_t)stream_, b_kcontig != 0, a, Z);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
