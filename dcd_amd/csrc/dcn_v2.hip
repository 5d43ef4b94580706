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
//  * grad_input is scattered with hardware fp32 atomics (global_atomic_add_f32), like the
//    reference's col2im; grad_weight / grad_bias are reduced across pixel splits with atomics.
//
// Sample validity follows the reference exactly: a tap contributes iff -1 < h < H and -1 < w < W
// (cuda/dcn_v2_im2col_cuda.cu:180), each corner iff it lies inside the image (:38-48).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

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
__global__ void dcn_prep_weights(const float *__restrict__ w, float *__restrict__ wf, float *__restrict__ wb, Geom g)
{
    const int n = g.Kp * g.Cop;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
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

// Per-(pixel, tap) sampling state shared by all channels of a deformable group.
struct Tap {
    int i1, i2, i3, i4;      // corner element offsets inside one H*W plane (0 when the corner is unused)
    float w1, w2, w3, w4;    // bilinear weights, zero for corners outside the image / invalid samples
    float lh, lw, hh, hw;    // fractional parts (for the coordinate gradient)
    float m;                 // modulation mask (0 when the sample is invalid or the pixel is padding)
    bool c1, c2, c3, c4;     // corner validity
};

__device__ __forceinline__ Tap make_tap(const float *__restrict__ off_b, const float *__restrict__ msk_b, const Geom &g,
                                        int seg, int t, int ho, int wo, int Pc, bool pv)
{
    Tap s;
    const int i = t / g.kw, j = t - i * g.kw;
    const float oh = off_b[(size_t)(2 * seg) * g.HoWo + Pc];
    const float ow = off_b[(size_t)(2 * seg + 1) * g.HoWo + Pc];
    const float m = msk_b[(size_t)seg * g.HoWo + Pc];
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
    return s;
}

// Uniform base + 32-bit per-lane BYTE offset -> `global_load_dword v, v_off, s[base:base+1]`.
__device__ __forceinline__ float ldg(const float *base, unsigned byte_off)
{
    return *(const float *)((const char *)base + byte_off);
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
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 32 >= g.HoWo) return;  // wave-uniform
    const int b = blockIdx.y;
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
        auto step = [&](const float *ipc, const float *wpc) {
            const float v1 = ldg(ipc, o1), v2 = ldg(ipc, o2), v3 = ldg(ipc, o3), v4 = ldg(ipc, o4);
            float a[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[mb] = ldg(wpc, wlane + mb * 128u);
            const float val = (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * s.m;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], val, acc[mb], 0, 0, 0);
        };
        constexpr int UN = (MB >= 8) ? 2 : 4;
        int it = 0;
        for (; it + UN <= npair; it += UN) {
#pragma unroll
            for (int u = 0; u < UN; ++u)
                step((const float *)((const char *)ip + (size_t)(it + u) * 2 * HW4),
                     (const float *)((const char *)wp + (size_t)(it + u) * 2 * Cop4));
        }
        for (; it < npair; ++it)
            step((const float *)((const char *)ip + (size_t)it * 2 * HW4),
                 (const float *)((const char *)wp + (size_t)it * 2 * Cop4));
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

// ---------------------------------------------------------------------------------------------
// Backward w.r.t. input, offset, mask (+ bias).
// grid = (ceil(tiles/4), B, nsplit); each z handles a contiguous range of 32-channel blocks.
// NS = Cop/2 register-cached dY values per lane (0: stream dY from memory, any Cout).
// ---------------------------------------------------------------------------------------------
template <int NS>
__global__ __launch_bounds__(256) void dcn_bwd_data_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                        const float *__restrict__ msk, const float *__restrict__ wb,
                                                        const float *__restrict__ gy, float *__restrict__ gin,
                                                        float *__restrict__ goff, float *__restrict__ gmsk,
                                                        float *__restrict__ gbias, Geom g, int nsplit)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile * 32 >= g.HoWo) return;
    const int b = blockIdx.y, z = blockIdx.z;
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

    // grad_bias: sum of dY over the 32 pixels of this tile, one atomic per (tile, o)
    if (z == 0) {
        if (NS > 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float v = dy[s];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (p == 0 && 2 * s + h < g.Co) atomicAdd(gbias + 2 * s + h, v);
            }
        } else {
            for (int s = 0; s < nsteps; ++s) {
                const int o = 2 * s + h;
                float v = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 1);
                if (p == 0 && o < g.Co) atomicAdd(gbias + o, v);
            }
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
            const int seg = grp * g.KK + t;
            const Tap s = make_tap(off_b, msk_b, g, seg, t, ho, wo, Pc, pv);
            float s_m = 0.f, s_h = 0.f, s_w = 0.f;
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
                for (int r = 0; r < 16; ++r) {
                    const int cc = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (cc < g.cpg) {
                        const float *ip = in_g + (size_t)cc * HW;
                        const float v1 = s.c1 ? ip[s.i1] : 0.f, v2 = s.c2 ? ip[s.i2] : 0.f;
                        const float v3 = s.c3 ? ip[s.i3] : 0.f, v4 = s.c4 ? ip[s.i4] : 0.f;
                        const float d = acc[r];
                        s_m += d * (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4);
                        const float dm = d * s.m;
                        // d/dh and d/dw of the bilinear sample (cuda/dcn_v2_im2col_cuda.cu:82-123)
                        s_h += dm * (s.hw * (v3 - v1) + s.lw * (v4 - v2));
                        s_w += dm * (s.hh * (v2 - v1) + s.lh * (v4 - v3));
                        float *gp = gin_g + (size_t)cc * HW;
                        if (s.c1) atomicAdd(gp + s.i1, dm * s.w1);
                        if (s.c2) atomicAdd(gp + s.i2, dm * s.w2);
                        if (s.c3) atomicAdd(gp + s.i3, dm * s.w3);
                        if (s.c4) atomicAdd(gp + s.i4, dm * s.w4);
                    }
                }
            }
            s_m += __shfl_xor(s_m, 32);
            s_h += __shfl_xor(s_h, 32);
            s_w += __shfl_xor(s_w, 32);
            if (h == 0 && pv) {
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
// Backward w.r.t. weight.  D[k', o] = sum_p col[k', p] dY[o, p].
// grid = (ceil(RB/4), S, ceil(Cop/32/MB)); block = 256: wave w owns row-block 4*blockIdx.x + w
// (32 k' rows = 32 channels of one (group, tap)) and MB 32-wide output-channel blocks.
// Each block walks its share of (image, 32-pixel tile) pairs; per tile the sampled columns and
// the dY tile are transposed through LDS (row stride 33 -> conflict-free ds_read_b32).
// ---------------------------------------------------------------------------------------------
template <int MB>
__global__ __launch_bounds__(256) void dcn_bwd_weight_f32(const float *__restrict__ in, const float *__restrict__ off,
                                                          const float *__restrict__ msk, const float *__restrict__ gy,
                                                          float *__restrict__ gw, Geom g, int tiles_per_img, int nsplit)
{
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
    const int seg = rbc / nblk, blk = rbc - seg * nblk;
    const int grp = seg / g.KK, t = seg - grp * g.KK;
    const int ob0 = blockIdx.z * MB;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

    const int total = g.B * tiles_per_img;
    const int t0 = (int)((int64_t)blockIdx.y * total / nsplit), t1 = (int)((int64_t)(blockIdx.y + 1) * total / nsplit);
    float *myT = colT + wave * 32 * 33;

    for (int ti = t0; ti < t1; ++ti) {
        const int b = ti / tiles_per_img, tile = ti - b * tiles_per_img;
        const int P = tile * 32 + p;
        const bool pv = P < g.HoWo;
        const int Pc = pv ? P : g.HoWo - 1;
        const int ho = Pc / g.Wo, wo = Pc - ho * g.Wo;
        const float *in_g = in + ((size_t)b * g.C + (size_t)grp * g.cpg) * HW;
        const float *off_b = off + (size_t)b * g.dg * 2 * g.KK * g.HoWo;
        const float *msk_b = msk + (size_t)b * g.dg * g.KK * g.HoWo;
        const float *gy_b = gy + (size_t)b * g.Co * g.HoWo;

        // (1) sampled columns for this wave's 32 channels -> colT[row][pixel]
        const Tap s = make_tap(off_b, msk_b, g, seg, t, ho, wo, Pc, pv && rbv);
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int row = 2 * q + h;
            const int cc = blk * 32 + row;
            const bool cv = cc < g.cpg;
            const float *ip = in_g + (size_t)(cv ? cc : 0) * HW;
            const float v1 = ip[s.i1], v2 = ip[s.i2], v3 = ip[s.i3], v4 = ip[s.i4];
            const float val = (s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4) * s.m;
            myT[row * 33 + p] = cv ? val : 0.f;
        }
        // (2) dY tile -> dyT[o][pixel]; the four waves split the rows
#pragma unroll
        for (int q = 0; q < MB * 4; ++q) {
            const int ol = q * 8 + wave * 2 + h;  // 0 .. MB*32-1
            const int o = ob0 * 32 + ol;
            dyT[ol * 33 + p] = (pv && o < g.Co) ? gy_b[(size_t)o * g.HoWo + P] : 0.f;
        }
        __syncthreads();
        // (3) contraction over the 32 pixels
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const float a = myT[p * 33 + 2 * k + h];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dyT[(mb * 32 + p) * 33 + 2 * k + h], acc[mb], 0, 0, 0);
        }
        __syncthreads();
    }

    if (!rbv) return;
    // lane holds D[row = (r&3)+8*(r>>2)+4h][o = ob*32 + p]
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int o = (ob0 + mb) * 32 + p;
        if (o >= g.Co) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cc = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cc < g.cpg) atomicAdd(gw + ((size_t)o * g.C + grp * g.cpg + cc) * g.KK + t, acc[mb][r]);
        }
    }
}

inline int pick_mb(int nb, int tiles_total)
{
    // largest MB in {8,4,2,1} (32-wide Cout blocks per wave) that still leaves >= 1024 waves
    for (int mb = 8; mb > 1; mb >>= 1) {
        if (mb > nb) continue;
        if ((int64_t)tiles_total * ((nb + mb - 1) / mb) >= 1024) return mb;
    }
    return 1;
}

}  // namespace

extern "C" {

size_t dcd_dcn_v2_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph,
                                  int pw, int dh, int dw, int dg)
{
    Geom g;
    if (!make_geom(g, B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return 0;
    return (size_t)g.Kp * g.Cop * sizeof(float) * 2 + 256;
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
    if (precision != DCD_PREC_F32) return DCD_ERR_BAD_ARG;
    const size_t nw = (size_t)g.Kp * g.Cop;
    if (workspace_bytes < nw * sizeof(float) * 2) return DCD_ERR_WORKSPACE;
    float *wf = (float *)workspace, *wb = wf + nw;

    hipLaunchKernelGGL(dcn_prep_weights, dim3((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048)), dim3(256),
                       0, stream, weight, wf, wb, g);

    const int tiles = (g.HoWo + 31) / 32;
    const int nb = g.Cop / 32;
    const int mb = pick_mb(nb, tiles * B);
    dim3 grid((tiles + 3) / 4, B, (nb + mb - 1) / mb), block(256);
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
    if (precision != DCD_PREC_F32) return DCD_ERR_BAD_ARG;
    const size_t nw = (size_t)g.Kp * g.Cop;
    if (workspace_bytes < nw * sizeof(float) * 2) return DCD_ERR_WORKSPACE;
    float *wf = (float *)workspace, *wb = wf + nw;

    hipLaunchKernelGGL(dcn_prep_weights, dim3((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048)), dim3(256),
                       0, stream, weight, wf, wb, g);

    const int tiles = (g.HoWo + 31) / 32;
    const int nblk = g.cpgp / 32;
    // split the channel blocks over grid.z when there are too few pixel tiles to fill the chip
    int nsplit = 1;
    while ((int64_t)tiles * B * nsplit < 1536 && nsplit * 2 <= nblk) nsplit *= 2;

    hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)B * Cin * H * W, stream);
    hipMemsetAsync(grad_weight, 0, sizeof(float) * (size_t)Cout * Cin * g.KK, stream);
    hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)Cout, stream);
    if (nsplit > 1) {
        hipMemsetAsync(grad_offset, 0, sizeof(float) * (size_t)B * dg * 2 * g.KK * g.HoWo, stream);
        hipMemsetAsync(grad_mask, 0, sizeof(float) * (size_t)B * dg * g.KK * g.HoWo, stream);
    }

    {
        dim3 grid((tiles + 3) / 4, B, nsplit), block(256);
        const int ns = g.Cop / 2;
#define DCD_LAUNCH_BD(NS)                                                                                          \
    hipLaunchKernelGGL(dcn_bwd_data_f32<NS>, grid, block, 0, stream, input, offset, mask, wb, grad_output, grad_input, \
                       grad_offset, grad_mask, grad_bias, g, nsplit)
        if (ns == 16) DCD_LAUNCH_BD(16);
        else if (ns == 32) DCD_LAUNCH_BD(32);
        else if (ns == 64) DCD_LAUNCH_BD(64);
        else if (ns == 128) DCD_LAUNCH_BD(128);
        else DCD_LAUNCH_BD(0);
#undef DCD_LAUNCH_BD
    }
    {
        const int RB = dg * g.KK * nblk;
        const int nb = g.Cop / 32;
        const int mb = nb >= 8 ? 8 : nb >= 4 ? 4 : nb >= 2 ? 2 : 1;
        const int gx = (RB + 3) / 4, gz = (nb + mb - 1) / mb;
        const int total = B * tiles;
        int S = 1024 / (gx * gz);
        if (S < 1) S = 1;
        if (S > total) S = total;
        dim3 grid(gx, S, gz), block(256);
        const size_t lds = (size_t)(4 * 32 * 33 + mb * 32 * 33) * sizeof(float);
#define DCD_LAUNCH_BW(MBV)                                                                                        \
    hipLaunchKernelGGL(dcn_bwd_weight_f32<MBV>, grid, block, lds, stream, input, offset, mask, grad_output, grad_weight, g, \
                       tiles, S)
        if (mb == 8) DCD_LAUNCH_BW(8);
        else if (mb == 4) DCD_LAUNCH_BW(4);
        else if (mb == 2) DCD_LAUNCH_BW(2);
        else DCD_LAUNCH_BW(1);
#undef DCD_LAUNCH_BW
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
