// heads.hip -- loss / decode / edge-depth kernels of the DGDE hot path for gfx950.
//
// These ops are HBM-trivial and launch-latency bound in the reference (tens of thousands of tiny
// PyTorch kernels per step); each becomes ONE launch here.
//   edge depth : one workgroup per object, keypoints in LDS, 2628 pairs, bitonic top-k in LDS
//   focal loss : grid-stride fused loss + gradient, wave shuffle reduction, one atomic per block
//   GIoU       : one lane per box, closed-form gradient
//   heat map   : 3x3 NMS + radix-select top-K, one workgroup per (image, class)
//   POI gather : direct strided gather from NCHW (no NHWC copy)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"
#include "zero_fill.h"

namespace {

constexpr int EDGE_MAXK = 128;
constexpr int EDGE_THREADS = 512;

// pair index q (row-major upper triangle, i<j) -> (i, j)
__device__ __forceinline__ void pair_from_index(int q, int K, int &i, int &j)
{
    // row i starts at S(i) = i*(2K-i-1)/2
    float kf = (float)(2 * K - 1);
    int ii = (int)floorf((kf - sqrtf(kf * kf - 8.f * (float)q)) * 0.5f);
    if (ii < 0) ii = 0;
    if (ii > K - 2) ii = K - 2;
    while (ii > 0 && ii * (2 * K - ii - 1) / 2 > q) --ii;
    while ((ii + 1) * (2 * K - ii - 2) / 2 <= q) ++ii;
    i = ii;
    j = q - ii * (2 * K - ii - 1) / 2 + ii + 1;
}

// value of lane ^ M (M = 1 .. 32) without the LDS crossbar
template <int M>
__device__ __forceinline__ unsigned xor_lane(unsigned v, int lane)
{
    if constexpr (M == 1) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);            // quad_perm [1,0,3,2]
    } else if constexpr (M == 2) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);            // quad_perm [2,3,0,1]
    } else if constexpr (M == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xF, 0xF, false);               // quad_perm [3,2,1,0]: i ^ 3
        return (unsigned)__builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, false);                // row_half_mirror: 7 - i   -> i ^ 4
    } else if constexpr (M == 8) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);              // row_mirror: 15 - i
        return (unsigned)__builtin_amdgcn_update_dpp(0, t, 0x141, 0xF, 0xF, false);                // row_half_mirror          -> i ^ 8
    } else if constexpr (M == 16) {
        // v_permlane16_swap: the odd rows of the first operand <-> the even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        // v_permlane32_swap: the upper half of the first operand <-> the lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

constexpr int EDGE_E = 8;                 // keys per thread of the register form of the sort (EDGE_E * EDGE_THREADS = 4 096 padded pairs)

template <int M>
__device__ __forceinline__ void edge_lane_stage(unsigned long long (&key)[EDGE_E], bool take_max, int tid)
{
#pragma unroll
    for (int r = 0; r < EDGE_E; ++r) {
        const unsigned lo = xor_lane<M>((unsigned)key[r], tid), hi = xor_lane<M>((unsigned)(key[r] >> 32), tid);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        key[r] = ((key[r] > other) == take_max) ? key[r] : other;
    }
}

// ---------------------------------------------------------------------------------------------
// Edge-constraint depth solve, forward.  DGDE/model/anno_encoder.py:326-390, GMW/main.py:373-416.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EDGE_THREADS) void edge_depth_fwd(const float *__restrict__ kps, const float *__restrict__ kps3d,
                                                               const float *__restrict__ rot_y, const float *__restrict__ Pm,
                                                               const uint8_t *__restrict__ kmask, int K, int topk, float zmin,
                                                               float zmax, int normalized, int sub_b3, float *__restrict__ depth,
                                                               int32_t *__restrict__ pair_idx, float *__restrict__ pair_mask)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int npairs = K * (K - 1) / 2;
    int npad = 1;
    while (npad < npairs) npad <<= 1;

    float *sv = (float *)smem_raw;            // v' per keypoint
    float *sh1 = sv + EDGE_MAXK;              // Y
    float *sh2 = sh1 + EDGE_MAXK;             // v' * C
    float *sz = sh2 + EDGE_MAXK;              // z per pair          [npairs]
    unsigned long long *skey = (unsigned long long *)(sz + ((npairs + 1) & ~1));  // sort keys [npad] (topk only)

    const float *Pn = Pm + (size_t)n * 12;
    const float fy = Pn[5], cy = Pn[6], b3 = Pn[11];
    const float rot = rot_y[n];
    const float sn = sinf(rot), cs = cosf(rot);

    for (int k = tid; k < K; k += EDGE_THREADS) {
        const float v = kps[((size_t)n * K + k) * 2 + 1];
        const float vn = normalized ? v : __fdiv_rn(__fsub_rn(v, cy), fy);
        const float X = kps3d[((size_t)n * K + k) * 3 + 0];
        const float Y = kps3d[((size_t)n * K + k) * 3 + 1];
        const float Z = kps3d[((size_t)n * K + k) * 3 + 2];
        const float C = __fsub_rn(__fmul_rn(X, sn), __fmul_rn(Z, cs));
        sv[k] = vn;
        sh1[k] = Y;
        sh2[k] = __fmul_rn(vn, C);
    }
    __syncthreads();

    for (int q = tid; q < (topk ? npad : npairs); q += EDGE_THREADS) {
        if (q < npairs) {
            int i, j;
            pair_from_index(q, K, i, j);
            const float hm = __fadd_rn(__fsub_rn(sh1[i], sh1[j]), __fsub_rn(sh2[i], sh2[j]));
            const float dv = fabsf(__fsub_rn(sv[i], sv[j]));
            float z = __fdiv_rn(fabsf(hm), fmaxf(dv, 1e-10f));
            z = fminf(fmaxf(z, zmin), zmax);
            sz[q] = z;
            if (topk) skey[q] = ((unsigned long long)__float_as_uint(dv) << 32) | (unsigned)(~(unsigned)q);
        } else {
            skey[q] = 0ull;
        }
    }
    __syncthreads();

    const float sub = sub_b3 ? b3 : 0.f;
    if (!topk) {
        for (int q = tid; q < npairs; q += EDGE_THREADS) depth[(size_t)n * npairs + q] = sz[q] - sub;
        return;
    }

    // bitonic sort, descending by (|dv|, lower pair index first).  4 096 keys (the reference's 73 keypoints: 2 628 pairs) on 512
    // threads: a thread owns the eight consecutive positions 8 tid .. 8 tid + 7 in registers, so the strides 1 / 2 / 4 are register
    // compare-exchanges, 8 .. 256 are lane exchanges inside the wave, and only 512 / 1 024 / 2 048 (6 of the 78 stages) go through
    // LDS and a barrier -- as 78 LDS passes with a barrier each the launch took 113 us whatever the batch.  The keys are distinct (the
    // pair index is their low half), so every correct network gives the same order.
    if (npad == EDGE_E * EDGE_THREADS) {
        unsigned long long key[EDGE_E];
        __syncthreads();                                    // (skey is rewritten below as the exchange buffer)
#pragma unroll
        for (int r = 0; r < EDGE_E; ++r) key[r] = skey[tid * EDGE_E + r];
        __syncthreads();
        for (int size = 2; size <= npad; size <<= 1) {
            const bool desc = ((tid * EDGE_E) & size) == 0;     // (for the strides >= EDGE_E: size >= 2 EDGE_E)
            for (int stride = size >> 1; stride >= EDGE_E * 64; stride >>= 1) {
                const int m = stride / EDGE_E;                  // partner thread = tid ^ m (another wave), same register
                const bool take_max = (((tid * EDGE_E) & stride) == 0) == desc;
                unsigned long long other[EDGE_E];
#pragma unroll
                for (int r = 0; r < EDGE_E; ++r) skey[r * EDGE_THREADS + tid] = key[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < EDGE_E; ++r) other[r] = skey[r * EDGE_THREADS + (tid ^ m)];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < EDGE_E; ++r) key[r] = ((key[r] > other[r]) == take_max) ? key[r] : other[r];
            }
            // partner lane = lane ^ M: data-parallel-primitive moves and the gfx950 row / half swaps, all on the vector ALU
            // (as ds_bpermute_b32 -- what __shfl_xor compiles to -- the 624 exchanges of a wave were 34 of the launch's 41 us)
            if (32 * EDGE_E < size) edge_lane_stage<32>(key, (((tid * EDGE_E) & (32 * EDGE_E)) == 0) == desc, tid);
            if (16 * EDGE_E < size) edge_lane_stage<16>(key, (((tid * EDGE_E) & (16 * EDGE_E)) == 0) == desc, tid);
            if (8 * EDGE_E < size) edge_lane_stage<8>(key, (((tid * EDGE_E) & (8 * EDGE_E)) == 0) == desc, tid);
            if (4 * EDGE_E < size) edge_lane_stage<4>(key, (((tid * EDGE_E) & (4 * EDGE_E)) == 0) == desc, tid);
            if (2 * EDGE_E < size) edge_lane_stage<2>(key, (((tid * EDGE_E) & (2 * EDGE_E)) == 0) == desc, tid);
            if (1 * EDGE_E < size) edge_lane_stage<1>(key, (((tid * EDGE_E) & (1 * EDGE_E)) == 0) == desc, tid);
#pragma unroll
            for (int st = EDGE_E / 2; st >= 1; st >>= 1) {
                if (st < size) {
#pragma unroll
                    for (int r = 0; r < EDGE_E; ++r) {
                        if ((r & st) == 0) {
                            const bool desc = ((tid * EDGE_E + r) & size) == 0;
                            const unsigned long long a = key[r], b = key[r | st];
                            const bool sw = (a < b) == desc;
                            key[r] = sw ? b : a;
                            key[r | st] = sw ? a : b;
                        }
                    }
                }
            }
        }
        // back to LDS in rank order: the output loop below walks the ranks with consecutive threads (coalesced stores, three
        // dependent mask loads per thread instead of eight)
#pragma unroll
        for (int e = 0; e < EDGE_E; ++e) skey[tid * EDGE_E + e] = key[e];
        __syncthreads();
    } else {
        for (int size = 2; size <= npad; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (npad >> 1); t += EDGE_THREADS) {
                    const int lo = ((t / stride) * stride * 2) + (t % stride);
                    const int hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const unsigned long long a = skey[lo], b = skey[hi];
                    if ((a < b) == desc) {
                        skey[lo] = b;
                        skey[hi] = a;
                    }
                }
                __syncthreads();
            }
        }
    }

    for (int r = tid; r < topk; r += EDGE_THREADS) {
        const unsigned q = ~(unsigned)(skey[r] & 0xffffffffull);
        depth[(size_t)n * topk + r] = sz[q] - sub;
        pair_idx[(size_t)n * topk + r] = (int32_t)q;
        if (pair_mask) {
            int i, j;
            pair_from_index((int)q, K, i, j);
            const float mi = kmask ? (kmask[(size_t)n * K + i] ? 1.f : 0.f) : 1.f;
            const float mj = kmask ? (kmask[(size_t)n * K + j] ? 1.f : 0.f) : 1.f;
            pair_mask[(size_t)n * topk + r] = mi * mj;
        }
    }
}

// Backward: d depth / d (v, X, Y, Z).  Autograd semantics of abs / clamp_min / clamp / gather.
__global__ __launch_bounds__(256) void edge_depth_bwd(const float *__restrict__ kps, const float *__restrict__ kps3d,
                                                      const float *__restrict__ rot_y, const float *__restrict__ Pm,
                                                      const float *__restrict__ gdepth, const int32_t *__restrict__ pair_idx,
                                                      int K, int topk, float zmin, float zmax, int normalized,
                                                      float *__restrict__ gkps, float *__restrict__ gkps3d)
{
    __shared__ float sv[EDGE_MAXK], sy[EDGE_MAXK], sc[EDGE_MAXK];
    __shared__ float gv[EDGE_MAXK], gy[EDGE_MAXK], gc[EDGE_MAXK];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int npairs = K * (K - 1) / 2;
    const int M = topk ? topk : npairs;
    const float *Pn = Pm + (size_t)n * 12;
    const float fy = Pn[5], cy = Pn[6];
    const float rot = rot_y[n];
    const float sn = sinf(rot), cs = cosf(rot);
    for (int k = tid; k < K; k += 256) {
        const float v = kps[((size_t)n * K + k) * 2 + 1];
        sv[k] = normalized ? v : (v - cy) / fy;
        const float X = kps3d[((size_t)n * K + k) * 3 + 0];
        sy[k] = kps3d[((size_t)n * K + k) * 3 + 1];
        const float Z = kps3d[((size_t)n * K + k) * 3 + 2];
        sc[k] = X * sn - Z * cs;
        gv[k] = gy[k] = gc[k] = 0.f;
    }
    __syncthreads();
    for (int r = tid; r < M; r += 256) {
        const float g = gdepth[(size_t)n * M + r];
        if (g == 0.f) continue;
        const int q = topk ? pair_idx[(size_t)n * M + r] : r;
        int i, j;
        pair_from_index(q, K, i, j);
        const float A = (sy[i] - sy[j]) + (sv[i] * sc[i] - sv[j] * sc[j]);
        const float D = sv[i] - sv[j];
        const float aD = fabsf(D), den = fmaxf(aD, 1e-10f);
        const float z = fabsf(A) / den;
        if (z < zmin || z > zmax) continue;  // clamp saturated: no gradient
        const float sA = (A > 0.f) ? 1.f : (A < 0.f ? -1.f : 0.f);
        const float gA = g * sA / den;
        float gD = 0.f;
        if (aD >= 1e-10f) {
            const float sD = (D > 0.f) ? 1.f : (D < 0.f ? -1.f : 0.f);
            gD = -g * fabsf(A) / (den * den) * sD;
        }
        atomicAdd(&gy[i], gA);
        atomicAdd(&gy[j], -gA);
        atomicAdd(&gv[i], gA * sc[i] + gD);
        atomicAdd(&gv[j], -gA * sc[j] - gD);
        atomicAdd(&gc[i], gA * sv[i]);
        atomicAdd(&gc[j], -gA * sv[j]);
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        gkps[((size_t)n * K + k) * 2 + 0] = 0.f;
        gkps[((size_t)n * K + k) * 2 + 1] = normalized ? gv[k] : gv[k] / fy;
        gkps3d[((size_t)n * K + k) * 3 + 0] = gc[k] * sn;
        gkps3d[((size_t)n * K + k) * 3 + 1] = gy[k];
        gkps3d[((size_t)n * K + k) * 3 + 2] = -gc[k] * cs;
    }
}

// ---------------------------------------------------------------------------------------------
// Penalty-reduced focal loss.  DGDE/model/layers/focal_loss.py:57-86.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float powi_or_f(float x, float e)
{
    if (e == 2.f) return x * x;
    if (e == 4.f) { const float x2 = x * x; return x2 * x2; }
    if (e == 1.f) return x;
    return powf(x, e);
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void focal_loss_kernel(const float *__restrict__ pred, const float *__restrict__ target,
                                                         int64_t n, float alpha, float beta, float *__restrict__ out,
                                                         float *__restrict__ gpred)
{
    float loss = 0.f, npos = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float praw = pred[i], t = target[i];
        const float p = fminf(fmaxf(praw, 1e-10f), 1.f - 1e-10f);
        const bool inside = (praw >= 1e-10f) && (praw <= 1.f - 1e-10f);  // clamp passes gradient inside only
        float l = 0.f, g = 0.f;
        if (t == 1.f) {
            const float omp = 1.f - p, lg = logf(p);
            l = -lg * powi_or_f(omp, alpha);
            g = -powi_or_f(omp, alpha) / p + alpha * lg * powi_or_f(omp, alpha - 1.f);
            npos += 1.f;
        } else if (t < 1.f && t >= 0.f) {
            const float w = powi_or_f(1.f - t, beta), lg = logf(1.f - p);
            l = -lg * powi_or_f(p, alpha) * w;
            g = w * (powi_or_f(p, alpha) / (1.f - p) - alpha * lg * powi_or_f(p, alpha - 1.f));
        }
        loss += l;
        if (gpred) gpred[i] = inside ? g : 0.f;
    }
    loss = wave_sum(loss);
    npos = wave_sum(npos);
    __shared__ float sl[4], sp[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sl[wave] = loss; sp[wave] = npos; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out + 0, sl[0] + sl[1] + sl[2] + sl[3]);
        atomicAdd(out + 1, sp[0] + sp[1] + sp[2] + sp[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// GIoU on (l,t,r,b).  DGDE/model/layers/iou_loss.py:12-49.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void min_grad(float a, float b, float g, float &ga, float &gb)
{   // torch.min(a, b) backward: the smaller gets g, ties split evenly
    if (a < b) ga += g; else if (b < a) gb += g; else { ga += 0.5f * g; gb += 0.5f * g; }
}
__device__ __forceinline__ void max_grad(float a, float b, float g, float &ga, float &gb)
{
    if (a > b) ga += g; else if (b > a) gb += g; else { ga += 0.5f * g; gb += 0.5f * g; }
}

__global__ void giou_kernel(const float *__restrict__ pred, const float *__restrict__ target, int N, float *__restrict__ losses,
                            float *__restrict__ ious, float *__restrict__ gpred)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float pl = pred[i * 4 + 0], pt = pred[i * 4 + 1], pr = pred[i * 4 + 2], pb = pred[i * 4 + 3];
    const float tl = target[i * 4 + 0], tt = target[i * 4 + 1], tr = target[i * 4 + 2], tb = target[i * 4 + 3];
    const float ta = (tl + tr) * (tt + tb);
    const float pa = (pl + pr) * (pt + pb);
    const float wi = fminf(pl, tl) + fminf(pr, tr);
    const float gwi = fmaxf(pl, tl) + fmaxf(pr, tr);
    const float hi = fminf(pb, tb) + fminf(pt, tt);
    const float ghi = fmaxf(pb, tb) + fmaxf(pt, tt);
    const float ac = gwi * ghi + 1e-7f;
    const float ai = wi * hi;
    const float au = ta + pa - ai;
    const float iou = (ai + 1.f) / (au + 1.f);
    const float giou = iou - (ac - au) / ac;
    losses[i] = 1.f - giou;
    ious[i] = iou;
    if (!gpred) return;
    // loss = 1 - iou + (ac - au)/ac = 2 - iou - au/ac
    // d loss = -d iou - d(au)/ac + au/ac^2 d(ac)
    const float d_ai_iou = 1.f / (au + 1.f), d_au_iou = -(ai + 1.f) / ((au + 1.f) * (au + 1.f));
    // in terms of (ai, au, ac):
    const float g_ai = -d_ai_iou;                 // via iou
    const float g_au = -d_au_iou - 1.f / ac;      // via iou and via au/ac
    const float g_ac = au / (ac * ac);
    // au = ta + pa - ai  ->  g_pa = g_au, g_ai_total = g_ai - g_au
    const float g_pa = g_au, g_ai_t = g_ai - g_au;
    const float g_wi = g_ai_t * hi, g_hi = g_ai_t * wi;
    const float g_gwi = g_ac * ghi, g_ghi = g_ac * gwi;
    float gl = g_pa * (pt + pb), gr = g_pa * (pt + pb), gt = g_pa * (pl + pr), gb = g_pa * (pl + pr), dummy = 0.f;
    min_grad(pl, tl, g_wi, gl, dummy);
    min_grad(pr, tr, g_wi, gr, dummy);
    min_grad(pb, tb, g_hi, gb, dummy);
    min_grad(pt, tt, g_hi, gt, dummy);
    max_grad(pl, tl, g_gwi, gl, dummy);
    max_grad(pr, tr, g_gwi, gr, dummy);
    max_grad(pb, tb, g_ghi, gb, dummy);
    max_grad(pt, tt, g_ghi, gt, dummy);
    gpred[i * 4 + 0] = gl;
    gpred[i * 4 + 1] = gt;
    gpred[i * 4 + 2] = gr;
    gpred[i * 4 + 3] = gb;
}

// ---------------------------------------------------------------------------------------------
// Heat-map NMS + top-K.  DGDE/model/layers/utils.py:45-100.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nms_value(const float *__restrict__ hm, int H, int W, int idx)
{
    const int y = idx / W, x = idx - y * W;
    const float v = hm[idx];
    float mx = v;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) mx = fmaxf(mx, hm[yy * W + xx]);
        }
    return (mx == v) ? v : v * 0.f;  // hm * (hmax == hm).float()
}

// order-preserving float -> uint key (larger float -> larger key), handles negatives
__device__ __forceinline__ unsigned f2key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void nms_hm_kernel(const float *__restrict__ heat, int H, int W, int64_t total, float *__restrict__ out)
{
    const int HW = H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t plane = i / HW;
        out[i] = nms_value(heat + plane * HW, H, W, (int)(i - plane * HW));
    }
}

constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_MAXK = 128;

// One workgroup per (image, class): radix-select the K-th largest key, collect, rank-sort.
__global__ __launch_bounds__(TOPK_THREADS) void heatmap_topk_class(const float *__restrict__ heat, int H, int W, int K, int fuse_nms,
                                                                   float *__restrict__ cls_scores, int *__restrict__ cls_inds)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining, s_cnt_gt, s_cnt_eq;
    __shared__ float cand_v[TOPK_MAXK];
    __shared__ int cand_i[TOPK_MAXK];
    __shared__ int eq_list[TOPK_MAXK];
    const int HW = H * W, tid = threadIdx.x;
    const float *hm = heat + (size_t)blockIdx.x * HW;

    unsigned prefix = 0, remaining = (unsigned)K;  // find the K-th largest key
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned himask = (pass == 3) ? 0u : (0xffffffffu << ((pass + 1) * 8));
        for (int i = tid; i < HW; i += TOPK_THREADS) {
            const unsigned key = f2key(fuse_nms ? nms_value(hm, H, W, i) : hm[i]);
            if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> (pass * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned rem = remaining, d = 255;
            for (;; --d) {
                if (hist[d] >= rem || d == 0) break;
                rem -= hist[d];
            }
            s_prefix = prefix | (d << (pass * 8));
            s_remaining = rem;
        }
        __syncthreads();
        prefix = s_prefix;
        remaining = s_remaining;
        __syncthreads();
    }
    // prefix == key of the K-th largest; `remaining` of the elements equal to it are needed (lowest index first)
    if (tid == 0) { s_cnt_gt = 0; s_cnt_eq = 0; }
    __syncthreads();
    for (int base = 0; base < HW; base += TOPK_THREADS) {
        const int i = base + tid;
        if (i < HW) {
            const float v = fuse_nms ? nms_value(hm, H, W, i) : hm[i];
            const unsigned key = f2key(v);
            if (key > prefix) {
                const unsigned slot = atomicAdd(&s_cnt_gt, 1u);
                cand_v[slot] = v;
                cand_i[slot] = i;
            } else if (key == prefix) {
                const unsigned slot = atomicAdd(&s_cnt_eq, 1u);
                if (slot < (unsigned)TOPK_MAXK) eq_list[slot] = i;  // may overflow; resolved below
            }
        }
        __syncthreads();
        // stop scanning for equal elements once enough low-index ones are collected (uniform decision)
    }
    __syncthreads();
    const unsigned ngt = s_cnt_gt;  // == K - remaining
    // equal elements: need the `remaining` lowest indices.  Recount deterministically by rank.
    const float eqv = (prefix & 0x80000000u) ? __uint_as_float(prefix & 0x7fffffffu) : __uint_as_float(~prefix);
    if (s_cnt_eq <= (unsigned)TOPK_MAXK) {
        // rank the collected equal indices
        const unsigned ne = s_cnt_eq;
        if (tid < (int)ne) {
            const int mine = eq_list[tid];
            unsigned rank = 0;
            for (unsigned e = 0; e < ne; ++e) rank += (eq_list[e] < mine) ? 1u : 0u;
            if (rank < remaining) { cand_v[ngt + rank] = eqv; cand_i[ngt + rank] = mine; }
        }
    } else {
        // many ties (e.g. zeros after NMS): take the first `remaining` in index order with a serial-prefix scan
        __shared__ unsigned s_taken;
        if (tid == 0) s_taken = 0;
        __syncthreads();
        for (int base = 0; base < HW && s_taken < remaining; base += TOPK_THREADS) {
            const int i = base + tid;
            bool iseq = false;
            if (i < HW) iseq = f2key(fuse_nms ? nms_value(hm, H, W, i) : hm[i]) == prefix;
            // block-wide exclusive count of iseq among lower tids
            const unsigned long long bal = __ballot(iseq);
            const int lane = tid & 63, wave = tid >> 6;
            __shared__ unsigned wcnt[TOPK_THREADS / 64];
            if (lane == 0) wcnt[wave] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = s_taken;
            for (int w = 0; w < wave; ++w) before += wcnt[w];
            before += (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            if (iseq && before < remaining) { cand_v[ngt + before] = eqv; cand_i[ngt + before] = i; }
            __syncthreads();
            if (tid == 0) {
                unsigned tot = 0;
                for (int w = 0; w < TOPK_THREADS / 64; ++w) tot += wcnt[w];
                s_taken += tot;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // rank sort the K candidates: value descending, index ascending
    if (tid < K) {
        const float v = cand_v[tid];
        const int ix = cand_i[tid];
        int rank = 0;
        for (int e = 0; e < K; ++e) {
            const float ve = cand_v[e];
            rank += (ve > v || (ve == v && cand_i[e] < ix)) ? 1 : 0;
        }
        cls_scores[(size_t)blockIdx.x * K + rank] = v;
        cls_inds[(size_t)blockIdx.x * K + rank] = ix;
    }
}

// Merge the per-class lists of one image: top-K over C*K (utils.py:86-98).
__global__ void heatmap_topk_merge(const float *__restrict__ cls_scores, const int *__restrict__ cls_inds, int C, int K, int W,
                                   float *__restrict__ scores, int64_t *__restrict__ inds, float *__restrict__ clses,
                                   float *__restrict__ ys, float *__restrict__ xs)
{
    const int b = blockIdx.x, n = C * K;
    const float *sc = cls_scores + (size_t)b * n;
    const int *si = cls_inds + (size_t)b * n;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const float v = sc[e];
        int rank = 0;
        for (int f = 0; f < n; ++f) rank += (sc[f] > v || (sc[f] == v && f < e)) ? 1 : 0;
        if (rank < K) {
            const int idx = si[e];
            scores[(size_t)b * K + rank] = v;
            inds[(size_t)b * K + rank] = idx;
            clses[(size_t)b * K + rank] = (float)e / (float)K;   // reference: true division (utils.py:91)
            ys[(size_t)b * K + rank] = (float)(idx / W);
            xs[(size_t)b * K + rank] = (float)(idx % W);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// POI gather / scatter.  DGDE/model/layers/utils.py:120-145.
// ---------------------------------------------------------------------------------------------
__global__ void poi_gather_kernel(const float *__restrict__ feat, const int64_t *__restrict__ index, int C, int HW, int M,
                                  int64_t total, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bm = i / C;
        const int64_t b = bm / M;
        const int64_t idx = index[bm];
        out[i] = (idx >= 0 && idx < HW) ? feat[((size_t)b * C + c) * HW + idx] : 0.f;
    }
}

__global__ void poi_scatter_kernel(const float *__restrict__ gout, const int64_t *__restrict__ index, int C, int HW, int M,
                                   int64_t total, float *__restrict__ gfeat)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bm = i / C;
        const int64_t b = bm / M;
        const int64_t idx = index[bm];
        if (idx >= 0 && idx < HW) atomicAdd(gfeat + ((size_t)b * C + c) * HW + idx, gout[i]);
    }
}

// Same sum with the lanes along the POSITIONS: lists of neighbouring cells (the 836 border cells of the edge-fusion branch walk the
// image border) then issue atomics to consecutive addresses -- one cache-line request per wave and channel instead of one per
// lane (atomics retire per cache-line request).  Used from 64 positions per image on.
// out[b][c][base[b][m] + tap_off(t)] += g[b][c*9 + t][m] for the nine taps of a 3x3 window in a plane of row pitch `pitch`:
// the gradient of patches gathered at listed cells (trunk_moments.py), lanes along the positions m.
__global__ void patch_scatter_kernel(const float *__restrict__ g, const int64_t *__restrict__ base, int C, int64_t L, int pitch, int M,
                                     int64_t total, float *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const int64_t r = i / M;                   // (b * C + c) * 9 + t
        const int t = (int)(r % 9);
        const int64_t bc = r / 9;
        const int64_t b = bc / C;
        const int64_t idx = base[b * M + m] + (t / 3) * pitch + (t % 3);
        if (idx >= 0 && idx < L) atomicAdd(out + bc * L + idx, g[i]);
    }
}

__global__ void poi_scatter_rows_kernel(const float *__restrict__ gout, const int64_t *__restrict__ index, int C, int HW, int M,
                                        int64_t total, float *__restrict__ gfeat)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const int64_t bc = i / M;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        const int64_t idx = index[b * M + m];
        if (idx >= 0 && idx < HW) atomicAdd(gfeat + ((size_t)b * C + c) * HW + idx, gout[((size_t)b * M + m) * C + c]);
    }
}


// ---------------------------------------------------------------------------------------------
// 3-D IoU of box pairs (logging metric).  DGDE/model/layers/iou_loss.py:99-136 (shapely there):
// bird's-eye-view overlap of the two bottom rectangles (corners 0..3, x-z plane) by Sutherland-Hodgman
// clipping, times the overlap of the height intervals.  One lane per pair.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float poly_area(const float *px, const float *py, int n)
{
    float a = 0.f;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1 == n) ? 0 : i + 1;
        a += px[i] * py[j] - px[j] * py[i];
    }
    return 0.5f * fabsf(a);
}

__global__ void iou3d_kernel(const float *__restrict__ A, const float *__restrict__ B, int N, float *__restrict__ iou)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float *a = A + (size_t)n * 24, *b = B + (size_t)n * 24;
    float ax[4], az[4], bx[4], bz[4];
    float min_ha = 0.f, max_ha = 0.f, min_hb = 0.f, max_hb = 0.f;
    for (int k = 0; k < 4; ++k) {
        ax[k] = a[k * 3 + 0]; az[k] = a[k * 3 + 2];
        bx[k] = b[k * 3 + 0]; bz[k] = b[k * 3 + 2];
        min_ha -= a[k * 3 + 1]; max_ha -= a[(k + 4) * 3 + 1];
        min_hb -= b[k * 3 + 1]; max_hb -= b[(k + 4) * 3 + 1];
    }
    min_ha *= 0.25f; max_ha *= 0.25f; min_hb *= 0.25f; max_hb *= 0.25f;
    const float h_overlap = fmaxf(0.f, fminf(max_ha, max_hb) - fmaxf(min_ha, min_hb));
    // clip polygon a by the four edges of b
    float px[16], py[16], qx[16], qy[16];
    int np = 4;
    for (int k = 0; k < 4; ++k) { px[k] = ax[k]; py[k] = az[k]; }
    float orient = 0.f;
    for (int k = 0; k < 4; ++k) { const int j = (k + 1) & 3; orient += bx[k] * bz[j] - bx[j] * bz[k]; }
    const float sgn = orient >= 0.f ? 1.f : -1.f;
    for (int e = 0; e < 4 && np > 0; ++e) {
        const int e2 = (e + 1) & 3;
        const float ex = bx[e2] - bx[e], ez = bz[e2] - bz[e];
        int nq = 0;
        for (int k = 0; k < np; ++k) {
            const int j = (k + 1 == np) ? 0 : k + 1;
            const float sk = sgn * (ex * (py[k] - bz[e]) - ez * (px[k] - bx[e]));
            const float sj = sgn * (ex * (py[j] - bz[e]) - ez * (px[j] - bx[e]));
            if (sk >= 0.f) { qx[nq] = px[k]; qy[nq] = py[k]; ++nq; }
            if ((sk >= 0.f) != (sj >= 0.f)) {
                const float t = sk / (sk - sj);
                qx[nq] = px[k] + t * (px[j] - px[k]);
                qy[nq] = py[k] + t * (py[j] - py[k]);
                ++nq;
            }
        }
        np = nq;
        for (int k = 0; k < np; ++k) { px[k] = qx[k]; py[k] = qy[k]; }
    }
    const float overlap = (np >= 3) ? poly_area(px, py, np) : 0.f;
    const float o3 = overlap * h_overlap;
    const float uni = poly_area(ax, az, 4) * (max_ha - min_ha) + poly_area(bx, bz, 4) * (max_hb - min_hb) - o3;
    iou[n] = o3 / uni;
}

inline int grid_for(int64_t n, int block) { int64_t g = (n + block - 1) / block; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

// Context normalisation of GMW's point-feature extractor (GMW/model/yi2018cvpr/ops.py:5-17): every (sample, channel) row of K
// points is shifted to zero mean and scaled by 1 / sqrt(unbiased variance + eps).  One wave per row; the row (2628 floats for 73
// keypoints) is read from L1/L2 three times (mean, variance, write) -- the stock chain is 7 launches forward and ~15 backward,
// 48 times per step.  backward: dx = inv (dy - mean(dy) - y sum(dy y) / (K - 1)).
__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void context_norm_fwd(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ inv_out,
                                                        int rows, int K, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * K;
    float s = 0.f;
    for (int i = lane; i < K; i += 64) s += xr[i];
    const float m = wave_sum64(s) / (float)K;
    float q = 0.f;
    for (int i = lane; i < K; i += 64) { const float d = xr[i] - m; q += d * d; }
    const float var = wave_sum64(q) / (float)(K - 1);
    const float inv = 1.f / sqrtf(var + eps);
    float *yr = y + (size_t)row * K;
    for (int i = lane; i < K; i += 64) yr[i] = (xr[i] - m) * inv;
    if (lane == 0) inv_out[row] = inv;
}

__global__ __launch_bounds__(256) void context_norm_bwd(const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ inv_in, float *__restrict__ dx, int rows, int K)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *gr = dy + (size_t)row * K, *yr = y + (size_t)row * K;
    float s0 = 0.f, s1 = 0.f;
    for (int i = lane; i < K; i += 64) { const float g = gr[i]; s0 += g; s1 += g * yr[i]; }
    const float mean_g = wave_sum64(s0) / (float)K, c = wave_sum64(s1) / (float)(K - 1);
    const float inv = inv_in[row];
    float *dr = dx + (size_t)row * K;
    for (int i = lane; i < K; i += 64) dr[i] = inv * (gr[i] - mean_g - yr[i] * c);
}

// Row-per-workgroup forms for K % 4 == 0, K <= 4096 (GMW: K = 2628): the row lives in registers (up to four float4 per thread), so
// it is read once and written once -- the wave-per-row kernels above make three dependent passes with four waves per CU (22 us for
// the 10.8 MB of an (8, 128, 2628) tensor; this form is launch- and HBM-bound).
__device__ __forceinline__ float block_sum256(float v, float *sh)
{
    v = wave_sum64(v);
    __syncthreads();                                        // sh may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void context_norm_fwd_row(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ inv_out,
                                                            int K, float eps)
{
    __shared__ float sh[4];
    const int row = blockIdx.x, K4 = K >> 2;
    const float4 *xr = reinterpret_cast<const float4 *>(x + (size_t)row * K);
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + 256 * j;
        v[j] = i < K4 ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float m = block_sum256(s, sh) / (float)K;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (threadIdx.x + 256 * j < K4) {
            v[j].x -= m; v[j].y -= m; v[j].z -= m; v[j].w -= m;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
    const float inv = 1.f / sqrtf(block_sum256(q, sh) / (float)(K - 1) + eps);
    float4 *yr = reinterpret_cast<float4 *>(y + (size_t)row * K);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < K4) yr[i] = make_float4(v[j].x * inv, v[j].y * inv, v[j].z * inv, v[j].w * inv);
    }
    if (threadIdx.x == 0) inv_out[row] = inv;
}

__global__ __launch_bounds__(256) void context_norm_bwd_row(const float *__restrict__ dy, const float *__restrict__ y,
                                                            const float *__restrict__ inv_in, float *__restrict__ dx, int K)
{
    __shared__ float sh[4];
    const int row = blockIdx.x, K4 = K >> 2;
    const float4 *gr = reinterpret_cast<const float4 *>(dy + (size_t)row * K), *yr = reinterpret_cast<const float4 *>(y + (size_t)row * K);
    float4 g[4], v[4];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + 256 * j;
        g[j] = v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < K4) { g[j] = gr[i]; v[j] = yr[i]; }
        s0 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s1 += (g[j].x * v[j].x + g[j].y * v[j].y) + (g[j].z * v[j].z + g[j].w * v[j].w);
    }
    const float mean_g = block_sum256(s0, sh) / (float)K;
    const float c = block_sum256(s1, sh) / (float)(K - 1);
    const float inv = inv_in[row];
    float4 *dr = reinterpret_cast<float4 *>(dx + (size_t)row * K);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < K4)
            dr[i] = make_float4(inv * (g[j].x - mean_g - v[j].x * c), inv * (g[j].y - mean_g - v[j].y * c),
                                inv * (g[j].z - mean_g - v[j].z * c), inv * (g[j].w - mean_g - v[j].w * c));
    }
}

// out = sum of n tensors (n <= 16) in one pass: the gradient of a feature map that fans out to the head trunks.
struct SumSrcs {
    const float *p[16];
};

__global__ __launch_bounds__(256) void sum_tensors_kernel(SumSrcs srcs, int n, float *out, int64_t n4, int64_t numel)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4 *>(srcs.p[0])[i];
        for (int k = 1; k < n; ++k) {
            const float4 v = reinterpret_cast<const float4 *>(srcs.p[k])[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[i] = a;
    }
    // tail: numel % 4 elements -- or everything, when a pointer is not 16-byte aligned (n4 = 0)
    for (int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < numel; t += stride) {
        float a = srcs.p[0][t];
        for (int k = 1; k < n; ++k) a += srcs.p[k][t];
        out[t] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// Output layers of the regression heads at listed rows (dcd_head_rows_*).  detector_predictor.py:84-101, :198-203.
// forward: a workgroup holds four rows' T trunk vectors in LDS and takes an eighth of the channels; a wave handles one channel
// at a time, its lanes split the K = 256 contraction into float4 pieces (one coalesced 1 KB read of the weight row serves the
// four rows) and meet in a wave sum.
// ---------------------------------------------------------------------------------------------
constexpr int HR_KMAX = 256, HR_TMAX = 16;

__device__ __forceinline__ int hr_head_of(const dcd_head_rows_args &a, int ch)
{
    int j = 0;
    while (j + 1 < a.n_heads && ch >= a.ch0[j + 1]) ++j;
    return j;
}

constexpr int HR_ROWS = 4, HR_CSPLIT = 8;      // rows per workgroup (one weight read serves four rows), channel slices per row group

__global__ __launch_bounds__(256) void head_rows_fwd_kernel(const dcd_head_rows_args a)
{
    extern __shared__ __attribute__((aligned(16))) float f[];          // [HR_ROWS][T][K]
    const int r0 = blockIdx.x * HR_ROWS, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int TK = a.T * a.K;
    for (int e = tid; e < HR_ROWS * TK; e += 256) {
        const int q = e / TK, rem = e - q * TK;
        const int t = rem / a.K, k = rem - t * a.K;
        f[e] = r0 + q < a.R ? a.feat[((size_t)t * a.R + r0 + q) * a.K + k] : 0.f;
    }
    __syncthreads();
    for (int ch = blockIdx.y * 4 + wave; ch < a.C; ch += 4 * HR_CSPLIT) {
        const int j = hr_head_of(a, ch), o = ch - a.ch0[j];
        const float *w = a.weight[j] + (size_t)o * a.K, *x = f + a.trunk[j] * a.K;
        float acc[HR_ROWS];
#pragma unroll
        for (int q = 0; q < HR_ROWS; ++q) acc[q] = 0.f;
        for (int k = 4 * lane; k < a.K; k += 256) {
            const float4 wv = *reinterpret_cast<const float4 *>(w + k);
#pragma unroll
            for (int q = 0; q < HR_ROWS; ++q) {
                const float4 xv = *reinterpret_cast<const float4 *>(x + q * TK + k);
                acc[q] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[q]))));
            }
        }
#pragma unroll
        for (int q = 0; q < HR_ROWS; ++q) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc[q] += __shfl_xor(acc[q], m, 64);
        }
        const float bv = a.bias[j] ? a.bias[j][o] : 0.f;
        if (lane < HR_ROWS && r0 + lane < a.R) {
            float v = acc[0];
#pragma unroll
            for (int q = 1; q < HR_ROWS; ++q) v = lane == q ? acc[q] : v;
            a.y[(size_t)(r0 + lane) * a.C + ch] = v + bv;
        }
    }
}

// feature gradient: a workgroup takes four rows (one read of a weight element serves the four) and ONE trunk, thread = k; the
// trunk's gradient is the sum over the channels of all its heads
__global__ __launch_bounds__(256) void head_rows_bwd_feat_kernel(const dcd_head_rows_args a)
{
    __shared__ float g[HR_ROWS][1024];
    const int r0 = blockIdx.x * HR_ROWS, t = blockIdx.y, k = threadIdx.x;
    int j0 = 0;
    while (j0 < a.n_heads && a.trunk[j0] != t) ++j0;
    int j1 = j0;
    while (j1 < a.n_heads && a.trunk[j1] == t) ++j1;
    const int c0 = j0 < a.n_heads ? a.ch0[j0] : 0, c1 = j1 > j0 ? a.ch0[j1 - 1] + a.out[j1 - 1] : c0;
    for (int e = k; e < HR_ROWS * (c1 - c0); e += 256) {
        const int q = e / (c1 - c0), c = e - q * (c1 - c0);
        g[q][c] = r0 + q < a.R ? a.grad_y[(size_t)(r0 + q) * a.C + c0 + c] : 0.f;
    }
    __syncthreads();
    if (k >= a.K) return;
    float acc[HR_ROWS];
#pragma unroll
    for (int q = 0; q < HR_ROWS; ++q) acc[q] = 0.f;
    for (int j = j0; j < j1; ++j) {
        const float *w = a.weight[j] + k;
        const int cb = a.ch0[j] - c0;
        for (int o = 0; o < a.out[j]; ++o) {
            const float wv = w[(size_t)o * a.K];
#pragma unroll
            for (int q = 0; q < HR_ROWS; ++q) acc[q] = fmaf(g[q][cb + o], wv, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < HR_ROWS; ++q)
        if (r0 + q < a.R) a.grad_feat[((size_t)t * a.R + r0 + q) * a.K + k] = acc[q];
}

// weight / bias gradient: one workgroup per output channel, thread = k, rows in order (reproducible)
__global__ __launch_bounds__(256) void head_rows_bwd_weight_kernel(const dcd_head_rows_args a)
{
    __shared__ float g[256];
    const int ch = blockIdx.x, k = threadIdx.x;
    const int j = hr_head_of(a, ch), o = ch - a.ch0[j];
    const float *x = a.feat + (size_t)a.trunk[j] * a.R * a.K;
    float acc = 0.f, gb = 0.f;
    for (int r0 = 0; r0 < a.R; r0 += 256) {
        __syncthreads();
        if (r0 + k < a.R) g[k] = a.grad_y[(size_t)(r0 + k) * a.C + ch];
        __syncthreads();
        const int n = a.R - r0 < 256 ? a.R - r0 : 256;
        if (k < a.K)
            for (int i = 0; i < n; ++i) acc = fmaf(g[i], x[(size_t)(r0 + i) * a.K + k], acc);
        if (k == 0)
            for (int i = 0; i < n; ++i) gb += g[i];
    }
    if (k < a.K) a.grad_weight[j][(size_t)o * a.K + k] = acc;
    if (k == 0 && a.grad_bias[j]) a.grad_bias[j][o] = gb;
}

}  // namespace

extern "C" {

const char *dcd_version(void) { return "dcd_hip 0.1 gfx950"; }

int dcd_edge_depth_forward(void *stream_, const float *kps, const float *kps3d, const float *rot_y, const float *P,
                           const uint8_t *kmask, int N, int K, int topk, float zmin, float zmax, int normalized,
                           int sub_b3, float *depth, int32_t *pair_idx, float *pair_mask)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (N == 0) return DCD_OK;
    if (!kps || !kps3d || !rot_y || !P || !depth || N < 0 || K < 2 || K > EDGE_MAXK) return DCD_ERR_BAD_ARG;
    const int npairs = K * (K - 1) / 2;
    if (topk < 0 || topk > npairs) return DCD_ERR_BAD_ARG;
    if (topk && !pair_idx) return DCD_ERR_BAD_ARG;
    int npad = 1;
    while (npad < npairs) npad <<= 1;
    size_t lds = sizeof(float) * (3 * EDGE_MAXK + ((npairs + 1) & ~1)) + (topk ? sizeof(unsigned long long) * npad : 0);
    hipLaunchKernelGGL(edge_depth_fwd, dim3(N), dim3(EDGE_THREADS), lds, stream, kps, kps3d, rot_y, P, kmask, K, topk, zmin,
                       zmax, normalized, sub_b3, depth, pair_idx, pair_mask);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_edge_depth_backward(void *stream_, const float *kps, const float *kps3d, const float *rot_y, const float *P,
                            const float *grad_depth, const int32_t *pair_idx, int N, int K, int topk, float zmin,
                            float zmax, int normalized, float *grad_kps, float *grad_kps3d)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (N == 0) return DCD_OK;
    if (!kps || !kps3d || !rot_y || !P || !grad_depth || !grad_kps || !grad_kps3d || N < 0 || K < 2 || K > EDGE_MAXK)
        return DCD_ERR_BAD_ARG;
    if (topk && !pair_idx) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(edge_depth_bwd, dim3(N), dim3(256), 0, stream, kps, kps3d, rot_y, P, grad_depth, pair_idx, K, topk,
                       zmin, zmax, normalized, grad_kps, grad_kps3d);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_focal_loss(void *stream_, const float *pred, const float *target, int64_t n, float alpha, float beta, float *out,
                   float *grad_pred)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!pred || !target || !out || n < 0) return DCD_ERR_BAD_ARG;
    if (!dcd_zero_fill(stream, out, 2)) return DCD_ERR_LAUNCH;          // a launch, not a memset node: zero_fill.h
    if (n == 0) return DCD_OK;
    const int grid = grid_for(n, 256 * 4);
    hipLaunchKernelGGL(focal_loss_kernel, dim3(grid), dim3(256), 0, stream, pred, target, n, alpha, beta, out, grad_pred);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_giou_loss(void *stream_, const float *pred, const float *target, int N, float *losses, float *ious, float *grad_pred)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (N == 0) return DCD_OK;
    if (!pred || !target || !losses || !ious || N < 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(giou_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, pred, target, N, losses, ious, grad_pred);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_nms_hm(void *stream_, const float *heat, int B, int C, int H, int W, float *out)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!heat || !out || B < 0 || C <= 0 || H <= 0 || W <= 0) return DCD_ERR_BAD_ARG;
    const int64_t total = (int64_t)B * C * H * W;
    if (total == 0) return DCD_OK;
    hipLaunchKernelGGL(nms_hm_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, heat, H, W, total, out);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

size_t dcd_heatmap_topk_workspace_bytes(int B, int C, int H, int W, int K)
{
    (void)H; (void)W;
    if (B <= 0 || C <= 0 || K <= 0) return 0;
    return (size_t)B * C * K * (sizeof(float) + sizeof(int)) + 256;
}

int dcd_heatmap_topk(void *stream_, const float *heat, int B, int C, int H, int W, int K, int fuse_nms, float *scores,
                     int64_t *inds, float *clses, float *ys, float *xs, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (B == 0) return DCD_OK;
    if (!heat || !scores || !inds || !clses || !ys || !xs || !workspace) return DCD_ERR_BAD_ARG;
    if (B < 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || K > TOPK_MAXK || (int64_t)H * W > (1 << 24) || K > H * W ||
        (int64_t)C * K > 4096)
        return DCD_ERR_BAD_ARG;
    if (workspace_bytes < dcd_heatmap_topk_workspace_bytes(B, C, H, W, K) - 256) return DCD_ERR_WORKSPACE;
    float *cs = (float *)workspace;
    int *ci = (int *)(cs + (size_t)B * C * K);
    hipLaunchKernelGGL(heatmap_topk_class, dim3(B * C), dim3(TOPK_THREADS), 0, stream, heat, H, W, K, fuse_nms, cs, ci);
    hipLaunchKernelGGL(heatmap_topk_merge, dim3(B), dim3(256), 0, stream, cs, ci, C, K, W, scores, inds, clses, ys, xs);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_poi_gather(void *stream_, const float *feat, const int64_t *index, int B, int C, int H, int W, int M, float *out)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int64_t total = (int64_t)B * M * C;
    if (total == 0) return DCD_OK;
    if (!feat || !index || !out || B < 0 || C <= 0 || H <= 0 || W <= 0 || M < 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(poi_gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, feat, index, C, H * W, M, total, out);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_poi_scatter_add(void *stream_, const float *grad_out, const int64_t *index, int B, int C, int H, int W, int M,
                        float *grad_feat)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int64_t total = (int64_t)B * M * C;
    if (total == 0) return DCD_OK;
    if (!grad_out || !index || !grad_feat || B < 0 || C <= 0 || H <= 0 || W <= 0 || M < 0) return DCD_ERR_BAD_ARG;
    if (M >= 64)
        hipLaunchKernelGGL(poi_scatter_rows_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, grad_out, index, C, H * W, M,
                           total, grad_feat);
    else
        hipLaunchKernelGGL(poi_scatter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, grad_out, index, C, H * W, M, total,
                           grad_feat);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_patch_scatter_add(void *stream_, const float *grad_patches, const int64_t *base, int B, int C, int64_t plane, int pitch,
                          int M, float *out)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int64_t total = (int64_t)B * C * 9 * M;
    if (total == 0) return DCD_OK;
    if (!grad_patches || !base || !out || B < 0 || C <= 0 || plane <= 0 || pitch <= 0 || M < 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(patch_scatter_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, grad_patches, base, C, plane, pitch, M,
                       total, out);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_iou3d(void *stream_, const float *pred_corners, const float *target_corners, int N, float *iou)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (N == 0) return DCD_OK;
    if (!pred_corners || !target_corners || !iou || N < 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(iou3d_kernel, dim3((N + 63) / 64), dim3(64), 0, stream, pred_corners, target_corners, N, iou);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_context_norm_forward(void *stream_, const float *x, float *y, float *inv, int rows, int K, float eps)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (rows == 0) return DCD_OK;
    if (!x || !y || !inv || rows < 0 || K < 2) return DCD_ERR_BAD_ARG;
    if ((K & 3) == 0 && K <= 4096 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
        hipLaunchKernelGGL(context_norm_fwd_row, dim3(rows), dim3(256), 0, stream, x, y, inv, K, eps);
    else
        hipLaunchKernelGGL(context_norm_fwd, dim3((rows + 3) / 4), dim3(256), 0, stream, x, y, inv, rows, K, eps);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_context_norm_backward(void *stream_, const float *grad_y, const float *y, const float *inv, float *grad_x, int rows, int K)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (rows == 0) return DCD_OK;
    if (!grad_y || !y || !inv || !grad_x || rows < 0 || K < 2) return DCD_ERR_BAD_ARG;
    if ((K & 3) == 0 && K <= 4096 && (((uintptr_t)grad_y | (uintptr_t)y | (uintptr_t)grad_x) & 15) == 0)
        hipLaunchKernelGGL(context_norm_bwd_row, dim3(rows), dim3(256), 0, stream, grad_y, y, inv, grad_x, K);
    else
        hipLaunchKernelGGL(context_norm_bwd, dim3((rows + 3) / 4), dim3(256), 0, stream, grad_y, y, inv, grad_x, rows, K);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_sum_tensors(void *stream_, const float *const *srcs, int n, float *out, int64_t numel)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (numel == 0) return DCD_OK;
    if (!srcs || !out || n < 1 || n > 16 || numel < 0) return DCD_ERR_BAD_ARG;
    SumSrcs a;
    bool aligned = ((uintptr_t)out & 15) == 0;
    for (int k = 0; k < 16; ++k) {
        a.p[k] = srcs[k < n ? k : 0];
        if (!a.p[k]) return DCD_ERR_BAD_ARG;
        aligned = aligned && ((uintptr_t)a.p[k] & 15) == 0;
    }
    const int64_t n4 = aligned ? numel / 4 : 0;           // views at odd offsets take the scalar loop
    int64_t blocks = ((aligned ? n4 : numel) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sum_tensors_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, n, out, n4, numel);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static bool head_rows_ok(const dcd_head_rows_args *a)
{
    if (!a || a->n_heads <= 0 || a->n_heads > DCD_HEADS_MAX || a->T <= 0 || a->T > HR_TMAX || a->R < 0 || a->K <= 0 || a->K > HR_KMAX ||
        (a->K & 3) || a->C <= 0 || a->C > 1024 || !a->feat)
        return false;
    int c = 0;
    for (int j = 0; j < a->n_heads; ++j) {
        if (a->ch0[j] != c || a->out[j] <= 0 || a->trunk[j] < 0 || a->trunk[j] >= a->T || !a->weight[j]) return false;
        if (j && a->trunk[j] < a->trunk[j - 1]) return false;                 // heads ordered by trunk
        c += a->out[j];
    }
    return c == a->C;
}

int dcd_head_rows_forward(void *stream_, const dcd_head_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!head_rows_ok(a) || !a->y) return DCD_ERR_BAD_ARG;
    if (a->R == 0) return DCD_OK;
    hipLaunchKernelGGL(head_rows_fwd_kernel, dim3((a->R + HR_ROWS - 1) / HR_ROWS, HR_CSPLIT), dim3(256),
                       sizeof(float) * HR_ROWS * a->T * a->K, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_head_rows_backward(void *stream_, const dcd_head_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!head_rows_ok(a) || !a->grad_y || !a->grad_feat) return DCD_ERR_BAD_ARG;
    for (int j = 0; j < a->n_heads; ++j)
        if (!a->grad_weight[j]) return DCD_ERR_BAD_ARG;
    if (a->R > 0) hipLaunchKernelGGL(head_rows_bwd_feat_kernel, dim3((a->R + HR_ROWS - 1) / HR_ROWS, a->T), dim3(256), 0, stream, *a);
    hipLaunchKernelGGL(head_rows_bwd_weight_kernel, dim3(a->C), dim3(256), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
