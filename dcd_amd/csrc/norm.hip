// Batch normalisation (training + eval) fused with the residual add and the ReLU that follow it everywhere in DLA-34
// and the CenterNet heads (DGDE/model/backbone/dla_dcn.py:71-101 BasicBlock, :187-207 Root, :267-295 conv levels,
// :398-410 DeformConv; DGDE/model/head/detector_predictor.py:52-60,112-120).  gfx950 only.
//
// Why it exists: MIOpen's spatial BN runs ONE workgroup per channel and ATen's native kernels one block-row per
// channel; with 16 / 32 channels at 384x1280 (251 MB tensors) that is 2-3 ms per call on a 256-CU part.  Here the
// reduction is split over (slice, channel) so every layer launches >= 1024 workgroups, sums are carried in fp64 (exact
// to ~1e-16, so var = E[x^2] - mean^2 has no cancellation problem), and normalise + residual + ReLU is one pass.
//
// HBM passes (fp32 tensors of N = B*C*HW elements):
//   forward : stats 1R;  apply 1R (+1R residual) 1W                       (stock: BN 2R 1W, add 2R 1W, relu 1R 1W)
//   backward: sums  3R (dy, y, x);  apply 3R 1W (+1W grad_residual)        (stock: relu 2R 1W, BN 4R 1W, add 1R ..)
// All kernels are HBM-bound; layout NCHW, one (b, c) plane chunk per workgroup so scale/shift are scalars.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/dcd_hip.h"
#include "tuning_env.h"

namespace {

constexpr int BT = 256;            // threads per workgroup
constexpr int CH = 4096;           // elements of one (b, c) plane handled per workgroup pass (4 float4 per thread)
constexpr int SMAX = 256;          // max slices per channel in the two-stage reductions

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// shift of y = x * scale + shift: ONE spelling for the forward kernels and for the backward kernels that recompute the ReLU mask
// from x (an explicit fma: the compiler's own contraction could differ from kernel to kernel)
__device__ inline float bn_shift(float bias, float mean, float scale) { return fmaf(-mean, scale, bias); }

// Block-wide sum of two doubles; result valid in thread 0.
__device__ inline void block_sum2(double &a, double &b)
{
    __shared__ double sh[2][BT / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = sh[0][0]; b = sh[1][0];
#pragma unroll
        for (int i = 1; i < BT / 64; ++i) { a += sh[0][i]; b += sh[1][i]; }
    }
}

// (sum0, sum1) of channel c: from the combined array, or -- local (non-synchronised) case -- by reducing the S partials of
// the two-stage reduction right here (first wave, fixed order), which saves the separate combine launch.
__device__ inline void channel_sums(const double *__restrict__ combined, const double *__restrict__ partial, int S, int c,
                                    double &s0, double &s1)
{
    if (!partial) { s0 = combined[2 * c]; s1 = combined[2 * c + 1]; return; }
    __shared__ double bc[2];
    if (threadIdx.x < 64) {
        double a = 0.0, b = 0.0;
        for (int s = threadIdx.x; s < S; s += 64) {
            a += partial[((long)c * S + s) * 2 + 0];
            b += partial[((long)c * S + s) * 2 + 1];
        }
        a = wave_sum(a);
        b = wave_sum(b);
        if (threadIdx.x == 0) { bc[0] = a; bc[1] = b; }
    }
    __syncthreads();
    s0 = bc[0];
    s1 = bc[1];
}

struct Plane {
    int B, C, cpp;      // cpp = chunks per plane
    long HW;
};

// ---- forward statistics: partial[c][s] = (sum x, sum x^2) over the chunks s, s+S, ... of channel c ------------------
__global__ __launch_bounds__(BT) void bn_partial(const float *__restrict__ x, Plane g, int S, double *__restrict__ partial)
{
    const int c = blockIdx.y, s = blockIdx.x;
    const int P = g.B * g.cpp;
    const bool vec = (g.HW & 3) == 0;
    double ds = 0.0, dq = 0.0;
    for (int j = s; j < P; j += S) {
        const int b = j / g.cpp, k = j - b * g.cpp;
        const float *p = x + ((long)b * g.C + c) * g.HW;
        const long lo = (long)k * CH, hi = min(g.HW, lo + CH);
        float fs = 0.f, fq = 0.f;
        if (vec) {
            for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * BT) {
                const float4 v = *reinterpret_cast<const float4 *>(p + i);
                fs += (v.x + v.y) + (v.z + v.w);
                fq = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, fq))));
            }
        } else {
            for (long i = lo + threadIdx.x; i < hi; i += BT) {
                const float v = p[i];
                fs += v;
                fq = fmaf(v, v, fq);
            }
        }
        ds += (double)fs;          // <= 16 fp32 terms per thread per chunk, then fp64
        dq += (double)fq;
    }
    block_sum2(ds, dq);
    if (threadIdx.x == 0) {
        partial[((long)c * S + s) * 2 + 0] = ds;
        partial[((long)c * S + s) * 2 + 1] = dq;
    }
}

// ---- combine S partials per channel (fixed order -> deterministic) ---------------------------------------------------
// gw / gb (backward sums only, may be null): this rank's grad_weight = sum(dz * (x - mean)) * invstd and grad_bias = sum(dz) as fp32
// -- under data parallelism the parameter gradients stay local while the sums are all-reduced for the input gradient.
__global__ __launch_bounds__(64) void bn_combine(const double *__restrict__ partial, int S, double *__restrict__ out,
                                                 const float *__restrict__ invstd, float *__restrict__ gw, float *__restrict__ gb)
{
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int s = threadIdx.x; s < S; s += 64) {
        a += partial[((long)c * S + s) * 2 + 0];
        b += partial[((long)c * S + s) * 2 + 1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (threadIdx.x == 0) {
        out[2 * c] = a;
        out[2 * c + 1] = b;
        if (gw) gw[c] = (float)(b * (double)invstd[c]);
        if (gb) gb[c] = (float)a;
    }
}

// ---- per-channel sums in ONE launch (a bias gradient): the slices of bn_partial, and the block that arrives last at a
// channel's counter adds the channel's S partials in bn_combine's order -- same value as bn_partial + bn_combine + a cast,
// two launches fewer.  The arrival counters live in the CALL's workspace (behind the partials; round 3 kept them in one
// process-global array, which two launches on different streams would have shared: advisor r3): the caller hands them in
// zeroed, every last block leaves its own at zero again.
__global__ __launch_bounds__(BT) void channel_sum_kernel(const float *__restrict__ x, Plane g, int S, double *__restrict__ partial,
                                                         unsigned *__restrict__ arrivals, float *__restrict__ out)
{
    const int c = blockIdx.y, s = blockIdx.x;
    const int P = g.B * g.cpp;
    const bool vec = (g.HW & 3) == 0;
    double ds = 0.0, unused = 0.0;
    for (int j = s; j < P; j += S) {
        const int b = j / g.cpp, k = j - b * g.cpp;
        const float *p = x + ((long)b * g.C + c) * g.HW;
        const long lo = (long)k * CH, hi = min(g.HW, lo + CH);
        float fs = 0.f;
        if (vec) {
            for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * BT) {
                const float4 v = *reinterpret_cast<const float4 *>(p + i);
                fs += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (long i = lo + threadIdx.x; i < hi; i += BT) fs += p[i];
        }
        ds += (double)fs;
    }
    block_sum2(ds, unused);
    __shared__ int is_last;
    if (!arrivals) {                                     // two-launch form: channel_sum_combine adds the partials
        if (threadIdx.x == 0) partial[(long)c * S + s] = ds;
        return;
    }
    if (threadIdx.x == 0) {
        partial[(long)c * S + s] = ds;
        __threadfence();
        is_last = atomicAdd(&arrivals[c], 1u) == (unsigned)(S - 1);
    }
    __syncthreads();
    if (!is_last || threadIdx.x >= 64) return;
    __threadfence();
    double a = 0.0;
    for (int q = threadIdx.x; q < S; q += 64) a += __builtin_nontemporal_load(partial + (long)c * S + q);
    a = wave_sum(a);
    if (threadIdx.x == 0) {
        out[c] = (float)a;
        arrivals[c] = 0u;
    }
}

__global__ __launch_bounds__(64) void channel_sum_combine(const double *__restrict__ partial, int S, float *__restrict__ out)
{
    const int c = blockIdx.x;
    double a = 0.0;
    for (int q = threadIdx.x; q < S; q += 64) a += partial[(long)c * S + q];
    a = wave_sum(a);
    if (threadIdx.x == 0) out[c] = (float)a;
}

// ---- forward apply: y = act((x - mean) * invstd * w + b [+ residual]) ------------------------------------------------
// stats != nullptr: training, batch statistics from (sum, sumsq, count); else eval with (mean_in, var_in).
__global__ __launch_bounds__(BT) void bn_apply(const float *__restrict__ x, const float *__restrict__ residual,
                                               const float *__restrict__ weight, const float *__restrict__ bias,
                                               const double *__restrict__ stats, const double *__restrict__ partial, int S,
                                               double count, float *__restrict__ running_mean, float *__restrict__ running_var,
                                               long long *__restrict__ num_batches_tracked, float momentum, float eps,
                                               int relu, float *__restrict__ y, float *__restrict__ save_mean,
                                               float *__restrict__ save_invstd, Plane g)
{
    const int c = blockIdx.y;
    const int b = blockIdx.x / g.cpp, k = blockIdx.x - b * g.cpp;
    float mean, invstd;
    if (stats || partial) {
        double s0, s1;
        channel_sums(stats, partial, S, c, s0, s1);
        const double m = s0 / count;
        double var = s1 / count - m * m;
        var = var < 0.0 ? 0.0 : var;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            save_mean[c] = mean;
            save_invstd[c] = invstd;
            if (running_mean) {
                const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
            if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
        }
    } else {
        mean = running_mean[c];
        invstd = 1.f / sqrtf(running_var[c] + eps);
    }
    const float scale = invstd * (weight ? weight[c] : 1.f);
    const float shift = bn_shift(bias ? bias[c] : 0.f, mean, scale);
    const long base = ((long)b * g.C + c) * g.HW;
    const long lo = (long)k * CH, hi = min(g.HW, lo + CH);
    if ((g.HW & 3) == 0) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * BT) {
            float4 v = *reinterpret_cast<const float4 *>(x + base + i);
            v.x = fmaf(v.x, scale, shift); v.y = fmaf(v.y, scale, shift);
            v.z = fmaf(v.z, scale, shift); v.w = fmaf(v.w, scale, shift);
            if (residual) {
                const float4 r = *reinterpret_cast<const float4 *>(residual + base + i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (relu) {      // NaN-propagating, like clamp_min
                v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
            }
            *reinterpret_cast<float4 *>(y + base + i) = v;
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += BT) {
            float v = fmaf(x[base + i], scale, shift);
            if (residual) v += residual[base + i];
            y[base + i] = (relu && v < 0.f) ? 0.f : v;
        }
    }
}

// ---- backward sums: partial[c][s] = (sum dz, sum dz * (x - mean)), dz = dy * [y > 0] when ReLU is fused ---------------
// invstd / gw / gb: only with S == 1 (the block's sums are the channel's: see bn_combine)
__global__ __launch_bounds__(BT) void bn_bwd_partial(const float *__restrict__ dy, const float *__restrict__ y,
                                                     const float *__restrict__ x, const float *__restrict__ save_mean,
                                                     Plane g, int S, double *__restrict__ partial,
                                                     const float *__restrict__ invstd = nullptr, float *__restrict__ gw = nullptr,
                                                     float *__restrict__ gb = nullptr, const float *__restrict__ mask_weight = nullptr,
                                                     const float *__restrict__ mask_bias = nullptr, int mask_from_x = 0)
{
    // mask_from_x (ReLU fused, no residual): y > 0  <=>  fma(x, scale, shift) > 0 with the forward's own scale / shift -- the
    // output map is not read at all (one of the kernel's three tensor reads)
    const int c = blockIdx.y, s = blockIdx.x;
    const int P = g.B * g.cpp;
    const bool vec = (g.HW & 3) == 0;
    const float mean = save_mean[c];
    const float msc = mask_from_x ? invstd[c] * (mask_weight ? mask_weight[c] : 1.f) : 0.f;
    const float msh = mask_from_x ? bn_shift(mask_bias ? mask_bias[c] : 0.f, mean, msc) : 0.f;
    double d0 = 0.0, d1 = 0.0;
    for (int j = s; j < P; j += S) {
        const int b = j / g.cpp, k = j - b * g.cpp;
        const long base = ((long)b * g.C + c) * g.HW;
        const long lo = (long)k * CH, hi = min(g.HW, lo + CH);
        float f0 = 0.f, f1 = 0.f;
        if (vec) {
            for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * BT) {
                float4 d = *reinterpret_cast<const float4 *>(dy + base + i);
                const float4 v = *reinterpret_cast<const float4 *>(x + base + i);
                if (mask_from_x) {
                    d.x = fmaf(v.x, msc, msh) <= 0.f ? 0.f : d.x; d.y = fmaf(v.y, msc, msh) <= 0.f ? 0.f : d.y;
                    d.z = fmaf(v.z, msc, msh) <= 0.f ? 0.f : d.z; d.w = fmaf(v.w, msc, msh) <= 0.f ? 0.f : d.w;
                } else if (y) {
                    const float4 o = *reinterpret_cast<const float4 *>(y + base + i);
                    d.x = o.x <= 0.f ? 0.f : d.x; d.y = o.y <= 0.f ? 0.f : d.y;
                    d.z = o.z <= 0.f ? 0.f : d.z; d.w = o.w <= 0.f ? 0.f : d.w;
                }
                f0 += (d.x + d.y) + (d.z + d.w);
                f1 = fmaf(d.x, v.x - mean, fmaf(d.y, v.y - mean, fmaf(d.z, v.z - mean, fmaf(d.w, v.w - mean, f1))));
            }
        } else {
            for (long i = lo + threadIdx.x; i < hi; i += BT) {
                float d = dy[base + i];
                if (mask_from_x) d = fmaf(x[base + i], msc, msh) <= 0.f ? 0.f : d;
                else if (y) d = y[base + i] <= 0.f ? 0.f : d;
                f0 += d;
                f1 = fmaf(d, x[base + i] - mean, f1);
            }
        }
        d0 += (double)f0;
        d1 += (double)f1;
    }
    block_sum2(d0, d1);
    if (threadIdx.x == 0) {
        partial[((long)c * S + s) * 2 + 0] = d0;
        partial[((long)c * S + s) * 2 + 1] = d1;
        if (gw) gw[c] = (float)(d1 * (double)invstd[c]);
        if (gb) gb[c] = (float)d0;
    }
}

// ---- backward apply: dx = (dz - mean(dz) - xhat * mean(dz * xhat)) * invstd * w;  grad_residual = dz ------------------
__global__ __launch_bounds__(BT) void bn_bwd_apply(const float *__restrict__ dy, const float *__restrict__ y,
                                                   const float *__restrict__ x, const float *__restrict__ weight,
                                                   const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                   const double *__restrict__ sums, const double *__restrict__ partial, int S,
                                                   double count, float *__restrict__ dx,
                                                   float *__restrict__ dres, float *__restrict__ dweight,
                                                   float *__restrict__ dbias, Plane g, const float *__restrict__ mask_bias = nullptr,
                                                   int mask_from_x = 0)
{
    const int c = blockIdx.y;
    const int b = blockIdx.x / g.cpp, k = blockIdx.x - b * g.cpp;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float msc = invstd * (weight ? weight[c] : 1.f);
    const float msh = bn_shift(mask_bias ? mask_bias[c] : 0.f, mean, msc);
    double s0, s1;
    channel_sums(sums, partial, S, c, s0, s1);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (dweight) dweight[c] = (float)(s1 * (double)invstd);
        if (dbias) dbias[c] = (float)s0;
    }
    const float k1 = invstd * (weight ? weight[c] : 1.f);
    const float a = (float)(s0 / count);
    const float bq = (float)(s1 / count * (double)invstd * (double)invstd);
    const long base = ((long)b * g.C + c) * g.HW;
    const long lo = (long)k * CH, hi = min(g.HW, lo + CH);
    if ((g.HW & 3) == 0) {
        for (long i = lo + 4 * threadIdx.x; i < hi; i += 4 * BT) {
            float4 d = dy ? *reinterpret_cast<const float4 *>(dy + base + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v = *reinterpret_cast<const float4 *>(x + base + i);
            if (mask_from_x) {
                d.x = fmaf(v.x, msc, msh) <= 0.f ? 0.f : d.x; d.y = fmaf(v.y, msc, msh) <= 0.f ? 0.f : d.y;
                d.z = fmaf(v.z, msc, msh) <= 0.f ? 0.f : d.z; d.w = fmaf(v.w, msc, msh) <= 0.f ? 0.f : d.w;
            } else if (y) {
                const float4 o = *reinterpret_cast<const float4 *>(y + base + i);
                d.x = o.x <= 0.f ? 0.f : d.x; d.y = o.y <= 0.f ? 0.f : d.y;
                d.z = o.z <= 0.f ? 0.f : d.z; d.w = o.w <= 0.f ? 0.f : d.w;
            }
            if (dres) *reinterpret_cast<float4 *>(dres + base + i) = d;
            float4 r;
            r.x = (d.x - a - (v.x - mean) * bq) * k1; r.y = (d.y - a - (v.y - mean) * bq) * k1;
            r.z = (d.z - a - (v.z - mean) * bq) * k1; r.w = (d.w - a - (v.w - mean) * bq) * k1;
            *reinterpret_cast<float4 *>(dx + base + i) = r;
        }
    } else {
        for (long i = lo + threadIdx.x; i < hi; i += BT) {
            float d = dy ? dy[base + i] : 0.f;
            if (mask_from_x) d = fmaf(x[base + i], msc, msh) <= 0.f ? 0.f : d;
            else if (y) d = y[base + i] <= 0.f ? 0.f : d;
            if (dres) dres[base + i] = d;
            dx[base + i] = (d - a - (x[base + i] - mean) * bq) * k1;
        }
    }
}

// ---- small channels: one launch per direction ---------------------------------------------------------------------------
// When a channel has at most SM_MAX values (B * HW <= 16384: the 24x80 and 12x40 maps of DLA levels 4 / 5 and of the first DCN
// up-sampling stages at 8 images) one workgroup per channel keeps them in registers (<= 16 float4 per thread): statistics and
// normalisation are ONE launch and one read instead of two launches (partial sums + apply) that each cost a dispatch whatever
// their size.  Same fp64 sums, other order of the additions.
constexpr int SM_V = 16;                     // float4 per thread
constexpr int SM_MAX = SM_V * 4 * BT;        // 16384 values per channel

__device__ inline void block_bcast2(double &a, double &b)
{
    __shared__ double bc[2];
    block_sum2(a, b);
    if (threadIdx.x == 0) { bc[0] = a; bc[1] = b; }
    __syncthreads();
    a = bc[0];
    b = bc[1];
}

__global__ __launch_bounds__(BT) void bn_small_fwd(const float *__restrict__ x, const float *__restrict__ residual,
                                                   const float *__restrict__ weight, const float *__restrict__ bias,
                                                   float *__restrict__ running_mean, float *__restrict__ running_var,
                                                   long long *__restrict__ num_batches_tracked, float momentum, float eps, int relu,
                                                   float *__restrict__ y, float *__restrict__ save_mean,
                                                   float *__restrict__ save_invstd, int B, int C, int HW4)
{
    const int c = blockIdx.x;
    const int total = B * HW4;
    float4 v[SM_V];
    long off[SM_V];
    double ds = 0.0, dq = 0.0;
#pragma unroll
    for (int j = 0; j < SM_V; ++j) {
        const int e = threadIdx.x + BT * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        off[j] = -1;
        if (e < total) {
            const int b = e / HW4, q = e - b * HW4;
            off[j] = ((long)b * C + c) * (4L * HW4) + 4L * q;
            v[j] = *reinterpret_cast<const float4 *>(x + off[j]);
            ds += (double)((v[j].x + v[j].y) + (v[j].z + v[j].w));
            dq += (double)fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, v[j].w * v[j].w)));
        }
    }
    block_bcast2(ds, dq);
    const double count = (double)total * 4.0;
    const double m = ds / count;
    double var = dq / count - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
    }
    const float scale = invstd * (weight ? weight[c] : 1.f);
    const float shift = bn_shift(bias ? bias[c] : 0.f, mean, scale);
#pragma unroll
    for (int j = 0; j < SM_V; ++j) {
        if (off[j] < 0) continue;
        float4 r = v[j];
        r.x = fmaf(r.x, scale, shift); r.y = fmaf(r.y, scale, shift); r.z = fmaf(r.z, scale, shift); r.w = fmaf(r.w, scale, shift);
        if (residual) {
            const float4 t = *reinterpret_cast<const float4 *>(residual + off[j]);
            r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w;
        }
        if (relu) {          // NaN-propagating, like clamp_min
            r.x = r.x < 0.f ? 0.f : r.x; r.y = r.y < 0.f ? 0.f : r.y; r.z = r.z < 0.f ? 0.f : r.z; r.w = r.w < 0.f ? 0.f : r.w;
        }
        *reinterpret_cast<float4 *>(y + off[j]) = r;
    }
}

__global__ __launch_bounds__(BT) void bn_small_bwd(const float *__restrict__ dy, const float *__restrict__ y,
                                                   const float *__restrict__ x, const float *__restrict__ weight,
                                                   const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                   float *__restrict__ dx, float *__restrict__ dres, float *__restrict__ dweight,
                                                   float *__restrict__ dbias, int B, int C, int HW4,
                                                   const float *__restrict__ mask_bias = nullptr, int mask_from_x = 0)
{
    const int c = blockIdx.x;
    const int total = B * HW4;
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float msc = invstd * (weight ? weight[c] : 1.f);
    const float msh = bn_shift(mask_bias ? mask_bias[c] : 0.f, mean, msc);
    float4 d[SM_V], v[SM_V];
    long off[SM_V];
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int j = 0; j < SM_V; ++j) {
        const int e = threadIdx.x + BT * j;
        d[j] = v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        off[j] = -1;
        if (e < total) {
            const int b = e / HW4, q = e - b * HW4;
            off[j] = ((long)b * C + c) * (4L * HW4) + 4L * q;
            d[j] = *reinterpret_cast<const float4 *>(dy + off[j]);
            v[j] = *reinterpret_cast<const float4 *>(x + off[j]);
            if (mask_from_x) {
                d[j].x = fmaf(v[j].x, msc, msh) <= 0.f ? 0.f : d[j].x; d[j].y = fmaf(v[j].y, msc, msh) <= 0.f ? 0.f : d[j].y;
                d[j].z = fmaf(v[j].z, msc, msh) <= 0.f ? 0.f : d[j].z; d[j].w = fmaf(v[j].w, msc, msh) <= 0.f ? 0.f : d[j].w;
            } else if (y) {
                const float4 o = *reinterpret_cast<const float4 *>(y + off[j]);
                d[j].x = o.x <= 0.f ? 0.f : d[j].x; d[j].y = o.y <= 0.f ? 0.f : d[j].y;
                d[j].z = o.z <= 0.f ? 0.f : d[j].z; d[j].w = o.w <= 0.f ? 0.f : d[j].w;
            }
            d0 += (double)((d[j].x + d[j].y) + (d[j].z + d[j].w));
            d1 += (double)fmaf(d[j].x, v[j].x - mean, fmaf(d[j].y, v[j].y - mean, fmaf(d[j].z, v[j].z - mean, d[j].w * (v[j].w - mean))));
        }
    }
    block_bcast2(d0, d1);
    if (threadIdx.x == 0) {
        if (dweight) dweight[c] = (float)(d1 * (double)invstd);
        if (dbias) dbias[c] = (float)d0;
    }
    const double count = (double)total * 4.0;
    const float k1 = invstd * (weight ? weight[c] : 1.f);
    const float a = (float)(d0 / count);
    const float bq = (float)(d1 / count * (double)invstd * (double)invstd);
#pragma unroll
    for (int j = 0; j < SM_V; ++j) {
        if (off[j] < 0) continue;
        if (dres) *reinterpret_cast<float4 *>(dres + off[j]) = d[j];
        float4 r;
        r.x = (d[j].x - a - (v[j].x - mean) * bq) * k1; r.y = (d[j].y - a - (v[j].y - mean) * bq) * k1;
        r.z = (d[j].z - a - (v[j].z - mean) * bq) * k1; r.w = (d[j].w - a - (v[j].w - mean) * bq) * k1;
        *reinterpret_cast<float4 *>(dx + off[j]) = r;
    }
}

inline bool small_channels(int B, int C, int64_t HW)
{
    static const bool off = dcd_env("DCD_BN_SMALL") && atoi(dcd_env("DCD_BN_SMALL")) == 0;      // A/B timing
    return !off && (HW & 3) == 0 && (int64_t)B * HW <= SM_MAX && C >= 64;
}

// ---- BN (+ReLU) evaluated at a list of positions only (training forward of the regression-head trunks: the loss reads the
// normalised features at <= 40 object centres per image, plus the 832 border cells for the edge-fusion branch).
// grid = (C); block c finalises its channel (mean / invstd from the stats partials or combined sums, running statistics)
// and normalises the gathered values.  pos (B, N) linear pixel indices; x_at / y_at (B, N, C).
__global__ __launch_bounds__(BT) void bn_at_forward(const float *__restrict__ x, const int64_t *__restrict__ pos,
                                                    const float *__restrict__ weight, const float *__restrict__ bias,
                                                    const double *__restrict__ stats, const double *__restrict__ partial, int S,
                                                    double count, float *__restrict__ running_mean, float *__restrict__ running_var,
                                                    long long *__restrict__ num_batches_tracked, float momentum, float eps,
                                                    int relu, float *__restrict__ x_at, float *__restrict__ y_at,
                                                    float *__restrict__ save_mean, float *__restrict__ save_invstd, int B, int C,
                                                    long HW, int N)
{
    const int c = blockIdx.x;
    double s0, s1;
    channel_sums(stats, partial, S, c, s0, s1);
    const double m = s0 / count;
    double var = s1 / count - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)m, invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
    }
    const float scale = invstd * (weight ? weight[c] : 1.f);
    const float shift = bn_shift(bias ? bias[c] : 0.f, mean, scale);
    for (int e = threadIdx.x; e < B * N; e += BT) {
        const int b = e / N;
        const float v = x[((long)b * C + c) * HW + pos[e]];
        float o = fmaf(v, scale, shift);
        if (relu) o = o < 0.f ? 0.f : o;
        x_at[(long)e * C + c] = v;
        y_at[(long)e * C + c] = o;
    }
}

// backward sums over the listed entries: dz = g * [y > 0];  sums[c] = (sum dz, sum dz (x - mean));  dzk = dz * invstd * w
__global__ __launch_bounds__(BT) void bn_at_backward_sums(const float *__restrict__ g, const float *__restrict__ x_at,
                                                          const float *__restrict__ y_at, const float *__restrict__ weight,
                                                          const float *__restrict__ save_mean, const float *__restrict__ save_invstd,
                                                          int relu, int total, int C, double *__restrict__ sums,
                                                          float *__restrict__ dzk)
{
    const int c = blockIdx.x;
    const float mean = save_mean[c], k1 = save_invstd[c] * (weight ? weight[c] : 1.f);
    double d0 = 0.0, d1 = 0.0;
    for (int e = threadIdx.x; e < total; e += BT) {
        float d = g[(long)e * C + c];
        if (relu) d = y_at[(long)e * C + c] <= 0.f ? 0.f : d;
        d0 += (double)d;
        d1 += (double)d * (double)(x_at[(long)e * C + c] - mean);
        dzk[(long)e * C + c] = d * k1;
    }
    block_sum2(d0, d1);
    if (threadIdx.x == 0) { sums[2 * c] = d0; sums[2 * c + 1] = d1; }
}

inline Plane make_plane(int B, int C, long HW)
{
    Plane g;
    g.B = B; g.C = C; g.HW = HW;
    g.cpp = (int)((HW + CH - 1) / CH);
    return g;
}

// slices per channel: enough workgroups to fill 256 CUs several times over, never more than there are chunks
inline int slices(const Plane &g)
{
    const long P = (long)g.B * g.cpp;
    long want = (4096 + g.C - 1) / g.C;
    if (want > SMAX) want = SMAX;
    if (want > P) want = P;
    return (int)(want < 1 ? 1 : want);
}

// Synchronised statistics (the sums are all-reduced between two launches): one slice per channel while a channel holds at most
// 32 K elements -- the block's sums ARE the channel's, written straight to the output, no combine launch (one image per GPU:
// every DGDE layer but the stem; 110 launches of a one-image data-parallel step).
inline int sync_slices(const Plane &g)
{
    return (long)g.B * g.HW <= 32768 ? 1 : slices(g);
}

// ---- batch-norm parameters of the head's regression trunks from the Gram form (model/head/trunk_moments.py) --------------------
// Row r = (trunk, output channel), K = 9 Cin.  WG (R, K + 1) = W [G | S1] in fp64 (one library product), Wd (R, K) the weights in
// fp64:  sum y = WG[r][K],  sum y^2 = sum_k WG[r][k] Wd[r][k].  Rounds 2-4 wrote what follows as ~30 small fp64 tensor
// operations and let autograd derive ~50 more for the backward; here: row sums, finalisation and their backward as four kernels.
__global__ __launch_bounds__(64) void trunk_row_sums(const double *__restrict__ WG, const double *__restrict__ Wd, int K,
                                                     double *__restrict__ sums)
{
    const int r = blockIdx.x;
    double a = 0.0;
    for (int k = threadIdx.x; k < K; k += 64) a += WG[(long)r * (K + 1) + k] * Wd[(long)r * K + k];
    a = wave_sum(a);
    if (threadIdx.x == 0) {
        sums[2 * r] = WG[(long)r * (K + 1) + K];
        sums[2 * r + 1] = a;
    }
}

// stats (R, 3) fp64 = (mean, biased variance after the clamp at 0, 1 / sqrt(var + eps)); scale = gamma invstd, shift = beta - mean scale
__global__ __launch_bounds__(256) void trunk_finalize_fwd(const double *__restrict__ sums, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, double n, double eps, int R,
                                                          float *__restrict__ scale, float *__restrict__ shift, double *__restrict__ stats)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double mean = sums[2 * r] / n;
    const double raw = sums[2 * r + 1] / n - mean * mean;
    const double var = raw > 0.0 ? raw : 0.0;
    const double invstd = 1.0 / sqrt(var + eps);
    const double sc = (double)gamma[r] * invstd;
    scale[r] = (float)sc;
    shift[r] = (float)((double)beta[r] - mean * sc);
    stats[3 * r] = mean;
    stats[3 * r + 1] = var;
    stats[3 * r + 2] = invstd;
}

// (dscale, dshift) -> d sums (R, 2) fp64, dgamma, dbeta.  raw variance <= 0 (clamped): no gradient through the variance.
__global__ __launch_bounds__(256) void trunk_finalize_bwd(const float *__restrict__ dscale, const float *__restrict__ dshift,
                                                          const float *__restrict__ gamma, const double *__restrict__ stats, double n, int R,
                                                          double *__restrict__ dsums, float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const double mean = stats[3 * r], var = stats[3 * r + 1], invstd = stats[3 * r + 2];
    const double ds = dscale[r], dh = dshift[r], g = gamma[r];
    dgamma[r] = (float)((ds - dh * mean) * invstd);
    dbeta[r] = (float)dh;
    const double dinv = (ds - dh * mean) * g;
    double dmean = -dh * g * invstd;
    const double dvar = var > 0.0 ? dinv * -0.5 * invstd * invstd * invstd : 0.0;
    dmean += dvar * -2.0 * mean;
    dsums[2 * r] = dmean / n;
    dsums[2 * r + 1] = dvar / n;
}

// dWG (R, K + 1): columns k < K = d(sum y^2) Wd[r][k], column K = d(sum y)
__global__ __launch_bounds__(256) void trunk_dwg(const double *__restrict__ dsums, const double *__restrict__ Wd, int R, int K,
                                                 double *__restrict__ dWG)
{
    const long n = (long)R * (K + 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = i / (K + 1);
        const int k = (int)(i - r * (K + 1));
        dWG[i] = k < K ? dsums[2 * r + 1] * Wd[r * K + k] : dsums[2 * r];
    }
}

inline bool bad_shape(int B, int C, long HW) { return B <= 0 || C <= 0 || HW <= 0 || C > 65535 || (double)B * ((HW + CH - 1) / CH) > 2.0e9; }

}  // namespace

extern "C" {

size_t dcd_bn_workspace_bytes(int C) { return C > 0 ? (size_t)C * SMAX * 2 * sizeof(double) : 0; }

int dcd_bn_stats(void *stream_, const float *x, int B, int C, int64_t HW, double *stats, void *ws, size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !stats || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    const Plane g = make_plane(B, C, HW);
    const int S = sync_slices(g);
    if (S == 1) {
        hipLaunchKernelGGL(bn_partial, dim3(1, C), dim3(BT), 0, stream, x, g, 1, stats);
    } else {
        hipLaunchKernelGGL(bn_partial, dim3(S, C), dim3(BT), 0, stream, x, g, S, (double *)ws);
        hipLaunchKernelGGL(bn_combine, dim3(C), dim3(64), 0, stream, (const double *)ws, S, stats, (const float *)nullptr, (float *)nullptr,
                           (float *)nullptr);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_channel_sums(void *stream_, const float *x, int B, int C, int64_t HW, float *sums, void *ws, size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !sums || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    const Plane g = make_plane(B, C, HW);
    const int S = slices(g);
    // partials: the first C * S doubles; arrival counters: C words at the start of the workspace's second half
    unsigned *arrivals = reinterpret_cast<unsigned *>(reinterpret_cast<double *>(ws) + (size_t)C * SMAX);
    // The one-launch form pays an agent-scope release (an L2 write-back on gfx950) per workgroup: fine for a few hundred
    // workgroups, dearer than a second launch beyond (27 channels x 152 slices @ 96x320 x 8: 45 us against 12).  Same partials,
    // same summation order, same result either way.  DCD_CHANNEL_SUM_ONE_LAUNCH=1|0 pins the form (A/B timing).
    static const int pin = dcd_env("DCD_CHANNEL_SUM_ONE_LAUNCH") ? atoi(dcd_env("DCD_CHANNEL_SUM_ONE_LAUNCH")) : -1;
    const bool one = pin >= 0 ? pin != 0 : (long)S * C <= 256;
    if (one) {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(S, C), dim3(BT), 0, stream, x, g, S, (double *)ws, arrivals, sums);
    } else {
        hipLaunchKernelGGL(channel_sum_kernel, dim3(S, C), dim3(BT), 0, stream, x, g, S, (double *)ws, (unsigned *)nullptr, sums);
        hipLaunchKernelGGL(channel_sum_combine, dim3(C), dim3(64), 0, stream, (const double *)ws, S, sums);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_bn_train_apply(void *stream_, const float *x, const float *residual, const float *weight, const float *bias,
                       const double *stats, double count, float *running_mean, float *running_var,
                       int64_t *num_batches_tracked, float momentum, float eps, int relu, float *y, float *save_mean,
                       float *save_invstd, int B, int C, int64_t HW)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !y || !stats || !save_mean || !save_invstd || bad_shape(B, C, HW) || !(count >= 1.0)) return DCD_ERR_BAD_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return DCD_ERR_BAD_ARG;
    const Plane g = make_plane(B, C, HW);
    hipLaunchKernelGGL(bn_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, x, residual, weight, bias, stats, (const double *)nullptr, 0,
                       count, running_mean, running_var, (long long *)num_batches_tracked, momentum, eps, relu, y, save_mean,
                       save_invstd, g);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_bn_eval_apply(void *stream_, const float *x, const float *residual, const float *weight, const float *bias,
                      const float *running_mean, const float *running_var, float eps, int relu, float *y, int B, int C,
                      int64_t HW)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !y || !running_mean || !running_var || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    const Plane g = make_plane(B, C, HW);
    hipLaunchKernelGGL(bn_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, x, residual, weight, bias, (const double *)nullptr,
                       (const double *)nullptr, 0, 1.0,
                       const_cast<float *>(running_mean), const_cast<float *>(running_var), (long long *)nullptr, 0.f, eps,
                       relu, y, (float *)nullptr, (float *)nullptr, g);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static int bn_backward_stats_run(hipStream_t stream, const float *grad_y, const float *y, const float *x, const float *save_mean,
                                 const float *save_invstd, int B, int C, int64_t HW, double *sums, float *grad_weight, float *grad_bias,
                                 void *ws, size_t ws_bytes, const float *mask_weight = nullptr, const float *mask_bias = nullptr,
                                 int mask_from_x = 0)
{
    (void)hipGetLastError();
    if (!grad_y || !x || !save_mean || !sums || bad_shape(B, C, HW) || ((grad_weight || grad_bias || mask_from_x) && !save_invstd))
        return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    const Plane g = make_plane(B, C, HW);
    const int S = sync_slices(g);
    if (S == 1) {
        hipLaunchKernelGGL(bn_bwd_partial, dim3(1, C), dim3(BT), 0, stream, grad_y, y, x, save_mean, g, 1, sums, save_invstd, grad_weight,
                           grad_bias, mask_weight, mask_bias, mask_from_x);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial, dim3(S, C), dim3(BT), 0, stream, grad_y, y, x, save_mean, g, S, (double *)ws, save_invstd,
                           (float *)nullptr, (float *)nullptr, mask_weight, mask_bias, mask_from_x);
        hipLaunchKernelGGL(bn_combine, dim3(C), dim3(64), 0, stream, (const double *)ws, S, sums, save_invstd, grad_weight, grad_bias);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_bn_backward_stats(void *stream_, const float *grad_y, const float *y, const float *x, const float *save_mean, int B,
                          int C, int64_t HW, double *sums, void *ws, size_t ws_bytes)
{
    return bn_backward_stats_run((hipStream_t)stream_, grad_y, y, x, save_mean, nullptr, B, C, HW, sums, nullptr, nullptr, ws, ws_bytes);
}

int dcd_bn_backward_stats_params(void *stream_, const float *grad_y, const float *y, const float *x, const float *save_mean,
                                 const float *save_invstd, int B, int C, int64_t HW, double *sums, float *grad_weight,
                                 float *grad_bias, void *ws, size_t ws_bytes)
{
    return bn_backward_stats_run((hipStream_t)stream_, grad_y, y, x, save_mean, save_invstd, B, C, HW, sums, grad_weight, grad_bias, ws,
                                 ws_bytes);
}

int dcd_bn_backward_stats_params_relu_from_x(void *stream_, const float *grad_y, const float *x, const float *weight, const float *bias,
                                             const float *save_mean, const float *save_invstd, int B, int C, int64_t HW, double *sums,
                                             float *grad_weight, float *grad_bias, void *ws, size_t ws_bytes)
{
    return bn_backward_stats_run((hipStream_t)stream_, grad_y, nullptr, x, save_mean, save_invstd, B, C, HW, sums, grad_weight, grad_bias, ws,
                                 ws_bytes, weight, bias, 1);
}

int dcd_bn_backward_apply_relu_from_x(void *stream_, const float *grad_y, const float *x, const float *weight, const float *bias,
                                      const float *save_mean, const float *save_invstd, const double *sums, double count, float *grad_x,
                                      int B, int C, int64_t HW)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_y || !x || !save_mean || !save_invstd || !sums || !grad_x || bad_shape(B, C, HW) || !(count >= 1.0)) return DCD_ERR_BAD_ARG;
    const Plane g = make_plane(B, C, HW);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, grad_y, (const float *)nullptr, x, weight, save_mean, save_invstd,
                       sums, (const double *)nullptr, 0, count, grad_x, (float *)nullptr, (float *)nullptr, (float *)nullptr, g, bias, 1);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_bn_backward_apply(void *stream_, const float *grad_y, const float *y, const float *x, const float *weight,
                          const float *save_mean, const float *save_invstd, const double *sums, double count,
                          float *grad_x, float *grad_residual, float *grad_weight, float *grad_bias, int B, int C, int64_t HW)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !save_mean || !save_invstd || !sums || !grad_x || bad_shape(B, C, HW) || !(count >= 1.0))   // grad_y NULL = zeros
        return DCD_ERR_BAD_ARG;
    const Plane g = make_plane(B, C, HW);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, grad_y, y, x, weight, save_mean, save_invstd, sums,
                       (const double *)nullptr, 0, count, grad_x, grad_residual, grad_weight, grad_bias, g);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

/* Local (single-rank) training forward: statistics + apply in two launches, no combine pass, no stats tensor. */
int dcd_bn_train_forward(void *stream_, const float *x, const float *residual, const float *weight, const float *bias,
                         float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                         int relu, float *y, float *save_mean, float *save_invstd, int B, int C, int64_t HW, void *ws,
                         size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !y || !save_mean || !save_invstd || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    if (small_channels(B, C, HW)) {
        hipLaunchKernelGGL(bn_small_fwd, dim3(C), dim3(BT), 0, stream, x, residual, weight, bias, running_mean, running_var,
                           (long long *)num_batches_tracked, momentum, eps, relu, y, save_mean, save_invstd, B, C, (int)(HW / 4));
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    const Plane g = make_plane(B, C, HW);
    const int S = slices(g);
    hipLaunchKernelGGL(bn_partial, dim3(S, C), dim3(BT), 0, stream, x, g, S, (double *)ws);
    hipLaunchKernelGGL(bn_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, x, residual, weight, bias, (const double *)nullptr,
                       (const double *)ws, S, (double)B * (double)HW, running_mean, running_var, (long long *)num_batches_tracked,
                       momentum, eps, relu, y, save_mean, save_invstd, g);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

/* Local backward: sums + apply in two launches. */
int dcd_bn_backward(void *stream_, const float *grad_y, const float *y, const float *x, const float *weight,
                    const float *save_mean, const float *save_invstd, float *grad_x, float *grad_residual, float *grad_weight,
                    float *grad_bias, int B, int C, int64_t HW, void *ws, size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_y || !x || !save_mean || !save_invstd || !grad_x || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    if (small_channels(B, C, HW)) {
        hipLaunchKernelGGL(bn_small_bwd, dim3(C), dim3(BT), 0, stream, grad_y, y, x, weight, save_mean, save_invstd, grad_x,
                           grad_residual, grad_weight, grad_bias, B, C, (int)(HW / 4));
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    const Plane g = make_plane(B, C, HW);
    const int S = slices(g);
    hipLaunchKernelGGL(bn_bwd_partial, dim3(S, C), dim3(BT), 0, stream, grad_y, y, x, save_mean, g, S, (double *)ws);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, grad_y, y, x, weight, save_mean, save_invstd,
                       (const double *)nullptr, (const double *)ws, S, (double)B * (double)HW, grad_x, grad_residual, grad_weight,
                       grad_bias, g);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_bn_backward_relu_from_x(void *stream_, const float *grad_y, const float *x, const float *weight, const float *bias,
                                const float *save_mean, const float *save_invstd, float *grad_x, float *grad_weight, float *grad_bias,
                                int B, int C, int64_t HW, void *ws, size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_y || !x || !save_mean || !save_invstd || !grad_x || bad_shape(B, C, HW)) return DCD_ERR_BAD_ARG;
    if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
    if (small_channels(B, C, HW)) {
        hipLaunchKernelGGL(bn_small_bwd, dim3(C), dim3(BT), 0, stream, grad_y, (const float *)nullptr, x, weight, save_mean, save_invstd,
                           grad_x, (float *)nullptr, grad_weight, grad_bias, B, C, (int)(HW / 4), bias, 1);
        return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
    }
    const Plane g = make_plane(B, C, HW);
    const int S = slices(g);
    hipLaunchKernelGGL(bn_bwd_partial, dim3(S, C), dim3(BT), 0, stream, grad_y, (const float *)nullptr, x, save_mean, g, S, (double *)ws,
                       save_invstd, (float *)nullptr, (float *)nullptr, weight, bias, 1);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(B * g.cpp, C), dim3(BT), 0, stream, grad_y, (const float *)nullptr, x, weight, save_mean,
                       save_invstd, (const double *)nullptr, (const double *)ws, S, (double)B * (double)HW, grad_x, (float *)nullptr,
                       grad_weight, grad_bias, g, bias, 1);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

/* BN (+ReLU) in training mode evaluated at listed positions only.  Statistics over the whole tensor as usual: either
 * `stats` (C x 2 combined sums, e.g. all-reduced) is given, or (stats == NULL) they are computed here. */
int dcd_bn_at_forward(void *stream_, const float *x, const int64_t *pos, const float *weight, const float *bias,
                      const double *stats, double count, float *running_mean, float *running_var,
                      int64_t *num_batches_tracked, float momentum, float eps, int relu, float *x_at, float *y_at,
                      float *save_mean, float *save_invstd, int B, int C, int64_t HW, int N, void *ws, size_t ws_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !pos || !x_at || !y_at || !save_mean || !save_invstd || bad_shape(B, C, HW) || N <= 0 || !(count >= 1.0))
        return DCD_ERR_BAD_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return DCD_ERR_BAD_ARG;
    const Plane g = make_plane(B, C, HW);
    int S = 0;
    if (!stats) {
        if (!ws || ws_bytes < dcd_bn_workspace_bytes(C)) return DCD_ERR_WORKSPACE;
        S = slices(g);
        hipLaunchKernelGGL(bn_partial, dim3(S, C), dim3(BT), 0, stream, x, g, S, (double *)ws);
    }
    hipLaunchKernelGGL(bn_at_forward, dim3(C), dim3(BT), 0, stream, x, pos, weight, bias, stats, stats ? (const double *)nullptr : (const double *)ws,
                       S, count, running_mean, running_var, (long long *)num_batches_tracked, momentum, eps, relu, x_at, y_at,
                       save_mean, save_invstd, B, C, (long)HW, N);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

/* sums (C x 2 doubles) and the scaled sparse gradient dzk (B*N, C) of the entries; then dcd_bn_backward_apply with
 * grad_y = NULL (dense part of grad_x) and a scatter-add of dzk finish the backward. */
int dcd_bn_at_backward_sums(void *stream_, const float *grad_at, const float *x_at, const float *y_at, const float *weight,
                            const float *save_mean, const float *save_invstd, int relu, int total, int C, double *sums,
                            float *dzk)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!grad_at || !x_at || !y_at || !save_mean || !save_invstd || !sums || !dzk || total <= 0 || C <= 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_at_backward_sums, dim3(C), dim3(BT), 0, stream, grad_at, x_at, y_at, weight, save_mean, save_invstd, relu,
                       total, C, sums, dzk);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_trunk_row_sums(void *stream_, const double *WG, const double *Wd, int R, int K, double *sums)
{
    (void)hipGetLastError();
    if (!WG || !Wd || !sums || R <= 0 || K <= 0) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(trunk_row_sums, dim3(R), dim3(64), 0, (hipStream_t)stream_, WG, Wd, K, sums);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_trunk_finalize_forward(void *stream_, const double *sums, const float *gamma, const float *beta, double count, double eps, int R,
                               float *scale, float *shift, double *stats)
{
    (void)hipGetLastError();
    if (!sums || !gamma || !beta || !scale || !shift || !stats || R <= 0 || !(count >= 1.0)) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(trunk_finalize_fwd, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream_, sums, gamma, beta, count, eps, R, scale,
                       shift, stats);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_trunk_finalize_backward(void *stream_, const float *grad_scale, const float *grad_shift, const float *gamma, const double *stats,
                                double count, int R, double *grad_sums, float *grad_gamma, float *grad_beta)
{
    (void)hipGetLastError();
    if (!grad_scale || !grad_shift || !gamma || !stats || !grad_sums || !grad_gamma || !grad_beta || R <= 0 || !(count >= 1.0))
        return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(trunk_finalize_bwd, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream_, grad_scale, grad_shift, gamma, stats,
                       count, R, grad_sums, grad_gamma, grad_beta);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_trunk_grad_wg(void *stream_, const double *grad_sums, const double *Wd, int R, int K, double *grad_WG)
{
    (void)hipGetLastError();
    if (!grad_sums || !Wd || !grad_WG || R <= 0 || K <= 0) return DCD_ERR_BAD_ARG;
    const long n = (long)R * (K + 1);
    hipLaunchKernelGGL(trunk_dwg, dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, (hipStream_t)stream_,
                       grad_sums, Wd, R, K, grad_WG);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
