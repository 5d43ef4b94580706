// optim.hip -- the optimizer end of the train step for gfx950: gradient-norm clip + AdamW over a LIST of tensors.
//
// What it replaces (DGDE/engine/trainer.py:144-147: clip_grad_norm_(model.parameters(), 15) ; optimizer.step(), AdamW with one
// parameter group per parameter, DGDE/solver/__init__.py:10-62): in rounds 1-5 `torch._foreach_norm` + `_foreach_mul_` + the library's
// fused AdamW -- 29 launches, 0.43 ms per step at 2 TB/s: its multi-tensor launches carry at most 320 blocks of 512 threads, 64 K
// elements each, i.e. about two blocks per CU for a kernel that streams 28 bytes per parameter.
// Here: the tensor pointers travel BY VALUE in the kernel arguments (nothing for a captured step to re-read from the host, no
// device-side table to keep alive), 4 096 elements per block, three kernels:
//   adam_grad_sqnorm  sum of g^2 per block of 8 192 (fp32 per thread, fp64 across the block), one partial per block
//   adam_finalize     partials in a fixed order -> total norm, clip coefficient, non-finite flag; step counters += 1
//   adamw_apply       g *= coefficient (written back only when it clips, like clip_grad_norm_'s in-place scale), then the library's
//                     fused AdamW arithmetic, in the same order and the same mixed double / float types
// Results equal the library's to rounding (tests/test_gpu_optim.py); a non-finite norm leaves parameters, moments and step counters
// untouched (the library's found_inf contract).

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"

namespace {

constexpr int OPT_NT = 256;               // threads per block
constexpr int OPT_CHUNK = 4096;           // elements per block of the update (four float4 per thread and tensor)
constexpr int OPT_NCHUNK = 8192;          // elements per block of the norm (one partial each: 2 048 made 10 K partials and a 38 us pass)
constexpr int OPT_MAXT = 80;              // tensors per launch: 80 * (5 pointers + count + first block) = 3.8 KB of kernel arguments
constexpr int OPT_MAXG = 240;             // gradients per launch of the norm: 240 * (pointer + count + first block) = 3.8 KB
constexpr int OPT_MAXS = 448;             // step counters per finalize launch (3.5 KB)

struct GradTable {
    const float *g[OPT_MAXG];
    int n[OPT_MAXG];
    int blk0[OPT_MAXG + 1];               // first block of tensor t in this launch; blk0[count] = blocks of the launch
    int count;
    int part0;                            // index of this launch's first partial
};

struct AdamTable {
    float *p[OPT_MAXT];
    float *g[OPT_MAXT];
    float *m[OPT_MAXT];
    float *v[OPT_MAXT];
    const float *step[OPT_MAXT];
    int n[OPT_MAXT];
    int blk0[OPT_MAXT + 1];
    int count;
};

struct StepTable {
    float *step[OPT_MAXS];
    int count;
};

__device__ __forceinline__ int tensor_of_block(const int *blk0, int count, int b)
{
    int lo = 0, hi = count - 1;           // largest t with blk0[t] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk0[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(OPT_NT) void adam_grad_sqnorm(GradTable t, double *__restrict__ partials)
{
    const int b = blockIdx.x;
    const int ti = tensor_of_block(t.blk0, t.count, b);
    const float *g = t.g[ti];
    const int n = t.n[ti];
    const int base = (b - t.blk0[ti]) * OPT_NCHUNK;
    float s = 0.f;
    if (((uintptr_t)g & 15) == 0) {
#pragma unroll
        for (int k = 0; k < OPT_NCHUNK / (4 * OPT_NT); ++k) {
            const int i = base + 4 * (threadIdx.x + OPT_NT * k);
            if (i + 3 < n) {
                const float4 x = *reinterpret_cast<const float4 *>(g + i);
                s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            } else {
                for (int j = i; j < n && j < i + 4; ++j) s += g[j] * g[j];
            }
        }
    } else {
        for (int i = base + threadIdx.x; i < n && i < base + OPT_NCHUNK; i += OPT_NT) s += g[i] * g[i];
    }
    double d = (double)s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    __shared__ double w[OPT_NT / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int k = 0; k < OPT_NT / 64; ++k) sum += w[k];
        partials[t.part0 + b] = sum;
    }
}

// scal[0] = total norm, scal[1] = clip coefficient (1 when max_norm <= 0), scal[2] = 1 when the norm is not finite
__global__ __launch_bounds__(OPT_NT) void adam_finalize(const double *__restrict__ partials, int npart, float max_norm,
                                                         float *__restrict__ scal)
{
    __shared__ double w[OPT_NT];
    double s = 0.0;
    for (int i = threadIdx.x; i < npart; i += OPT_NT) s += partials[i];       // fixed assignment, fixed tree: reproducible
    w[threadIdx.x] = s;
    __syncthreads();
    for (int o = OPT_NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) w[threadIdx.x] += w[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(w[0]);
        float coef = 1.f;
        if (max_norm > 0.f) {
            coef = max_norm / (total + 1e-6f);                                 // clip_grad_norm_: clip_coef = max_norm / (total + 1e-6)
            if (coef > 1.f) coef = 1.f;                                        //                  clamped to 1
        }
        const bool finite = isfinite(total);
        scal[0] = total;
        scal[1] = finite ? coef : 1.f;
        scal[2] = finite ? 0.f : 1.f;
    }
}

__global__ __launch_bounds__(OPT_NT) void adam_advance_steps(StepTable t, const float *__restrict__ scal)
{
    const int i = blockIdx.x * OPT_NT + threadIdx.x;
    if (i < t.count && scal[2] == 0.f) *t.step[i] += 1.f;
}

struct AdamMath {
    double lr, beta1, beta2, wd, eps;
    float bc1, bc2_sqrt, coef;
    bool store_g;
    __device__ __forceinline__ void operator()(float &p, float &g, float &m, float &v) const
    {
        // the library's adam_math (ADAMW mode, no amsgrad, no maximize): double hyper-parameters against float operands
        float grad = g * coef;
        if (store_g) g = grad;
        float param = p;
        if (wd != 0.0) param = (float)((double)param - lr * wd * (double)param);
        const float exp_avg = (float)(beta1 * (double)m + (1.0 - beta1) * (double)grad);
        const float exp_avg_sq = (float)(beta2 * (double)v + (1.0 - beta2) * (double)grad * (double)grad);
        const float step_size = (float)(lr / (double)bc1);
        const float denom = (float)((double)(sqrtf(exp_avg_sq) / bc2_sqrt) + eps);
        param -= step_size * exp_avg / denom;
        p = param; m = exp_avg; v = exp_avg_sq;
    }
};

__global__ __launch_bounds__(OPT_NT) void adamw_apply(AdamTable t, const float *__restrict__ lr_ptr, double beta1, double beta2,
                                                      double eps, double wd, const float *__restrict__ scal)
{
    if (scal[2] != 0.f) return;                                 // non-finite gradient norm: nothing moves
    const int b = blockIdx.x;
    const int ti = tensor_of_block(t.blk0, t.count, b);
    float *p = t.p[ti], *g = t.g[ti], *m = t.m[ti], *v = t.v[ti];
    const int n = t.n[ti];
    const int base = (b - t.blk0[ti]) * OPT_CHUNK;
    AdamMath f;
    f.lr = (double)*lr_ptr; f.beta1 = beta1; f.beta2 = beta2; f.wd = wd; f.eps = eps;
    const double step = (double)*t.step[ti];                    // already advanced by adam_advance_steps
    f.bc1 = (float)(1.0 - pow(beta1, step));
    f.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, step));
    f.coef = scal[1];
    f.store_g = f.coef != 1.f;
    const bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (aligned) {
#pragma unroll
        for (int k = 0; k < OPT_CHUNK / (4 * OPT_NT); ++k) {
            const int i = base + 4 * (threadIdx.x + OPT_NT * k);
            if (i + 3 < n) {
                float4 pp = *reinterpret_cast<float4 *>(p + i), gg = *reinterpret_cast<float4 *>(g + i);
                float4 mm = *reinterpret_cast<float4 *>(m + i), vv = *reinterpret_cast<float4 *>(v + i);
                f(pp.x, gg.x, mm.x, vv.x); f(pp.y, gg.y, mm.y, vv.y); f(pp.z, gg.z, mm.z, vv.z); f(pp.w, gg.w, mm.w, vv.w);
                *reinterpret_cast<float4 *>(p + i) = pp;
                *reinterpret_cast<float4 *>(m + i) = mm;
                *reinterpret_cast<float4 *>(v + i) = vv;
                if (f.store_g) *reinterpret_cast<float4 *>(g + i) = gg;
            } else {
                for (int j = i; j < n && j < i + 4; ++j) {
                    float gj = g[j];
                    f(p[j], gj, m[j], v[j]);
                    if (f.store_g) g[j] = gj;
                }
            }
        }
    } else {
        for (int i = base + threadIdx.x; i < n && i < base + OPT_CHUNK; i += OPT_NT) {
            float gi = g[i];
            f(p[i], gi, m[i], v[i]);
            if (f.store_g) g[i] = gi;
        }
    }
}

inline int blocks_of(int64_t n) { return (int)((n + OPT_CHUNK - 1) / OPT_CHUNK); }
inline int norm_blocks_of(int64_t n) { return (int)((n + OPT_NCHUNK - 1) / OPT_NCHUNK); }

}  // namespace

extern "C" {

size_t dcd_clip_adamw_workspace_bytes(int ntensors, const int64_t *numel)
{
    if (ntensors <= 0 || !numel) return 0;
    int64_t blocks = 0;
    for (int i = 0; i < ntensors; ++i) {
        if (numel[i] < 0 || numel[i] >= (1ll << 31)) return 0;
        blocks += norm_blocks_of(numel[i]);
    }
    return (size_t)blocks * sizeof(double) + 16 * sizeof(float);
}

// Gradient norm of ALL listed tensors (the clip is over the whole model, whatever the parameter groups): scal[0..2] on the device.
int dcd_clip_grad_norm_scalars(void *stream_, int ntensors, const void *const *grads, const int64_t *numel, float max_norm, void *workspace,
                               size_t workspace_bytes, float *scal)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (ntensors < 0 || (ntensors && (!grads || !numel)) || !workspace || !scal) return DCD_ERR_BAD_ARG;
    const size_t need = dcd_clip_adamw_workspace_bytes(ntensors, numel);
    if (ntensors && (need == 0 || workspace_bytes < need)) return DCD_ERR_WORKSPACE;
    double *partials = (double *)workspace;
    int part0 = 0;
    for (int i = 0; i < ntensors;) {
        GradTable t;
        t.count = 0;
        t.part0 = part0;
        int blocks = 0;
        for (; i < ntensors && t.count < OPT_MAXG; ++i) {
            if (numel[i] == 0) continue;
            if (!grads[i]) return DCD_ERR_BAD_ARG;
            t.g[t.count] = (const float *)grads[i];
            t.n[t.count] = (int)numel[i];
            t.blk0[t.count] = blocks;
            blocks += norm_blocks_of(numel[i]);
            ++t.count;
        }
        if (!t.count) break;
        t.blk0[t.count] = blocks;
        hipLaunchKernelGGL(adam_grad_sqnorm, dim3(blocks), dim3(OPT_NT), 0, stream, t, partials);
        part0 += blocks;
    }
    hipLaunchKernelGGL(adam_finalize, dim3(1), dim3(OPT_NT), 0, stream, (const double *)partials, part0, max_norm, scal);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

// One parameter group: every tensor's step counter += 1, then p / m / v (and g when the clip coefficient is below 1) updated in place.
// lr: a float on the device (the schedulers write it in place); scal: what dcd_clip_grad_norm_scalars left.
int dcd_adamw_apply(void *stream_, int ntensors, void *const *params, void *const *grads, void *const *exp_avg, void *const *exp_avg_sq,
                    void *const *steps, const int64_t *numel, const float *lr, double beta1, double beta2, double eps, double weight_decay,
                    const float *scal)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (ntensors == 0) return DCD_OK;
    if (ntensors < 0 || !params || !grads || !exp_avg || !exp_avg_sq || !steps || !numel || !lr || !scal) return DCD_ERR_BAD_ARG;
    for (int i = 0; i < ntensors; ++i)
        if (numel[i] < 0 || numel[i] >= (1ll << 31) || !steps[i] || (numel[i] && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i])))
            return DCD_ERR_BAD_ARG;
    for (int t0 = 0; t0 < ntensors; t0 += OPT_MAXS) {
        StepTable st;
        st.count = 0;
        for (int i = t0; i < ntensors && st.count < OPT_MAXS; ++i) st.step[st.count++] = (float *)steps[i];
        hipLaunchKernelGGL(adam_advance_steps, dim3((st.count + OPT_NT - 1) / OPT_NT), dim3(OPT_NT), 0, stream, st, scal);
    }
    for (int i = 0; i < ntensors;) {
        AdamTable t;
        t.count = 0;
        int blocks = 0;
        for (; i < ntensors && t.count < OPT_MAXT; ++i) {
            if (numel[i] == 0) continue;
            const int k = t.count++;
            t.p[k] = (float *)params[i]; t.g[k] = (float *)grads[i]; t.m[k] = (float *)exp_avg[i]; t.v[k] = (float *)exp_avg_sq[i];
            t.step[k] = (const float *)steps[i];
            t.n[k] = (int)numel[i];
            t.blk0[k] = blocks;
            blocks += blocks_of(numel[i]);
        }
        if (!t.count) break;
        t.blk0[t.count] = blocks;
        hipLaunchKernelGGL(adamw_apply, dim3(blocks), dim3(OPT_NT), 0, stream, t, lr, beta1, beta2, eps, weight_decay, scal);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
