// stem.hip -- the low-channel, full-resolution convolutions of DLA-34's stem on the fp32 matrix pipe (gfx950):
//   base_layer  Conv2d(3, 16, 7, stride 1, padding 3, bias=False)     DGDE/model/backbone/dla_dcn.py:236-240
//   level0      Conv2d(16, 16, 3, stride 1, padding 1, bias=False)    DGDE/model/backbone/dla_dcn.py:241-242,268-283
// forward, input gradient (level0) and weight gradient.  At bs 8 / 384x1280 each moves 0.3-0.5 GB and does 18 GFLOP: a
// memory-bound job (0.08-0.13 ms at 4 TB/s) that the stock solvers take 0.46-1.37 ms for (4.0 ms per step together,
// tools/time_stem.py) -- 16 channels fill a quarter of their 64-wide tiles.  v_mfma_f32_16x16x4_f32 fits 16 channels exactly.
//
// Forward / input gradient: D[16 o x 16 pixels] += W[16 o x 4 j] . patch[4 j x 16 pixels], j = (tap, c) over NJ = Cin K K.
//   One workgroup = 4 waves owns an 8 x 64 pixel tile; the raw input window (Cin x (8+K-1) x 72) is staged in LDS once, the
//   B operand of every step is ONE ds_read_b32 straight from it (lane = (pixel, j mod 4); plane stride 16 * odd: the four
//   j of a step are four channels -> 64 distinct banks), the A operands (weights, pre-swizzled to lane order) live in
//   registers for the whole kernel.  A wave runs four independent 16-pixel accumulators at a time.
// Weight gradient: D_nb[16 o x 16 j] += dY[16 o x 4 pixels] . patch[4 pixels x 16 j] for the NB = ceil(NJ/16) column
//   blocks (level0: block = tap, j = channel); both operands are "lane = channel" reads from LDS (plane strides 4 * odd);
//   persistent workgroups keep their accumulators over all their tiles, then partials -> reduce kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"
#include "zero_fill.h"
#include "lds_limit.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ST_R = 8;              // tile rows
constexpr int ST_C = 64;             // tile columns
constexpr int ST_RS = 72;            // staged columns c0-4 .. c0+67
constexpr int ST_DPLANE = ST_R * ST_C + 4;       // 516 = 4 * 129

constexpr int round_odd_mult(int x, int m)       // smallest m * odd >= x
{
    int p = (x + m - 1) / m;
    if ((p & 1) == 0) ++p;
    return p * m;
}

template <int CI, int KS>
struct StemCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int TAPS = KS * KS;
    static constexpr int NJ = CI * TAPS;
    static constexpr int NSTEP = (NJ + 3) / 4;
    static constexpr int NB = (NJ + 15) / 16;
    static constexpr int ROWS = ST_R + KS - 1;
    static constexpr int PLANE_F = round_odd_mult(ROWS * ST_RS, 16);      // forward: 16 * odd
    static constexpr int PLANE_W = round_odd_mult(ROWS * ST_RS, 4);       // weight gradient: 4 * odd
};

// LDS offset of contraction index j = tap * CI + c relative to the output pixel's window position
template <int CI, int KS, int PLANE>
__host__ __device__ constexpr int joff(int j)
{
    return j >= CI * KS * KS ? 0 : (j % CI) * PLANE + ((j / CI) / KS) * ST_RS + (j / CI) % KS;
}

// wp[s][lane]: A operand of step s for lane (i = lane & 15, k = lane >> 4): W_eff[i][j = 4 s + k]
//   mode 0 (forward):        W_eff[o][(tap, c)] = w[o][c][ty][tx]
//   mode 1 (input gradient): W_eff[c][(tap, o)] = w[o][c][K-1-ty][K-1-tx]          (CI == 16 only: square)
template <int CI, int KS>
__global__ void stem_prep_weights(const float *__restrict__ w, float *__restrict__ wp, int mode)
{
    using C = StemCfg<CI, KS>;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C::NSTEP * 64) return;
    const int s = e >> 6, l = e & 63;
    const int i = l & 15, k = l >> 4;
    const int j = 4 * s + k;
    float v = 0.f;
    if (j < C::NJ) {
        const int tap = j / CI, c = j - tap * CI;
        const int ty = tap / KS, tx = tap - ty * KS;
        v = mode == 0 ? w[((i * CI + c) * KS + ty) * KS + tx] : w[((c * CI + i) * KS + (KS - 1 - ty)) * KS + (KS - 1 - tx)];
    }
    wp[e] = v;
}

// All loads of a thread are issued before the first LDS store (one exposed global-memory latency per tile, not one per quad).
template <int CI, int KS, int PLANE>
__device__ __forceinline__ void stage_window(float *__restrict__ win, const float *__restrict__ xb, int H, int W, int r0, int c0, int tid)
{
    using C = StemCfg<CI, KS>;
    constexpr int Q = ST_RS / 4;                                  // 18 dwordx4 per row
    constexpr int N = CI * C::ROWS * Q;
    constexpr int IT = (N + 255) / 256;
    const int HW = H * W;
    f32x4 v[IT];
    int dst[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = tid + 256 * i;
        const int ch = e / (C::ROWS * Q), rem = e - ch * (C::ROWS * Q);
        const int row = rem / Q, q = rem - row * Q;
        const int yy = r0 - C::PAD + row, xx = c0 - 4 + 4 * q;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        dst[i] = e < N ? ch * PLANE + row * ST_RS + 4 * q : -1;
        if (e < N && yy >= 0 && yy < H && xx >= 0 && xx < W) v[i] = *reinterpret_cast<const f32x4 *>(xb + (size_t)ch * HW + (size_t)yy * W + xx);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i)
        if (dst[i] >= 0) *reinterpret_cast<f32x4 *>(win + dst[i]) = v[i];
}

// grid = (tiles_x * tiles_y, B), block = 256.  x: (B, CI, H, W) -> y: (B, 16, H, W)
template <int CI, int KS>
__global__ __launch_bounds__(256) void stem_fwd_f32(const float *__restrict__ x, const float *__restrict__ wp, float *__restrict__ y,
                                                    int H, int W, int tiles_x)
{
    using C = StemCfg<CI, KS>;
    constexpr int PLANE = C::PLANE_F;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // CI * PLANE
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, k = lane >> 4;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.y;
    const int r0 = ty * ST_R, c0 = tx * ST_C;
    const int HW = H * W;

    float a[C::NSTEP];
#pragma unroll
    for (int s = 0; s < C::NSTEP; ++s) a[s] = wp[s * 64 + lane];

    stage_window<CI, KS, PLANE>(lds, x + (size_t)b * CI * HW, H, W, r0, c0, tid);
    __syncthreads();

#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
        const int row = 2 * wave + rr;
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *base = lds + row * ST_RS + i16 + 4 - C::PAD;
#pragma unroll
        for (int s = 0; s < C::NSTEP; ++s) {
            int off;
            if constexpr (CI % 4 == 0) {
                off = joff<CI, KS, PLANE>(4 * s) + k * PLANE;                    // the four j of a step: four channels of one tap
            } else {
                const int o0 = joff<CI, KS, PLANE>(4 * s), o1 = joff<CI, KS, PLANE>(4 * s + 1);
                const int o2 = joff<CI, KS, PLANE>(4 * s + 2), o3 = joff<CI, KS, PLANE>(4 * s + 3);
                off = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : o3;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], base[off + 16 * g], acc[g], 0, 0, 0);
        }
        // D[i = 4 k + r][pixel i16]
        const int yy = r0 + row;
        if (yy < H) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int xx = c0 + 16 * g + i16;
                if (xx < W) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) y[((size_t)b * 16 + 4 * k + r) * HW + (size_t)yy * W + xx] = acc[g][r];
                }
            }
        }
    }
}

// persistent: grid = nwg, block = 256.  part[(wg * 4 + wave) * NB + nb][o][j16]
template <int CI, int KS>
__global__ __launch_bounds__(256) void stem_wrw_f32(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ part,
                                                    int B, int H, int W, int tiles_x, int tiles_y)
{
    using C = StemCfg<CI, KS>;
    constexpr int PLANE = C::PLANE_W;
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [CI * PLANE | 16 * ST_DPLANE]
    float *win = lds, *dyt = lds + CI * PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, k = lane >> 4;
    const int HW = H * W;
    const int total = B * tiles_x * tiles_y;

    int boff[C::NB];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) {
        const int j = nb * 16 + j16;
        const int jc = j < C::NJ ? j : 0;
        const int tap = jc / CI, c = jc - tap * CI;
        boff[nb] = c * PLANE + (tap / KS) * ST_RS + tap % KS + k + 4 - C::PAD;
    }
    f32x4 acc[C::NB];
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int b = t / (tiles_x * tiles_y), rem = t - b * (tiles_x * tiles_y);
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int r0 = ty * ST_R, c0 = tx * ST_C;
        __syncthreads();                                   // previous tile fully consumed
        stage_window<CI, KS, PLANE>(win, x + (size_t)b * CI * HW, H, W, r0, c0, tid);
        const float *gb = gy + (size_t)b * 16 * HW;
        {
            constexpr int ITD = 16 * ST_R * (ST_C / 4) / 256;          // 8 dwordx4 per thread
            f32x4 v[ITD];
#pragma unroll
            for (int i = 0; i < ITD; ++i) {
                const int e = tid + 256 * i;
                const int o = e / (ST_R * 16), rem2 = e - o * (ST_R * 16);
                const int row = rem2 >> 4, q = rem2 & 15;
                const int yy = r0 + row, xx = c0 + 4 * q;
                v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (yy < H && xx < W) v[i] = *reinterpret_cast<const f32x4 *>(gb + (size_t)o * HW + (size_t)yy * W + xx);
            }
#pragma unroll
            for (int i = 0; i < ITD; ++i) {
                const int e = tid + 256 * i;
                const int o = e / (ST_R * 16), rem2 = e - o * (ST_R * 16);
                *reinterpret_cast<f32x4 *>(dyt + o * ST_DPLANE + (rem2 >> 4) * ST_C + 4 * (rem2 & 15)) = v[i];
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr;
            const float *ap = dyt + j16 * ST_DPLANE + row * ST_C + k;
            const float *bp = win + row * ST_RS;
#pragma unroll 4
            for (int gx = 0; gx < 16; ++gx) {
                const float a = ap[4 * gx];
#pragma unroll
                for (int nb = 0; nb < C::NB; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[boff[nb] + 4 * gx], acc[nb], 0, 0, 0);
            }
        }
    }
    float *mine = part + ((size_t)blockIdx.x * 4 + wave) * C::NB * 256;
#pragma unroll
    for (int nb = 0; nb < C::NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[nb * 256 + (4 * k + r) * 16 + j16] = acc[nb][r];
}

// grid = (NB, 32), block = 256: each thread adds every 32nd partial of its (nb, o, j16) and atomically adds the sum to gw
template <int CI, int KS>
__global__ __launch_bounds__(256) void stem_wrw_reduce(const float *__restrict__ part, int nparts, float *__restrict__ gw)
{
    using C = StemCfg<CI, KS>;
    const int nb = blockIdx.x, e = threadIdx.x;
    const int o = e >> 4, j = nb * 16 + (e & 15);
    float s = 0.f;
    for (int p = blockIdx.y; p < nparts; p += gridDim.y) s += part[((size_t)p * C::NB + nb) * 256 + e];
    if (j < C::NJ) {
        const int tap = j / CI, c = j - tap * CI;
        atomicAdd(gw + (o * CI + c) * C::TAPS + tap, s);
    }
}

template <int CI, int KS>
int launch_fwd(hipStream_t stream, const float *input, const float *weight, float *output, int B, int H, int W, int backward_data,
               float *wp)
{
    using C = StemCfg<CI, KS>;
    static LdsLimit lds_limit;
    const size_t ldsb = (size_t)CI * C::PLANE_F * sizeof(float);
    if (!lds_limit.raise((int)ldsb, stem_fwd_f32<CI, KS>)) return DCD_ERR_LAUNCH;
    hipLaunchKernelGGL((stem_prep_weights<CI, KS>), dim3((C::NSTEP * 64 + 255) / 256), dim3(256), 0, stream, weight, wp, backward_data);
    const int tiles_x = (W + ST_C - 1) / ST_C, tiles_y = (H + ST_R - 1) / ST_R;
    hipLaunchKernelGGL((stem_fwd_f32<CI, KS>), dim3(tiles_x * tiles_y, B), dim3(256), ldsb, stream, input, (const float *)wp, output, H, W,
                       tiles_x);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

constexpr int ST_WRW_WGS = 512;

template <int CI, int KS>
int launch_wrw(hipStream_t stream, const float *input, const float *grad_output, float *grad_weight, int B, int H, int W, float *part)
{
    using C = StemCfg<CI, KS>;
    static LdsLimit lds_limit;
    const size_t ldsb = (size_t)(CI * C::PLANE_W + 16 * ST_DPLANE) * sizeof(float);
    if (!lds_limit.raise((int)ldsb, stem_wrw_f32<CI, KS>)) return DCD_ERR_LAUNCH;
    const int tiles_x = (W + ST_C - 1) / ST_C, tiles_y = (H + ST_R - 1) / ST_R;
    const int64_t total = (int64_t)B * tiles_x * tiles_y;
    const int nwg = total < ST_WRW_WGS ? (int)total : ST_WRW_WGS;
    if (!dcd_zero_fill(stream, grad_weight, (size_t)16 * C::NJ)) return DCD_ERR_LAUNCH;      // a launch, not a memset node: zero_fill.h
    hipLaunchKernelGGL((stem_wrw_f32<CI, KS>), dim3(nwg), dim3(256), ldsb, stream, input, grad_output, part, B, H, W, tiles_x, tiles_y);
    hipLaunchKernelGGL((stem_wrw_reduce<CI, KS>), dim3(C::NB, 32), dim3(256), 0, stream, (const float *)part, nwg * 4, grad_weight);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

bool stem_shape_ok(int B, int Cin, int H, int W, int Cout, int ksize)
{
    if (B <= 0 || H <= 0 || W <= 0 || (W & 3) || Cout != 16) return false;
    if ((int64_t)B * 16 * H * W >= (1ll << 40)) return false;
    return (Cin == 16 && ksize == 3) || (Cin == 3 && ksize == 7);
}

}  // namespace

extern "C" {

size_t dcd_conv_stem_workspace_bytes(int Cin, int Cout, int ksize)
{
    if (Cout != 16) return 0;
    if (Cin == 16 && ksize == 3) return (size_t)StemCfg<16, 3>::NSTEP * 64 * sizeof(float);
    if (Cin == 3 && ksize == 7) return (size_t)StemCfg<3, 7>::NSTEP * 64 * sizeof(float);
    return 0;
}

int dcd_conv_stem(void *stream_, const float *input, const float *weight, float *output, int B, int Cin, int H, int W, int Cout,
                  int ksize, int backward_data, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !weight || !output || !workspace || !stem_shape_ok(B, Cin, H, W, Cout, ksize)) return DCD_ERR_BAD_ARG;
    if (backward_data && Cin != 16) return DCD_ERR_BAD_ARG;
    if (workspace_bytes < dcd_conv_stem_workspace_bytes(Cin, Cout, ksize)) return DCD_ERR_WORKSPACE;
    if (Cin == 16) return launch_fwd<16, 3>(stream, input, weight, output, B, H, W, backward_data ? 1 : 0, (float *)workspace);
    return launch_fwd<3, 7>(stream, input, weight, output, B, H, W, 0, (float *)workspace);
}

size_t dcd_conv_stem_wrw_workspace_bytes(int Cin, int Cout, int ksize)
{
    if (Cout != 16) return 0;
    if (Cin == 16 && ksize == 3) return (size_t)ST_WRW_WGS * 4 * StemCfg<16, 3>::NB * 256 * sizeof(float);
    if (Cin == 3 && ksize == 7) return (size_t)ST_WRW_WGS * 4 * StemCfg<3, 7>::NB * 256 * sizeof(float);
    return 0;
}

int dcd_conv_stem_wrw(void *stream_, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                      int Cout, int ksize, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!input || !grad_output || !grad_weight || !workspace || !stem_shape_ok(B, Cin, H, W, Cout, ksize)) return DCD_ERR_BAD_ARG;
    if (workspace_bytes < dcd_conv_stem_wrw_workspace_bytes(Cin, Cout, ksize)) return DCD_ERR_WORKSPACE;
    if (Cin == 16) return launch_wrw<16, 3>(stream, input, grad_output, grad_weight, B, H, W, (float *)workspace);
    return launch_wrw<3, 7>(stream, input, grad_output, grad_weight, B, H, W, (float *)workspace);
}

}  // extern "C"
