// upsample.hip -- depthwise transposed convolution of IDAUp (DGDE/model/backbone/dla_dcn.py:412-438):
//   nn.ConvTranspose2d(o, o, 2f, stride=f, padding=f/2, output_padding=0, groups=o, bias=False), f in {2, 4, 8},
// the learnable bilinear up-sampling between the DCN projection and node convolutions (weights initialised by
// fill_up_weights, :386-395).  MIOpen has no solver for it and falls back to naive / im2col kernels: the eight layers cost
// 5.1 ms per bs-8 step (tools/time_upsample.py) for 4 multiply-adds per output element.  Here:
//   forward        y[b,c,Y,X] = sum over the 2x2 inputs (iy, ix) whose kernel window covers (Y, X) of x * w[c, Y+p-iy f, X+p-ix f]
//   backward       one pass over grad_y per input element: its (2f)^2 patch gives grad_x (dot with w) and the element's
//                  contribution to grad_w (x * patch), reduced per block and added with (2f)^2 atomics per block.
// All HBM-bound (forward: one write of y; backward: one read of grad_y); NCHW fp32, one (b, c) plane chunk per workgroup.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <cstdint>
#include <stdint.h>

#include "../../include/dcd_hip.h"
#include "tuning_env.h"
#include "zero_fill.h"

namespace {

// grid = (ceil(Ho*Wo / 1024), B*C); 256 threads x 4 outputs (consecutive X)
template <int F>
__global__ __launch_bounds__(256) void up_dw_fwd(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                 int C, int H, int W, const float *__restrict__ skip)
{
    constexpr int K = 2 * F, P = F / 2;
    __shared__ float ws[K * K];
    const int plane = blockIdx.y, c = plane % C;
    for (int i = threadIdx.x; i < K * K; i += 256) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const int Ho = H * F, Wo = W * F;
    const float *xp = x + (size_t)plane * H * W;
    float *yp = y + (size_t)plane * Ho * Wo;
    const int o0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (o0 >= Ho * Wo) return;
    const int Y = o0 / Wo, X0 = o0 - Y * Wo;                  // Wo % 4 == 0: the four outputs share the row
    const int iy1 = (Y + P) / F, iy0 = iy1 - 1;
    const int ky1 = Y + P - iy1 * F, ky0 = ky1 + F;           // kernel rows used with input rows iy1 / iy0
    float out[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int X = X0 + u;
        const int ix1 = (X + P) / F, ix0 = ix1 - 1;
        const int kx1 = X + P - ix1 * F, kx0 = kx1 + F;
        float acc = 0.f;
        if (iy1 < H) {
            if (ix1 < W) acc += xp[iy1 * W + ix1] * ws[ky1 * K + kx1];
            if (ix0 >= 0) acc += xp[iy1 * W + ix0] * ws[ky1 * K + kx0];
        }
        if (iy0 >= 0) {
            if (ix1 < W) acc += xp[iy0 * W + ix1] * ws[ky0 * K + kx1];
            if (ix0 >= 0) acc += xp[iy0 * W + ix0] * ws[ky0 * K + kx0];
        }
        out[u] = acc;
    }
    if (skip) {                                               // IDAUp: node(up(x) + skip) -- the sum leaves this kernel
        const float4 sk = *reinterpret_cast<const float4 *>(skip + (size_t)plane * Ho * Wo + o0);
        out[0] += sk.x; out[1] += sk.y; out[2] += sk.z; out[3] += sk.w;
    }
    *reinterpret_cast<float4 *>(yp + o0) = make_float4(out[0], out[1], out[2], out[3]);
}

// F = 2 (seven of IDAUp's eight layers): a thread writes a 2 x 4 block of outputs (rows 2a, 2a+1; columns 4m .. 4m+3), which read
// the same 3 x 4 inputs (rows a-1 .. a+1, columns 2m-1 .. 2m+2): nine loads per eight outputs where up_dw_fwd issues sixteen per four.
// Kernel taps (ky = Y + 1 - 2 iy, kx = X + 1 - 2 ix): row 2a <- (a, 1), (a-1, 3); row 2a+1 <- (a+1, 0), (a, 2); column 4m+u <-
// u = 0: (2m, 1), (2m-1, 3);  1: (2m+1, 0), (2m, 2);  2: (2m+1, 1), (2m, 3);  3: (2m+2, 0), (2m+1, 2).   W even (Wo % 4 == 0).
__global__ __launch_bounds__(256) void up_dw_fwd2_block(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ y,
                                                        int C, int H, int W, const float *__restrict__ skip)
{
    const int plane = blockIdx.y, c = plane % C;
    float k[4][4];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i >> 2][i & 3] = w[(size_t)c * 16 + i];      // uniform address: scalar loads
    const int W2 = W >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W2) return;
    const int a = t / W2, m = t - a * W2;
    const float *xp = x + (size_t)plane * H * W;
    float v[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = a - 1 + r;
        const bool rv = iy >= 0 && iy < H;
        const float *row = xp + (size_t)(rv ? iy : 0) * W;
        const float2 mid = rv ? *reinterpret_cast<const float2 *>(row + 2 * m) : make_float2(0.f, 0.f);      // columns 2m, 2m+1
        v[r][0] = (rv && m > 0) ? row[2 * m - 1] : 0.f;
        v[r][1] = mid.x;
        v[r][2] = mid.y;
        v[r][3] = (rv && 2 * m + 2 < W) ? row[2 * m + 2] : 0.f;
    }
    const int Wo = 2 * W;
    const size_t o0 = (size_t)plane * (4 * (size_t)H * W) + (size_t)(2 * a) * Wo + 4 * m;
    float out[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        // (input row index into v, kernel row) pairs of this output row
        const int r1 = rr == 0 ? 1 : 2, ky1 = rr == 0 ? 1 : 0, r0 = rr == 0 ? 0 : 1, ky0 = rr == 0 ? 3 : 2;
        out[rr][0] = v[r1][1] * k[ky1][1] + v[r1][0] * k[ky1][3] + v[r0][1] * k[ky0][1] + v[r0][0] * k[ky0][3];
        out[rr][1] = v[r1][2] * k[ky1][0] + v[r1][1] * k[ky1][2] + v[r0][2] * k[ky0][0] + v[r0][1] * k[ky0][2];
        out[rr][2] = v[r1][2] * k[ky1][1] + v[r1][1] * k[ky1][3] + v[r0][2] * k[ky0][1] + v[r0][1] * k[ky0][3];
        out[rr][3] = v[r1][3] * k[ky1][0] + v[r1][2] * k[ky1][2] + v[r0][3] * k[ky0][0] + v[r0][2] * k[ky0][2];
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        float4 o = make_float4(out[rr][0], out[rr][1], out[rr][2], out[rr][3]);
        if (skip) {                                           // IDAUp: node(up(x) + skip) -- the sum leaves this kernel
            const float4 sk = *reinterpret_cast<const float4 *>(skip + o0 + (size_t)rr * Wo);
            o.x += sk.x; o.y += sk.y; o.z += sk.z; o.w += sk.w;
        }
        *reinterpret_cast<float4 *>(y + o0 + (size_t)rr * Wo) = o;
    }
}

// grid = (ceil(H*W / 256), B*C); one input element per thread
template <int F>
__global__ __launch_bounds__(256) void up_dw_bwd(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ gy,
                                                 float *__restrict__ gx, float *__restrict__ gw, int C, int H, int W)
{
    constexpr int K = 2 * F, P = F / 2;
    __shared__ float ws[K * K];
    __shared__ float red[4][K * K];
    const int plane = blockIdx.y, c = plane % C;
    for (int i = threadIdx.x; i < K * K; i += 256) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const int Ho = H * F, Wo = W * F;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool qv = q < H * W;
    const int iy = qv ? q / W : 0, ix = qv ? q - iy * W : 0;
    const float xv = qv ? x[(size_t)plane * H * W + q] : 0.f;
    const float *gp = gy + (size_t)plane * Ho * Wo;
    const int Y0 = iy * F - P, X0 = ix * F - P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dx = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int Y = Y0 + ky;
        const bool yv = qv && Y >= 0 && Y < Ho;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int X = X0 + kx;
            const float g = (yv && X >= 0 && X < Wo) ? gp[(size_t)Y * Wo + X] : 0.f;
            dx += g * ws[ky * K + kx];
            float s = g * xv;                                   // this element's share of grad_w[c][ky][kx]
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red[wave][ky * K + kx] = s;
        }
    }
    if (qv) gx[(size_t)plane * H * W + q] = dx;
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += 256)
        atomicAdd(gw + (size_t)c * K * K + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
}

// F = 2, 4: grid = (row chunks, B*C); a workgroup walks `rows` input rows of one plane, one input element per thread and trip, and
// keeps its share of grad_w[c] (K*K = 16 / 64 values) in registers: ONE wave reduction and K*K atomics per workgroup at the end,
// where up_dw_bwd pays a six-step shuffle reduction per tap and element (96 shuffles per element at F = 2) and K*K atomics per
// 256 elements.  Same sums; the order of the additions into grad_w differs (atomics: it was not fixed before either).
template <int F>
__global__ __launch_bounds__(256) void up_dw_bwd_rows(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ gy,
                                                      float *__restrict__ gx, float *__restrict__ gw, int C, int H, int W, int rows)
{
    constexpr int K = 2 * F, P = F / 2;
    __shared__ float ws[K * K];
    __shared__ float red[4][K * K];
    const int plane = blockIdx.y, c = plane % C;
    for (int i = threadIdx.x; i < K * K; i += 256) ws[i] = w[(size_t)c * K * K + i];
    __syncthreads();
    const int Ho = H * F, Wo = W * F;
    const int r0 = blockIdx.x * rows, r1 = r0 + rows < H ? r0 + rows : H;
    const float *gp = gy + (size_t)plane * Ho * Wo;
    const float *xp = x + (size_t)plane * H * W;
    float *gxp = gx + (size_t)plane * H * W;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int q = r0 * W + threadIdx.x; q < r1 * W; q += 256) {
        const int iy = q / W, ix = q - iy * W;
        const float xv = xp[q];
        const int Y0 = iy * F - P, X0 = ix * F - P;
        float dx = 0.f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int Y = Y0 + ky;
            const bool yv = Y >= 0 && Y < Ho;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int X = X0 + kx;
                const float g = (yv && X >= 0 && X < Wo) ? gp[(size_t)Y * Wo + X] : 0.f;
                dx += g * ws[ky * K + kx];
                acc[ky * K + kx] += g * xv;
            }
        }
        gxp[q] = dx;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        float s = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K; i += 256)
        atomicAdd(gw + (size_t)c * K * K + i, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
}

// 2x2 / stride-2 max pooling (the `downsample` of every DLA Tree, dla_dcn.py:228).  thread = two outputs = a 2 x 4 input patch
// (two float4 rows).  The backward re-derives the arg-max from x instead of storing indices: first maximum in scan order wins,
// a NaN takes over -- the rule of the stock kernel -- and the gradient leaves as two float4 rows (stock: 142 us for the
// 126 MB map of level 2, index tensor included; this: read x + gy, write gx).  W % 4 == 0, H % 2 == 0.
__device__ __forceinline__ int argmax4(float a, float b, float c, float d, float &m)
{
    int k = 0;
    m = a;
    if (b > m || b != b) { m = b; k = 1; }
    if (c > m || c != c) { m = c; k = 2; }
    if (d > m || d != d) { m = d; k = 3; }
    return k;
}

__global__ __launch_bounds__(256) void maxpool2_fwd(const float *__restrict__ x, float *__restrict__ y, int64_t planes, int H, int W)
{
    const int Ho = H / 2, Wq = W / 4;
    const int64_t n = planes * Ho * Wq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % Wq);
        const int64_t r = i / Wq;
        const int oy = (int)(r % Ho);
        const int64_t pl = r / Ho;
        const float *p = x + (pl * H + 2 * oy) * W + 4 * q;
        const float4 t = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + W);
        float m0, m1;
        argmax4(t.x, t.y, b.x, b.y, m0);
        argmax4(t.z, t.w, b.z, b.w, m1);
        *reinterpret_cast<float2 *>(y + (pl * Ho + oy) * (W / 2) + 2 * q) = make_float2(m0, m1);
    }
}

__global__ __launch_bounds__(256) void maxpool2_bwd(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gx,
                                                    int64_t planes, int H, int W)
{
    const int Ho = H / 2, Wq = W / 4;
    const int64_t n = planes * Ho * Wq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % Wq);
        const int64_t r = i / Wq;
        const int oy = (int)(r % Ho);
        const int64_t pl = r / Ho;
        const size_t o = (size_t)(pl * H + 2 * oy) * W + 4 * q;
        const float4 t = *reinterpret_cast<const float4 *>(x + o), b = *reinterpret_cast<const float4 *>(x + o + W);
        const float2 g = *reinterpret_cast<const float2 *>(gy + (pl * Ho + oy) * (W / 2) + 2 * q);
        float m;
        const int k0 = argmax4(t.x, t.y, b.x, b.y, m), k1 = argmax4(t.z, t.w, b.z, b.w, m);
        const float4 gt = make_float4(k0 == 0 ? g.x : 0.f, k0 == 1 ? g.x : 0.f, k1 == 0 ? g.y : 0.f, k1 == 1 ? g.y : 0.f);
        const float4 gb = make_float4(k0 == 2 ? g.x : 0.f, k0 == 3 ? g.x : 0.f, k1 == 2 ? g.y : 0.f, k1 == 3 ? g.y : 0.f);
        *reinterpret_cast<float4 *>(gx + o) = gt;
        *reinterpret_cast<float4 *>(gx + o + W) = gb;
    }
}

}  // namespace

extern "C" {

int dcd_maxpool2x2_forward(void *stream_, const float *x, float *y, int64_t planes, int H, int W)
{
    (void)hipGetLastError();
    if (!x || !y || planes <= 0 || H < 2 || W < 4 || (H & 1) || (W & 3) || (((uintptr_t)x | (uintptr_t)y) & 15)) return DCD_ERR_BAD_ARG;
    const int64_t n = planes * (H / 2) * (W / 4);
    const unsigned grid = (unsigned)((n + 255) / 256 < 65535 * 4 ? (n + 255) / 256 : 65535 * 4);
    hipLaunchKernelGGL(maxpool2_fwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, x, y, planes, H, W);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_maxpool2x2_backward(void *stream_, const float *x, const float *grad_y, float *grad_x, int64_t planes, int H, int W)
{
    (void)hipGetLastError();
    if (!x || !grad_y || !grad_x || planes <= 0 || H < 2 || W < 4 || (H & 1) || (W & 3) ||
        (((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x) & 15))
        return DCD_ERR_BAD_ARG;
    const int64_t n = planes * (H / 2) * (W / 4);
    const unsigned grid = (unsigned)((n + 255) / 256 < 65535 * 4 ? (n + 255) / 256 : 65535 * 4);
    hipLaunchKernelGGL(maxpool2_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream_, x, grad_y, grad_x, planes, H, W);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

static int upsample_forward(hipStream_t stream, const float *x, const float *weight, const float *skip, float *y, int B, int C, int H,
                            int W, int f)
{
    (void)hipGetLastError();
    if (!x || !weight || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535) return DCD_ERR_BAD_ARG;
    if ((f != 2 && f != 4 && f != 8) || ((W * f) & 3)) return DCD_ERR_BAD_ARG;
    const int no = H * f * W * f;
    dim3 grid((no / 4 + 255) / 256, B * C), block(256);
    static const bool old_fwd = dcd_env("DCD_UP_FWD_OLD") != nullptr;                          // A/B timing
    if (f == 2 && !old_fwd)
        hipLaunchKernelGGL(up_dw_fwd2_block, dim3((H * (W / 2) + 255) / 256, B * C), block, 0, stream, x, weight, y, C, H, W, skip);
    else if (f == 2) hipLaunchKernelGGL(up_dw_fwd<2>, grid, block, 0, stream, x, weight, y, C, H, W, skip);
    else if (f == 4) hipLaunchKernelGGL(up_dw_fwd<4>, grid, block, 0, stream, x, weight, y, C, H, W, skip);
    else hipLaunchKernelGGL(up_dw_fwd<8>, grid, block, 0, stream, x, weight, y, C, H, W, skip);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_upsample_dw_forward(void *stream_, const float *x, const float *weight, float *y, int B, int C, int H, int W, int f)
{
    return upsample_forward((hipStream_t)stream_, x, weight, nullptr, y, B, C, H, W, f);
}

int dcd_upsample_dw_forward_add(void *stream_, const float *x, const float *weight, const float *skip, float *y, int B, int C, int H,
                                int W, int f)
{
    if (!skip) return DCD_ERR_BAD_ARG;
    return upsample_forward((hipStream_t)stream_, x, weight, skip, y, B, C, H, W, f);
}

int dcd_upsample_dw_backward(void *stream_, const float *x, const float *weight, const float *grad_y, float *grad_x,
                             float *grad_weight, int B, int C, int H, int W, int f)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!x || !weight || !grad_y || !grad_x || !grad_weight || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (int64_t)B * C > 65535)
        return DCD_ERR_BAD_ARG;
    if (f != 2 && f != 4 && f != 8) return DCD_ERR_BAD_ARG;
    if (!dcd_zero_fill(stream, grad_weight, (size_t)C * 4 * f * f)) return DCD_ERR_LAUNCH;   // a launch, not a memset node: zero_fill.h
    dim3 grid((H * W + 255) / 256, B * C), block(256);
    // f = 2, 4: row-chunk kernel, about 2048 workgroups (8 per CU) or at least four trips per thread
    int chunks = (2048 + B * C - 1) / (B * C);
    const int max_chunks = (H * W + 1023) / 1024;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int rows = (H + chunks - 1) / chunks;
    dim3 rgrid((H + rows - 1) / rows, B * C);
    static const bool old_kernel = dcd_env("DCD_UP_BWD_OLD") != nullptr;                       // A/B timing
    if (f == 2 && !old_kernel) hipLaunchKernelGGL(up_dw_bwd_rows<2>, rgrid, block, 0, stream, x, weight, grad_y, grad_x, grad_weight, C, H, W, rows);
    else if (f == 4 && !old_kernel) hipLaunchKernelGGL(up_dw_bwd_rows<4>, rgrid, block, 0, stream, x, weight, grad_y, grad_x, grad_weight, C, H, W, rows);
    else if (f == 2) hipLaunchKernelGGL(up_dw_bwd<2>, grid, block, 0, stream, x, weight, grad_y, grad_x, grad_weight, C, H, W);
    else if (f == 4) hipLaunchKernelGGL(up_dw_bwd<4>, grid, block, 0, stream, x, weight, grad_y, grad_x, grad_weight, C, H, W);
    else hipLaunchKernelGGL(up_dw_bwd<8>, grid, block, 0, stream, x, weight, grad_y, grad_x, grad_weight, C, H, W);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
