// Microbenchmark (round 3): how many independent VALU instructions hide under one f32-input MFMA for ONE wave per SIMD (and for
// two)?  Loop body = 1 MFMA (16x16x4 f32 or 32x32x2 f32, rotating over 4 accumulators) + N v_fma_f32 on 8 independent chains.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N, int BIG>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x4 a4[4];
    f32x16 a16[2];
    for (int i = 0; i < 4; ++i) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) a16[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (BIG) a16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a16[u & 1], 0, 0, 0);
            else a4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4[u & 3], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(y), "v"(x));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += a4[i].x + a4[i].w;
    for (int i = 0; i < 2; ++i) s += a16[i][0] + a16[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N, int BIG> void run(float *out, int waves_per_simd)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * waves_per_simd;      // 256 threads = one wave per SIMD per block
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<N, BIG>), dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    // cycles per MFMA per SIMD at 2.4 GHz nominal (the real clock under load is lower: compare rows, not absolutes)
    printf("%s N=%2d waves/SIMD=%d : %7.3f ms  %6.1f clk(2.4GHz)/MFMA/SIMD\n", BIG ? "32x32x2" : "16x16x4", N, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 8 * waves_per_simd));
}

int main()
{
    float *out; hipMalloc(&out, 1 << 22);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(out, w); run<2, 0>(out, w); run<4, 0>(out, w); run<6, 0>(out, w); run<8, 0>(out, w); run<12, 0>(out, w); run<16, 0>(out, w);
    }
    for (int w = 1; w <= 2; ++w) {
        run<0, 1>(out, w); run<4, 1>(out, w); run<8, 1>(out, w); run<12, 1>(out, w); run<16, 1>(out, w); run<24, 1>(out, w); run<32, 1>(out, w);
    }
    return 0;
}
