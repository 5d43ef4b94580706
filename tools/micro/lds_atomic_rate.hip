// Microbenchmark (round 3): throughput of LDS fp32 atomics (`ds_add_f32`, no return) on gfx950 in the two address
// patterns a one-pass DCNv2 backward could use for its grad_input window, next to plain LDS stores/loads of the same pattern.
//   mode 0  lane = channel : addr = (lane & 31) * PLANE + cell(half, i)        (PLANE odd -> 32 distinct banks per lane group)
//   mode 1  lane = pixel   : addr = plane(i) * PLANE + row * 40 + (lane & 31) + jitter(lane)   (adjacent lanes -> adjacent cells)
//   mode 2  all lanes one address (worst case)
// op 0 ds_add_f32, 1 ds_write_b32, 2 ds_read_b32, 3 ds_add_rtn_f32
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int PLANE = 489;
constexpr int NPL = 32;

template <int OP, int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < NPL * PLANE; i += blockDim.x) s[i] = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned h = lane >> 5, p = lane & 31;
    const unsigned jit = (lane * 2654435761u >> 30) & 1;      // 0/1 px jitter
    float acc = 0.f;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
        float rr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            rr[u] = 0.f;
            unsigned a;
            if (MODE == 0) a = p * PLANE + h * 5 + wave * 11 + ((it * 16 + u) * 7) % 400;
            else if (MODE == 1) a = ((u + 4 * h + wave) & 31) * PLANE + ((it + u) % 10) * 40 + p + jit + (u & 3);
            else a = 17;
            const unsigned byte = a * 4;
            if (OP == 0) asm volatile("ds_add_f32 %0, %1" ::"v"(byte), "v"(v) : "memory");
            else if (OP == 1) asm volatile("ds_write_b32 %0, %1" ::"v"(byte), "v"(v) : "memory");
            else if (OP == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(rr[u]) : "v"(byte) : "memory");
            else asm volatile("ds_add_rtn_f32 %0, %1, %2" : "=v"(rr[u]) : "v"(byte), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (OP >= 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += rr[u];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5] + acc;
}

template <int OP, int MODE> int run(const char *name, float *out, int wg_per_cu)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wg_per_cu * 4, iters = 512;
    const size_t sh = NPL * PLANE * 4;       // 62.6 KB -> at most 2 workgroups per CU
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP, MODE>), dim3(blocks), dim3(256), sh, 0, out, iters);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double waveinst = (double)blocks * 4 * iters * 16;
    // cycles per wave instruction per CU at 2.4 GHz
    printf("%-34s %8.3f ms  %8.1f G lane-ops/s  %6.2f clk/wave-inst/CU\n", name, ms, waveinst * 64 / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / waveinst);
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 65536 * 4));
    CK(hipFuncSetAttribute((const void *)k<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    run<0, 0>("ds_add_f32   lane=channel", out, 2);
    run<0, 1>("ds_add_f32   lane=pixel", out, 2);
    run<0, 2>("ds_add_f32   one address", out, 2);
    run<1, 0>("ds_write_b32 lane=channel", out, 2);
    run<1, 1>("ds_write_b32 lane=pixel", out, 2);
    run<2, 0>("ds_read_b32  lane=channel", out, 2);
    run<2, 1>("ds_read_b32  lane=pixel", out, 2);
    run<3, 0>("ds_add_rtn_f32 lane=channel", out, 2);
    run<3, 1>("ds_add_rtn_f32 lane=pixel", out, 2);
    return 0;
}
