// Microbenchmark: fp32 atomic-add throughput on gfx950 (global coalesced / strided / random, LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void g_atomic(float *buf, unsigned n_mask, int mode, int iters)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned idx = tid;
    for (int i = 0; i < iters; ++i) {
        unsigned a;
        if (mode == 0) a = (idx + i * 64) & n_mask;                                  // coalesced: 64 consecutive floats
        else if (mode == 1) a = ((idx * 16) + i * 1024) & n_mask;                    // one lane per 64B line
        else if (mode == 2) { unsigned h = (idx + i * 7919u) * 2654435761u; a = (h >> 7) & n_mask; }  // random
        else { unsigned row = (tid & 31) + ((tid * 2654435761u >> 28) & 3) * 320; a = ((tid >> 5) * 64 + row + i * 4096) & n_mask; } // jittered rows
        atomicAdd(buf + a, 1.0f);
    }
}
__global__ void g_store(float *buf, unsigned n_mask, int iters)
{
    unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = 0; i < iters; ++i) buf[(tid + i * 64 * 1024) & n_mask] = 1.0f;
}
__global__ void l_atomic(float *out, int mode, int iters)
{
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0;
    __syncthreads();
    unsigned t = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        unsigned a;
        if (mode == 0) a = (t + i * 64) & 8191;                       // conflict free
        else if (mode == 1) a = ((t * 2654435761u + i * 40503u) >> 9) & 8191;  // random
        else a = ((t >> 1) + i * 64) & 8191;                          // 2 lanes same address
        atomicAdd(&s[a], 1.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5];
}
int main()
{
    const unsigned N = 1u << 24;  // 64 MB
    float *buf, *out;
    CK(hipMalloc(&buf, N * 4)); CK(hipMalloc(&out, 4096 * 4));
    CK(hipMemset(buf, 0, N * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, threads = 256, iters = 64;
    const char *names[] = {"global coalesced", "global 1 lane/line", "global random", "global jittered rows"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(g_atomic, dim3(blocks), dim3(threads), 0, 0, buf, N - 1, mode, iters);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-22s %8.3f ms  %8.1f G lane-atomics/s\n", names[mode], ms, (double)blocks * threads * iters / ms / 1e6);
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(g_store, dim3(blocks), dim3(threads), 0, 0, buf, N - 1, iters);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-22s %8.3f ms  %8.1f G lane-stores/s\n", "global store coalesced", ms, (double)blocks * threads * iters / ms / 1e6);
    }
    const char *ln[] = {"LDS conflict-free", "LDS random", "LDS 2-way same addr"};
    for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(l_atomic, dim3(2048), dim3(256), 0, 0, out, mode, 1024);
        hipEventRecord(e1); CK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%-22s %8.3f ms  %8.1f G lane-atomics/s (chip)\n", ln[mode], ms, 2048.0 * 256 * 1024 / ms / 1e6);
    }
    return 0;
}
