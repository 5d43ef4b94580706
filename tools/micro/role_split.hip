// Microbenchmark (round 4): does a VALU / LDS-only wave run beside an MFMA-only wave of the SAME SIMD at their own rates?
// (round 3 measured one wave that mixes both: every VALU instruction adds 5.5 clk to the f32-input MFMA stream, 3.2 clk with two
// such waves per SIMD.)  A workgroup of 4 * R waves: wave w sits on SIMD (w & 3)-ish (dispatch order 0,2,1,3 repeating), so waves
// w, w + 4, w + 8 ... share a SIMD.  Role of wave w = role[w >> 2]:  M = back-to-back v_mfma_f32_16x16x4_f32 on 8 accumulators,
// V = v_fma_f32 on 8 independent chains, L = ds_read_b64 / ds_write_b32 pairs + 2 VALU (the scatter wave's instruction mix),
// X = 1 MFMA + 8 VALU interleaved (round 3's mixed wave), '-' = exits at once.  Every wave stamps s_memtime around its loop.
// build: hipcc --offload-arch=gfx950 -O3 -o bin/role_split role_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k(const char *roles, int iters, unsigned long long *stamps, float *sink)
{
    __shared__ float lds[16384];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char role = roles[wave >> 2];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    if (role == '-') return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 1e-3f + i;
    const float x = lane * 1e-3f, y = 1.0001f;
    float *mine = lds + wave * 1024 + lane * 2;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 'M') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[u & 7], 0, 0, 0);
        }
    } else if (role == 'V') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(y), "v"(x));
        }
    } else if (role == 'L') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                f32x2 r = *reinterpret_cast<volatile f32x2 *>(mine + (u & 3) * 128);
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[u & 7]) : "v"(r.x), "v"(y));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[(u + 1) & 7]) : "v"(r.y), "v"(y));
                *reinterpret_cast<volatile float *>(mine + (u & 3) * 128 + 1) = v[u & 7];
            }
        }
    } else if (role == 'P') {                       // packed f32: two FMAs per lane and instruction
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 pv[8];
        for (int i = 0; i < 8; ++i) pv[i] = f2{v[i], v[i] + 1.f};
        const f2 px = {x, x}, py = {y, y};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[u & 7]) : "v"(py), "v"(px));
        }
        for (int i = 0; i < 8; ++i) v[i] = pv[i].x + pv[i].y;
    } else if (role == 'R') {                       // LDS only: ds_read_b64 x 2 + ds_write_b32 x 2 per group, no VALU
        typedef float f2 __attribute__((ext_vector_type(2)));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                f2 r0 = *reinterpret_cast<volatile f2 *>(mine + (u & 3) * 128);
                f2 r1 = *reinterpret_cast<volatile f2 *>(mine + (u & 3) * 128 + 256);
                *reinterpret_cast<volatile float *>(mine + (u & 3) * 128 + 1) = r0.x;
                *reinterpret_cast<volatile float *>(mine + (u & 3) * 128 + 257) = r1.x;
            }
        }
    } else if (role == 'Y') {                       // round-3-like mixed wave: 1 MFMA + 4 VALU + 2 LDS
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[u & 7], 0, 0, 0);
                f32x2 r = *reinterpret_cast<volatile f32x2 *>(mine + (u & 3) * 128);
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(y), "v"(x));
                *reinterpret_cast<volatile float *>(mine + (u & 3) * 128 + 1) = r.x + r.y;
            }
        }
    } else if (role == 'X') {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[u & 7], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < 8; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[n & 7]) : "v"(y), "v"(x));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + acc[i].x + acc[i].w;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { stamps[(blockIdx.x * 16 + wave) * 2] = t0; stamps[(blockIdx.x * 16 + wave) * 2 + 1] = t1; }
}

int main()
{
    const char *configs[] = {"M", "V", "P", "L", "R", "X", "Y", "MV", "MP", "ML", "MR", "MVV", "MPP", "MLL", "MRR", "MVVV", "MLLL", "MM", "VV", "PP", "LL", "RR", "VVVV", "LLLL", "XX", "YY", "MMV", "MML", "MMLL"};
    float *sink; hipMalloc(&sink, 1 << 22);
    unsigned long long *stamps; hipMalloc(&stamps, 256 * 16 * 2 * 8);
    char *droles; hipMalloc(&droles, 8);
    const int iters = 2048;
    printf("per role group: instructions per iteration M=16 mfma, V=64 fma, P=64 pk_fma, L=16 x (ds_read_b64 + 2 fma + ds_write_b32), R=16 x (2 ds_read_b64 + 2 ds_write_b32), X=8 x (mfma + 8 fma), Y=8 x (mfma + ds_read_b64 + 4 fma + add + ds_write_b32)\n");
    for (const char *cfg : configs) {
        const int R = (int)strlen(cfg);
        char buf[8] = {0};
        memcpy(buf, cfg, R);
        hipMemcpy(droles, buf, 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(stamps, 0, 256 * 16 * 2 * 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(256 * R), 0, 0, droles, iters, stamps, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<unsigned long long> h(256 * 16 * 2);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        printf("%-6s kernel %7.3f ms |", cfg, ms);
        for (int r = 0; r < R; ++r) {
            // median over blocks and the group's four waves of the loop duration, in s_memtime ticks
            std::vector<double> d;
            for (int b = 0; b < 256; ++b)
                for (int w = 4 * r; w < 4 * r + 4; ++w) d.push_back((double)(h[(b * 16 + w) * 2 + 1] - h[(b * 16 + w) * 2]));
            std::sort(d.begin(), d.end());
            const double ticks = d[d.size() / 2];
            const char role = cfg[r];
            const double per_iter_inst = role == 'M' ? 16 : (role == 'V' || role == 'P') ? 64 : (role == 'L' || role == 'R') ? 16 : 8;
            printf("  %c: %8.0f ticks  %7.2f ticks/1k-inst-group", role, ticks, ticks / (iters * per_iter_inst) * 1000.0);
        }
        printf("\n");
    }
    return 0;
}
