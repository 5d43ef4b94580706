// Microbenchmark: issue rate of vector-memory load instructions on gfx950 as a function of the lane address pattern.
// Question it answers: what does one wave-level gather cost in the texture-addresser / L1 pipeline when the data is
// cache resident?  (The DCN kernels issue 18 dwordx2 gathers per channel pair per 32-pixel tile.)
//   clk/instr/CU = elapsed * f_clk * n_CU / total wave instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int PLANE = 96 * 320;     // floats per channel plane
constexpr int NPL = 64;             // planes

// mode: 0 coalesced 256 B; 1 two aligned 128-B segments in two planes; 2 two misaligned segments;
//       3 misaligned + per-lane jitter {0,1}; 4 = 3 with dwordx2; 5 lane stride 256 B dwordx4 (NHWC pixel-per-lane);
//       6 half-wave broadcast (32 lanes same dword); 7 = 1 with dwordx4 per lane... (contiguous 512 B per half)
//       8 lane=channel NHWC: 32 lanes x 4 B contiguous at a jittered pixel per half-wave
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ buf, float *__restrict__ out, int iters, int wrap)
{
    const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned jit = (lane * 2654435761u >> 13) & 1u;
    // each wave walks a small private window (stays in L1/L2) -> measures instruction issue, not DRAM
    const unsigned base = (wid * 64u) % (unsigned)(PLANE - 8192);
    unsigned idx;
    if (MODE == 0) idx = base + lane;
    else if (MODE == 1) idx = base + p + h * PLANE;
    else if (MODE == 2) idx = base + p + 5 + h * PLANE;
    else if (MODE == 3 || MODE == 4) idx = base + p + 5 + jit + h * PLANE;
    else if (MODE == 5) idx = (base & ~3u) + lane * 64u;
    else if (MODE == 6) idx = base + h * PLANE;
    else if (MODE == 7) idx = ((base + h * PLANE) & ~3u) + p * 4u;
    else idx = (base & ~31u) + p + h * 4096u;
    if (MODE == 4) idx &= ~1u;
    const float *q = buf + idx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    const unsigned wmask = (unsigned)wrap - 1u;
#define LD(u, acc)                                                                                   \
    if (MODE == 4) { const f32x2 v = *reinterpret_cast<const f32x2 *>(r + (u) * 32); acc += v.x + v.y; }   \
    else if (MODE == 5 || MODE == 7) { const f32x4 v = *reinterpret_cast<const f32x4 *>(r + (u) * 32); acc += v.x + v.w; } \
    else acc += r[(u) * 32];
    for (int i = 0; i < iters; ++i) {
        const float *r = q + ((unsigned)i & wmask) * 256u;
        LD(0, a0) LD(1, a1) LD(2, a2) LD(3, a3) LD(4, a4) LD(5, a5) LD(6, a6) LD(7, a7)
    }
    const float acc = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE>
int run(const float *buf, float *out, const char *name, double bytes_per_lane)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 512;
    for (int wrap : {1, 16}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, out, iters, wrap);
            hipEventRecord(e1);
            CK(hipEventSynchronize(e1));
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double instrs = (double)blocks * 4 * iters * 8;
        printf("%-52s wrap %2d  %7.3f ms  %6.2f clk/instr/CU  %7.1f B/clk/CU\n", name, wrap, best,
               best * 1e-3 * 2.4e9 * 256 / instrs, instrs * 64 * bytes_per_lane / (best * 1e-3 * 2.4e9 * 256));
    }
    return 0;
}

int main()
{
    float *buf, *out;
    CK(hipMalloc(&buf, sizeof(float) * (size_t)PLANE * NPL + 65536));
    CK(hipMalloc(&out, 4096));
    CK(hipMemset(buf, 0, sizeof(float) * (size_t)PLANE * NPL + 65536));
    run<0>(buf, out, "0 coalesced dword (256 B contiguous)", 4);
    run<1>(buf, out, "1 two aligned 128-B segments (2 planes)", 4);
    run<2>(buf, out, "2 two misaligned segments", 4);
    run<3>(buf, out, "3 misaligned + lane jitter (dword)", 4);
    run<4>(buf, out, "4 misaligned + lane jitter (dwordx2)", 8);
    run<5>(buf, out, "5 lane stride 256 B, dwordx4", 16);
    run<6>(buf, out, "6 half-wave broadcast dword", 4);
    run<7>(buf, out, "7 contiguous dwordx4 (512 B per half-wave)", 16);
    run<8>(buf, out, "8 aligned 128 B per half-wave, far apart", 4);
    return 0;
}
