// Does the shader clock depend on how many CUs are busy?  A fixed dependent chain (FMA + barrier per step) on G workgroups;
// reports shader cycles (clock64) and elapsed time (wall_clock64, 100 MHz) -> effective MHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/clock_probe tools/micro/clock_probe.hip && tools/micro/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(float *out, long long *cyc, long long *wall, int steps)
{
    __shared__ float sh[256];
    float v = threadIdx.x * 1e-3f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < steps; ++i) {
        sh[threadIdx.x] = v;
        __syncthreads();
        v = v * 1.0001f + sh[(threadIdx.x + 1) & 255];
        __syncthreads();
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = v;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}

int main()
{
    float *out; long long *cyc, *wall;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8); hipMalloc(&wall, 4096 * 8);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    for (int G : {8, 8, 64, 256, 2048, 8}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(G), dim3(256), 0, 0, out, cyc, wall, 300);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(chain, dim3(G), dim3(256), 0, 0, out, cyc, wall, 300);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c, w;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
        printf("G %5d: kernel %.1f us (events); block 0: %lld shader cycles, %lld wall ticks (%d kHz) -> %.0f MHz, %.0f cycles per step\n", G,
               ms * 1e3, c, w, wall_khz, (double)c / ((double)w / wall_khz * 1e-3) * 1e-6, (double)c / 300);
    }
    return 0;
}
