// Zero-filling a small accumulation target with a KERNEL, not hipMemsetAsync.
// Inside a captured HIP graph a memset node followed by a kernel that accumulates into the same memory with atomics is not
// reliable on this stack (ROCm 7.0 runtime bundled with PyTorch 2.10, gfx950): replays intermittently saw the stale contents
// of the previous replay -- torch's own multi-block reductions (hipMemsetAsync on their semaphores) returned another
// reduction's result or nothing from the second replay on (profiles/r02_graph_memset_hazard.txt).  Kernel -> kernel ordering
// inside a graph is fine, so every accumulator of ours is cleared by a launch.
#pragma once
#include <hip/hip_runtime.h>

namespace {
__global__ void dcd_zero_fill_kernel(float *__restrict__ p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

inline bool dcd_zero_fill(hipStream_t stream, float *p, size_t n)
{
    if (n == 0) return true;
    const unsigned grid = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(dcd_zero_fill_kernel, dim3(grid), dim3(256), 0, stream, p, n);
    return hipGetLastError() == hipSuccess;
}
}  // namespace
