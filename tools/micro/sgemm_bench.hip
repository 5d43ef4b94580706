// Standalone check + timing of dcd_amd/csrc/sgemm_f32.inc on the three product shapes of the dense DCN path.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/sgemm_bench tools/micro/sgemm_bench.hip && tools/micro/sgemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "../../dcd_amd/csrc/sgemm_f32.inc"

static float frand() { return (float)rand() / RAND_MAX - 0.5f; }

static double check(bool ak, bool bk, int M, int N, int K, int Z, int nsplit)
{
    const int lda = ak ? K : M, ldb = bk ? K : N;
    std::vector<float> A((size_t)Z * M * K), B((size_t)Z * K * N), C((size_t)Z * nsplit * M * N), bias(M);
    for (auto &v : A) v = frand();
    for (auto &v : B) v = frand();
    for (auto &v : bias) v = frand();
    float *dA, *dB, *dC, *dbias;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dbias, M * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dbias, bias.data(), M * 4, hipMemcpyHostToDevice);
    SgemmArgs a{dA, dB, dC, nsplit == 1 ? dbias : nullptr, M, N, K, lda, ldb, N, (long long)M * K, (long long)K * N,
                (long long)nsplit * M * N, (long long)M * N, nsplit, ((K + nsplit - 1) / nsplit + 15) / 16 * 16, 0};
    sgemm_f32(0, ak, bk, a, Z);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int z = 0; z < Z; ++z)
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N; n += 5) {
                double ref = nsplit == 1 ? bias[m] : 0.0, got = 0;
                for (int k = 0; k < K; ++k) {
                    const float av = ak ? A[((size_t)z * M + m) * K + k] : A[((size_t)z * K + k) * M + m];
                    const float bv = bk ? B[((size_t)z * N + n) * K + k] : B[((size_t)z * K + k) * N + n];
                    ref += (double)av * bv;
                }
                for (int s = 0; s < nsplit; ++s) got += C[(((size_t)z * nsplit + s) * M + m) * N + n];
                worst = fmax(worst, fabs(got - ref));
            }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dbias);
    return worst;
}

static void timeit(const char *name, bool ak, bool bk, int M, int N, int K, int Z, int nsplit, long long sA, long long sB)
{
    const int lda = ak ? K : M, ldb = bk ? K : N;
    float *dA, *dB, *dC;
    const size_t nA = sA ? (size_t)Z * M * K : (size_t)M * K, nB = (size_t)Z * K * N, nC = (size_t)Z * nsplit * M * N;
    hipMalloc(&dA, nA * 4); hipMalloc(&dB, nB * 4); hipMalloc(&dC, nC * 4);
    hipMemset(dA, 0, nA * 4); hipMemset(dB, 0, nB * 4);
    SgemmArgs a{dA, dB, dC, nullptr, M, N, K, lda, ldb, N, sA, sB, (long long)nsplit * M * N, (long long)M * N, nsplit,
                ((K + nsplit - 1) / nsplit + 15) / 16 * 16, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) sgemm_f32(0, ak, bk, a, Z);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) sgemm_f32(0, ak, bk, a, Z);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-34s M %5d N %5d K %5d Z %d split %d: %.3f ms  %.1f TF/s\n", name, M, N, K, Z, nsplit, ms,
           2.0 * M * N * K * Z / ms / 1e9);
    hipFree(dA); hipFree(dB); hipFree(dC);
}


// 64-row tile with B rows at listed offsets (shifted views of one padded buffer): the autocorrelation products
static double check_rows(bool bk, int M, int N, int K, int nsplit)
{
    // buffer of R rows of length L; B row (BK: n, else k) starts at an arbitrary 4-byte aligned offset
    const int nrows = bk ? N : K, len = bk ? K : N;
    const int L = len + 37;
    std::vector<float> A((size_t)M * K), buf((size_t)nrows * L + 64), C((size_t)nsplit * M * N);
    std::vector<long long> off(nrows);
    for (auto &v : A) v = frand();
    for (auto &v : buf) v = frand();
    for (int r = 0; r < nrows; ++r) off[r] = (long long)r * L + (r * 7) % 31;
    float *dA, *dB, *dC;
    long long *doff;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, buf.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&doff, nrows * 8);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, buf.data(), buf.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, off.data(), nrows * 8, hipMemcpyHostToDevice);
    SgemmArgs a{dA, dB, dC, nullptr, M, N, K, K, 0, N, 0, 0, (long long)nsplit * M * N, (long long)M * N, nsplit,
                ((K + nsplit - 1) / nsplit + 15) / 16 * 16, 0, doff};
    sgemm_f32_rows64(0, bk, a, 1);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int m = 0; m < M; m += 3)
        for (int n = 0; n < N; n += 5) {
            double ref = 0, got = 0;
            for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (bk ? buf[off[n] + k] : buf[off[k] + n]);
            for (int s = 0; s < nsplit; ++s) got += C[((size_t)s * M + m) * N + n];
            worst = fmax(worst, fabs(got - ref));
        }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(doff);
    return worst;
}

static void time_rows(const char *name, bool bk, int M, int N, int K, int Z, int nsplit, int L)
{
    const int nrows = bk ? N : K;
    float *dA, *dB, *dC;
    long long *doff;
    std::vector<long long> off(nrows);
    for (int r = 0; r < nrows; ++r) off[r] = (long long)(r % 64) * L + 700 + (r / 64) * 3;
    const size_t nB = (size_t)Z * 64 * L + 4096;
    hipMalloc(&dA, (size_t)Z * M * K * 4); hipMalloc(&dB, nB * 4); hipMalloc(&dC, (size_t)Z * nsplit * M * N * 4); hipMalloc(&doff, nrows * 8);
    hipMemset(dA, 0, (size_t)Z * M * K * 4); hipMemset(dB, 0, nB * 4);
    hipMemcpy(doff, off.data(), nrows * 8, hipMemcpyHostToDevice);
    SgemmArgs a{dA, dB, dC, nullptr, M, N, K, K, 0, N, bk ? (long long)64 * L : 0, (long long)64 * L, (long long)nsplit * M * N, (long long)M * N,
                nsplit, ((K + nsplit - 1) / nsplit + 15) / 16 * 16, 0, doff};
    if (bk) a.lda = L;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) sgemm_f32_rows64(0, bk, a, Z);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) sgemm_f32_rows64(0, bk, a, Z);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-34s M %5d N %5d K %5d Z %d split %d: %.3f ms  %.1f TF/s\n", name, M, N, K, Z, nsplit, ms, 2.0 * M * N * K * Z / ms / 1e9);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(doff);
}

int main()
{
    printf("rows64: BK %.2e  free-contig %.2e  BK split3 %.2e  edges %.2e\n", check_rows(true, 64, 192, 100, 1), check_rows(false, 64, 200, 96, 1),
           check_rows(true, 50, 130, 333, 3), check_rows(false, 40, 70, 50, 1));
    time_rows("gram fwd  R = X Xshift^T (13 d)", true, 64, 832, 32400, 8, 8, 32400);
    time_rows("gram fwd  (13 d) split 16", true, 64, 832, 32400, 8, 16, 32400);
    time_rows("gram fwd  (13 d) split 24", true, 64, 832, 32400, 8, 24, 32400);
    time_rows("gram fwd  (25 d)", true, 64, 1600, 32400, 8, 8, 32400);
    time_rows("gram bwd  dX = K1 Xshift (25 d)", false, 64, 32400, 1600, 8, 1, 32400);
    printf("max abs err  NN %.2e  NT-ish(ak) %.2e  TN(bk only) %.2e  (ak,bk) split3 %.2e  edges %.2e\n",
           check(false, false, 256, 384, 64, 2, 1), check(true, false, 256, 256, 80, 1, 1), check(false, true, 128, 256, 48, 2, 1),
           check(true, true, 256, 384, 200, 2, 3), check(true, false, 200, 100, 40, 1, 1));
    // 256x256 @ 24x80, batch 8
    timeit("fwd   Y = W col        (ak, -)", true, false, 256, 1920, 2304, 8, 1, 0, 2304LL * 1920);
    timeit("T = W^T dY             (-, -)", false, false, 2304, 1920, 256, 8, 1, 0, 256LL * 1920);
    timeit("dW = dY col^T         (ak, bk)", true, true, 256, 2304, 1920, 8, 2, 256LL * 1920, 2304LL * 1920);
    // 512x256 @ 12x40
    timeit("fwd 512", true, false, 256, 480, 4608, 8, 1, 0, 4608LL * 480);
    timeit("T 512", false, false, 4608, 480, 256, 8, 1, 0, 256LL * 480);
    timeit("dW 512", true, true, 256, 4608, 480, 8, 1, 256LL * 480, 4608LL * 480);
    // 128x128 @ 48x160
    timeit("fwd 128", true, false, 128, 7680, 1152, 8, 1, 0, 1152LL * 7680);
    timeit("T 128", false, false, 1152, 7680, 128, 8, 1, 0, 128LL * 7680);
    timeit("dW 128", true, true, 128, 1152, 7680, 8, 8, 128LL * 7680, 1152LL * 7680);
    return 0;
}
