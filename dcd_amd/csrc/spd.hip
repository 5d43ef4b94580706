// Batched symmetric-positive-definite solve  y = S^-1 r  (one right-hand side per matrix) on the matrix pipe: blocked
// right-looking Cholesky (128-wide panels) whose panel and trailing updates are sgemm_f32.inc products, plus two blocked
// substitutions that use the inverses of the diagonal blocks.  Replaces, in the backward of GMW's optimal-transport layer
// (GMW/lib/optimal_transport.py:102-128: torch.cholesky + cholesky_inverse of the n x n Schur complement, n = 2628 edges),
// MAGMA's potrf (8.7 ms for eight systems) and the inverse of the factor (5.0 ms) -- both latency-bound there.
//
//   per 128-block k:  spd_diag_block   L_kk = chol(A_kk) in LDS (left-looking, one workgroup per matrix), D_k = L_kk^-1
//                     panel            L_ik = A_ik D_k^T           (sgemm, in place: one 128-column tile)
//                     trailing         A_ij -= L_ik L_jk^T, i >= j (sgemm, alpha = -1, accumulate, lower tiles only)
//   forward solve:    free -- the right-hand side rides along as row n of the matrix (it becomes z = L^-1 r)
//   backward solve:   spd_back_block   per block from the last: y_k = D_k^T z_k, z[0:k0] -= L[k-rows][0:k0]^T y_k
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dcd_hip.h"
#include "lds_limit.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#include "sgemm_f32.inc"

namespace {

constexpr int SP_NB = 128;
constexpr int SP_LS = SP_NB + 1;         // LDS row stride

// One workgroup per matrix factorises the diagonal block at (k0, k0) and inverts the factor, both in LDS and both blocked by
// 32 columns so that the barrier-separated steps stay short.  Measured 170 us per block at 8 matrices -- the same as a first
// version with 8 x 8 register tiles and one rank-1 update of the whole block per column, i.e. the time is not set by the
// arithmetic (0.7 MFLOP + 0.7 MFLOP per block) but by running ~300 dependent, barrier-separated steps on eight of 256 CUs:
//   Cholesky  per 32-column panel: 32 x (pivot, scale the column, rank-1 update of the REST OF THE PANEL only: <= 16 FMAs per
//             thread), then one rank-32 update of the trailing block from registers (8 x 8 tile per thread, no barrier inside);
//   inverse   X = L^-1 by block rows of 32: T = E - L[i, <i] X[<i, :] (4 x 4 tile per thread), then L_ii x = t per column
//             (one thread per column, 32 unknowns in registers, fully unrolled: no barrier inside).
// A narrower last block is padded with the identity.  A: batch x rows x n (row stride n, matrix stride `mstride`); lower
// triangle read, overwritten by L.  LDS: two 128 x 129 float arrays (dynamic, 132 KB).
constexpr int SP_IB = 32;

__global__ __launch_bounds__(256) void spd_diag_block(float *__restrict__ A, int n, long long mstride, int k0, int nb,
                                                      float *__restrict__ dinv, int blk, int nblk, int *__restrict__ info)
{
    extern __shared__ float lds[];
    float *Ls = lds, *Xs = lds + SP_NB * SP_LS;
    __shared__ int bad_flag;
    __shared__ float colbuf[2][SP_IB];
    const int b = blockIdx.x, tid = threadIdx.x;
    float *Ab = A + (size_t)b * mstride + (size_t)k0 * n + k0;
    if (tid == 0) bad_flag = 0;
#ifdef SPD_PROFILE
    long long tc[6];
    tc[0] = clock64();
#endif
    for (int e0 = tid; e0 < SP_NB * SP_NB; e0 += 8 * 256) {       // eight independent loads in flight, then the LDS stores
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, row = e >> 7, col = e & 127;
            v[u] = (row < nb && col <= row) ? Ab[(size_t)row * n + col] : (row == col ? 1.f : 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u;
            Ls[(e >> 7) * SP_LS + (e & 127)] = v[u];
        }
    }
    __syncthreads();
#ifdef SPD_PROFILE
    tc[1] = clock64();
#endif
    const int ti = tid >> 4, tj = tid & 15;                     // 8 x 8 tiles of the trailing update
    for (int c0 = 0; c0 < SP_NB; c0 += SP_IB) {
        // panel: columns c0 .. c0+31, rows >= c0.  Thread = row, its 32 panel entries in registers; per column ONE barrier: the
        // panel's rows publish their (unscaled) entry of column j, everyone reads the pivot and the <= 31 entries it needs as
        // broadcasts and updates its own registers.  (A version that updated the panel in LDS -- read-modify-write per entry --
        // serialised on the LDS round trips: 2 000 cycles per column, 260 k for the block.)
        {
            const int i = tid;                                   // row (threads 128..255 idle here)
            float a[SP_IB];
            if (i < SP_NB) {
#pragma unroll
                for (int c = 0; c < SP_IB; ++c) a[c] = Ls[i * SP_LS + c0 + c];
            }
#pragma unroll
            for (int jj = 0; jj < SP_IB; ++jj) {
                const int j = c0 + jj, cur = jj & 1;
                if (i >= c0 && i < c0 + SP_IB) colbuf[cur][i - c0] = a[jj];
                __syncthreads();
                float piv = colbuf[cur][jj];
                // not positive definite: poison the factor (NaN) so that the solution is NaN and the step's non-finite guard skips
                // the update -- a clamped pivot gave a finite but meaningless y that trained on (advisor r2)
                if (!(piv > 0.f)) { bad_flag = 1; piv = __builtin_nanf(""); }
                const float inv = __frsqrt_rn(piv);
                if (i < SP_NB && i >= j) {
                    const float lij = i == j ? piv * inv : a[jj] * inv;
                    a[jj] = lij;
                    if (i > j) {
#pragma unroll
                        for (int cc = jj + 1; cc < SP_IB; ++cc) a[cc] -= lij * (colbuf[cur][cc] * inv);
                    }
                }
            }
            __syncthreads();
            if (i >= c0 && i < SP_NB) {
#pragma unroll
                for (int c = 0; c < SP_IB; ++c) Ls[i * SP_LS + c0 + c] = (c0 + c <= i) ? a[c] : 0.f;
            }
        }
        __syncthreads();
        // trailing block (rows, columns >= c0+32; lower tiles only): A -= L_panel L_panel^T
        const int t0 = c0 + SP_IB;
        if (t0 < SP_NB && ti * 8 + 7 >= t0 && tj * 8 + 7 >= t0 && tj <= ti) {
            float acc[8][8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
            for (int k = c0; k < c0 + SP_IB; ++k) {
                float lr[8], lc[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) lr[r] = Ls[(ti * 8 + r) * SP_LS + k];
#pragma unroll
                for (int c = 0; c < 8; ++c) lc[c] = Ls[(tj * 8 + c) * SP_LS + k];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[r][c] += lr[r] * lc[c];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int row = ti * 8 + r, col = tj * 8 + c;
                    if (row >= t0 && col >= t0 && col <= row) Ls[row * SP_LS + col] -= acc[r][c];
                }
        }
        __syncthreads();
    }
#ifdef SPD_PROFILE
    tc[2] = clock64();
#endif
    if (tid == 0 && bad_flag && info) atomicMax(info + b, k0 + 1);      // not positive definite (or NaN) somewhere in this block
    for (int e = tid; e < SP_NB * SP_NB; e += 256) {
        const int row = e >> 7, col = e & 127;
        if (row < nb && col <= row) Ab[(size_t)row * n + col] = Ls[row * SP_LS + col];
    }
#ifdef SPD_PROFILE
    tc[3] = clock64();
#endif
    // X = L^-1 by block rows
    const int tr = tid >> 5, tcol = tid & 31;                   // 4 x 4 tiles of a 32 x 128 block row
    for (int r0 = 0; r0 < SP_NB; r0 += SP_IB) {
        {
            float acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][c] = (r0 + 4 * tr + a == 4 * tcol + c) ? 1.f : 0.f;
            if (4 * tcol < r0) {                                  // X[k][c] = 0 for c > k: columns >= r0 see no earlier rows
                for (int k = 0; k < r0; ++k) {
                    float lr[4], xc[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) lr[a] = Ls[(r0 + 4 * tr + a) * SP_LS + k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) xc[c] = Xs[k * SP_LS + 4 * tcol + c];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[a][c] -= lr[a] * xc[c];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) Xs[(r0 + 4 * tr + a) * SP_LS + 4 * tcol + c] = acc[a][c];
        }
        __syncthreads();
        if (tid < r0 + SP_IB) {                                 // column tid: L_ii x = t (columns beyond the block row are zero)
            float x[SP_IB];
#pragma unroll
            for (int r = 0; r < SP_IB; ++r) {
                float s_ = Xs[(r0 + r) * SP_LS + tid];
#pragma unroll
                for (int k = 0; k < r; ++k) s_ -= Ls[(r0 + r) * SP_LS + r0 + k] * x[k];
                x[r] = s_ / Ls[(r0 + r) * SP_LS + r0 + r];
            }
#pragma unroll
            for (int r = 0; r < SP_IB; ++r) Xs[(r0 + r) * SP_LS + tid] = x[r];
        }
        __syncthreads();
    }
#ifdef SPD_PROFILE
    tc[4] = clock64();
#endif
    float *D = dinv + ((size_t)b * nblk + blk) * SP_NB * SP_NB;
    for (int e = tid; e < SP_NB * SP_NB; e += 256) {
        const int row = e >> 7, col = e & 127;
        D[e] = (row < nb && col <= row) ? Xs[row * SP_LS + col] : 0.f;
    }
#ifdef SPD_PROFILE
    tc[5] = clock64();
    if (tid == 0 && b == 0 && blk == 1 && info)
        for (int q = 0; q < 5; ++q) info[8 + q] = (int)(tc[q + 1] - tc[q]);
#endif
}

// Backward substitution, one launch per block (from the last one): y_k = D_k^T z_k, then z[0:k0] -= L[k-rows][0:k0]^T y_k.
// z = row n of the augmented matrix (the factorisation left L^-1 r there).  grid = (max(1, ceil(k0/256)), batch), block = 256.
__global__ __launch_bounds__(256) void spd_back_block(float *__restrict__ A, int n, long long mstride, const float *__restrict__ dinv,
                                                      int blk, int nblk, float *__restrict__ y)
{
    __shared__ float yk[SP_NB], zk[SP_NB];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int k0 = blk * SP_NB, nb = min(SP_NB, n - k0);
    float *Ab = A + (size_t)b * mstride;
    float *z = Ab + (size_t)n * n;
    if (tid < SP_NB) zk[tid] = tid < nb ? z[k0 + tid] : 0.f;
    __syncthreads();
    const float *D = dinv + ((size_t)b * nblk + blk) * SP_NB * SP_NB;
    if (tid < SP_NB) {
        float s = 0.f;
        for (int r = 0; r < nb; ++r) s += D[r * SP_NB + tid] * zk[r];     // consecutive threads read consecutive addresses
        yk[tid] = s;
        if (blockIdx.x == 0 && tid < nb) y[(size_t)b * n + k0 + tid] = s;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + tid;
    if (j < k0) {
        const float *Lk = Ab + (size_t)k0 * n + j;
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < nb; ++r) s += Lk[(size_t)r * n] * yk[r];
        z[j] -= s;
    }
}

}  // namespace

extern "C" {

int dcd_sgemm(void *stream_, const float *A, int lda, long long strideA, int a_kcontig, const float *B, int ldb, long long strideB,
              int b_kcontig, float *C, int ldc, long long strideC, int M, int N, int K, int Z, float alpha, int accumulate,
              int lower_only)
{
    (void)hipGetLastError();
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || Z <= 0 || ((lda | ldb) & 3) || ((strideA | strideB) & 3) ||
        (((uintptr_t)A | (uintptr_t)B) & 15))
        return DCD_ERR_BAD_ARG;
    SgemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.strideA = strideA; a.strideB = strideB; a.strideC = strideC; a.strideCs = 0;
    a.nsplit = 1; a.kchunk = (K + SG_K - 1) / SG_K * SG_K; a.ct = 0; a.b_off = nullptr;
    a.alpha = alpha; a.accumulate = accumulate; a.lower_only = lower_only;
    sgemm_f32((hipStream_t)stream_, a_kcontig != 0, b_kcontig != 0, a, Z);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

size_t dcd_spd_solve_workspace_bytes(int batch, int n)
{
    if (batch <= 0 || n <= 0) return 0;
    const size_t nblk = (size_t)(n + SP_NB - 1) / SP_NB;
    return (size_t)batch * nblk * SP_NB * SP_NB * sizeof(float) + 256;
}

int dcd_spd_solve(void *stream_, float *S, float *y, int batch, int n, int rows, int *info, void *workspace, size_t workspace_bytes)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!S || !y || !workspace || batch <= 0 || n <= 0 || (n & 3) || rows < n + 1 || ((uintptr_t)S & 15)) return DCD_ERR_BAD_ARG;
    if (workspace_bytes < dcd_spd_solve_workspace_bytes(batch, n)) return DCD_ERR_WORKSPACE;
    const int nblk = (n + SP_NB - 1) / SP_NB;
    float *dinv = (float *)workspace;
    const long long ms = (long long)rows * n;
    static LdsLimit lim_diag;
    const int lds_diag = 2 * SP_NB * SP_LS * (int)sizeof(float);
    if (!lim_diag.raise(lds_diag, spd_diag_block)) return DCD_ERR_LAUNCH;
    // Row n of every matrix holds the right-hand side: carried through the panel and trailing updates like a row of the matrix,
    // it ends up as z = L^-1 r (the last row of the Cholesky factor of [[S, r], [r^T, .]]) -- the forward substitution for free.
    for (int blk = 0; blk < nblk; ++blk) {
        const int k0 = blk * SP_NB, nb = n - k0 < SP_NB ? n - k0 : SP_NB;
        hipLaunchKernelGGL(spd_diag_block, dim3(batch), dim3(256), lds_diag, stream, S, n, ms, k0, nb, dinv, blk, nblk, info);
        const int rest = n + 1 - k0 - nb;                      // rows below the block, including the right-hand-side row
        SgemmArgs a;
        // panel: L_ik = A_ik D_k^T  (in place: one column tile)
        a.A = S + (size_t)(k0 + nb) * n + k0; a.B = dinv + (size_t)blk * SP_NB * SP_NB; a.C = S + (size_t)(k0 + nb) * n + k0;
        a.bias = nullptr; a.M = rest; a.N = nb; a.K = nb; a.lda = n; a.ldb = SP_NB; a.ldc = n;
        a.strideA = ms; a.strideB = (long long)nblk * SP_NB * SP_NB; a.strideC = ms; a.strideCs = 0;
        a.nsplit = 1; a.kchunk = SP_NB; a.ct = 0; a.b_off = nullptr;
        sgemm_f32(stream, true, true, a, batch);
        if (rest <= 1) break;
        // trailing update of the lower triangle (and of the right-hand-side row): A_ij -= L_ik L_jk^T
        a.A = S + (size_t)(k0 + nb) * n + k0; a.B = a.A; a.C = S + (size_t)(k0 + nb) * n + (k0 + nb);
        a.M = rest; a.N = rest - 1; a.K = nb; a.lda = n; a.ldb = n; a.ldc = n;
        a.strideA = ms; a.strideB = ms; a.strideC = ms;
        a.alpha = -1.f; a.accumulate = 1; a.lower_only = 1;
        sgemm_f32(stream, true, true, a, batch);
    }
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int k0 = blk * SP_NB;
        hipLaunchKernelGGL(spd_back_block, dim3(k0 > 0 ? (k0 + 255) / 256 : 1, batch), dim3(256), 0, stream, S, n, ms, dinv, blk, nblk, y);
    }
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
