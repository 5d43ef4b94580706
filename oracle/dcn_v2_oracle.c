/*
 * oracle/dcn_v2_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's modulated deformable convolution (DCNv2)
 * forward and backward.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported
 * CPU baseline.  The product path (dcd_amd/) never links or calls it.
 *
 * Reference (paths relative to /root/reference/DGDE/model/backbone/DCNv2/DCN/src):
 *   forward  host : cpu/dcn_v2_cpu.cpp:22-111   (bias, im2col, W*col, per sample)
 *   backward host : cpu/dcn_v2_cpu.cpp:113-229  (col=W^T dY, coord, col2im, im2col, dW, db)
 *   bilinear      : cpu/dcn_v2_im2col_cpu.cpp:27-56
 *   grad weight   : cpu/dcn_v2_im2col_cpu.cpp:58-82
 *   coord weight  : cpu/dcn_v2_im2col_cpu.cpp:84-125
 *   im2col        : cpu/dcn_v2_im2col_cpu.cpp:127-196
 *   col2im        : cpu/dcn_v2_im2col_cpu.cpp:198-257
 *   col2im_coord  : cpu/dcn_v2_im2col_cpu.cpp:259-329
 * The CUDA kernels (cuda/dcn_v2_im2col_cuda.cu:25-327) have the same bodies.
 *
 * Parity pin: the reference C++ cannot be built in this image (it includes
 * <TH/TH.h>, removed from PyTorch; building it would need a stand-in header,
 * which is not allowed).  This restatement is pinned instead by the
 * reference's own tests, restated in tests/test_oracle_dcn.py:
 *   - check_zero_offset   (DCN/testcpu.py:32-67)   known answer, exact
 *   - check_gradient_dconv(DCN/testcpu.py:69-97)   backward == d(forward), run on
 *     the f64 instantiation below where finite differences are meaningful.
 *
 * The file is compiled twice through the REAL macro: f32 (the reference's
 * arithmetic type: `using scalar_t = float`, dcn_v2_cpu.cpp:73) and f64
 * (gradcheck only).  Loops marked `omp` are parallel only over indices that
 * own their outputs, so results are identical with or without OpenMP; set
 * OMP_NUM_THREADS=1 for the reference-faithful serial timing (variant A in
 * BASELINE.md) and leave it unset for variant B.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#define SUF(name) name##_f32
#endif

typedef REAL real;

/* cpu/dcn_v2_im2col_cpu.cpp:27-56 */
static real SUF(bilinear)(const real *im, int data_width, int height, int width, real h, real w)
{
    int h_low = (int)floor(h), w_low = (int)floor(w);
    int h_high = h_low + 1, w_high = w_low + 1;
    real lh = h - h_low, lw = w - w_low;
    real hh = 1 - lh, hw = 1 - lw;
    real v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * data_width + w_low];
    if (h_low >= 0 && w_high <= width - 1) v2 = im[h_low * data_width + w_high];
    if (h_high <= height - 1 && w_low >= 0) v3 = im[h_high * data_width + w_low];
    if (h_high <= height - 1 && w_high <= width - 1) v4 = im[h_high * data_width + w_high];
    real w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

/* cpu/dcn_v2_im2col_cpu.cpp:58-82 */
static real SUF(grad_weight)(real ah, real aw, int h, int w, int height, int width)
{
    if (ah <= -1 || ah >= height || aw <= -1 || aw >= width) return 0;
    int hl = (int)floor(ah), wl = (int)floor(aw);
    int hh = hl + 1, wh = wl + 1;
    real weight = 0;
    if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
    if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
    if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
    if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
    return weight;
}

/* cpu/dcn_v2_im2col_cpu.cpp:84-125 */
static real SUF(coord_weight)(real ah, real aw, int height, int width, const real *im,
                              int data_width, int bp_dir)
{
    if (ah <= -1 || ah >= height || aw <= -1 || aw >= width) return 0;
    int hl = (int)floor(ah), wl = (int)floor(aw);
    int hh = hl + 1, wh = wl + 1;
    real weight = 0;
    if (bp_dir == 0) {
        if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * im[hl * data_width + wl];
        if (hl >= 0 && wh <= width - 1) weight += -1 * (aw - wl) * im[hl * data_width + wh];
        if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - aw) * im[hh * data_width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (aw - wl) * im[hh * data_width + wh];
    } else {
        if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * im[hl * data_width + wl];
        if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - ah) * im[hl * data_width + wh];
        if (hh <= height - 1 && wl >= 0) weight += -1 * (ah - hl) * im[hh * data_width + wl];
        if (hh <= height - 1 && wh <= width - 1) weight += (ah - hl) * im[hh * data_width + wh];
    }
    return weight;
}

typedef struct {
    int C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg;
} SUF(geom);

/* cpu/dcn_v2_im2col_cpu.cpp:127-196, batch_size == 1 as the host calls it
 * (dcn_v2_cpu.cpp:93-100).  col is (C*kh*kw, Ho*Wo). */
static void SUF(im2col)(const real *im, const real *off, const real *msk, const SUF(geom) *g, real *col)
{
    const int HoWo = g->Ho * g->Wo, cpg = g->C / g->dg, KK = g->kh * g->kw;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < g->C; ++c) {
        const int grp = c / cpg;
        const real *imc = im + (size_t)c * g->H * g->W;
        const real *offg = off + (size_t)grp * 2 * KK * HoWo;
        const real *mskg = msk + (size_t)grp * KK * HoWo;
        for (int ho = 0; ho < g->Ho; ++ho)
            for (int wo = 0; wo < g->Wo; ++wo) {
                const int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
                for (int i = 0; i < g->kh; ++i)
                    for (int j = 0; j < g->kw; ++j) {
                        const int t = i * g->kw + j;
                        const real oh = offg[(size_t)(2 * t) * HoWo + ho * g->Wo + wo];
                        const real ow = offg[(size_t)(2 * t + 1) * HoWo + ho * g->Wo + wo];
                        const real m = mskg[(size_t)t * HoWo + ho * g->Wo + wo];
                        const real h_im = h_in + i * g->dh + oh;
                        const real w_im = w_in + j * g->dw + ow;
                        real val = 0;
                        if (h_im > -1 && w_im > -1 && h_im < g->H && w_im < g->W)
                            val = SUF(bilinear)(imc, g->W, g->H, g->W, h_im, w_im);
                        col[((size_t)c * KK + t) * HoWo + ho * g->Wo + wo] = val * m;
                    }
            }
    }
}

/* cpu/dcn_v2_im2col_cpu.cpp:198-257.  The reference truncates the sampling
 * position with (int) and scans a 5x5 window; that is reproduced literally.
 * Parallel over c only: channel c writes grad_im[c] only. */
static void SUF(col2im)(const real *col, const real *off, const real *msk, const SUF(geom) *g, real *grad_im)
{
    const int HoWo = g->Ho * g->Wo, cpg = g->C / g->dg, KK = g->kh * g->kw;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < g->C; ++c) {
        const int grp = c / cpg;
        const real *offg = off + (size_t)grp * 2 * KK * HoWo;
        const real *mskg = msk + (size_t)grp * KK * HoWo;
        real *gim = grad_im + (size_t)c * g->H * g->W;
        for (int i = 0; i < g->kh; ++i)
            for (int j = 0; j < g->kw; ++j) {
                const int t = i * g->kw + j;
                for (int ho = 0; ho < g->Ho; ++ho)
                    for (int wo = 0; wo < g->Wo; ++wo) {
                        const int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
                        const real oh = offg[(size_t)(2 * t) * HoWo + ho * g->Wo + wo];
                        const real ow = offg[(size_t)(2 * t + 1) * HoWo + ho * g->Wo + wo];
                        const real m = mskg[(size_t)t * HoWo + ho * g->Wo + wo];
                        const real ih = h_in + i * g->dh + oh;
                        const real iw = w_in + j * g->dw + ow;
                        const real top = col[((size_t)c * KK + t) * HoWo + ho * g->Wo + wo] * m;
                        const int ch = (int)ih, cw = (int)iw;
                        for (int dy = -2; dy <= 2; ++dy)
                            for (int dx = -2; dx <= 2; ++dx)
                                if (ch + dy >= 0 && ch + dy < g->H && cw + dx >= 0 && cw + dx < g->W &&
                                    fabs(ih - (ch + dy)) < 1 && fabs(iw - (cw + dx)) < 1) {
                                    real wgt = SUF(grad_weight)(ih, iw, ch + dy, cw + dx, g->H, g->W);
                                    gim[(ch + dy) * g->W + cw + dx] += wgt * top;
                                }
                    }
            }
    }
}

/* cpu/dcn_v2_im2col_cpu.cpp:259-329, batch_size == 1.  One output per
 * (offset channel, ho, wo); serial sum over the input channels of the group. */
static void SUF(col2im_coord)(const real *col, const real *im, const real *off, const real *msk,
                              const SUF(geom) *g, real *grad_off, real *grad_msk)
{
    const int HoWo = g->Ho * g->Wo, cpg = g->C / g->dg, KK = g->kh * g->kw;
    const int offc = 2 * KK * g->dg;
#pragma omp parallel for schedule(static)
    for (int oc = 0; oc < offc; ++oc) {
        const int grp = oc / (2 * KK);
        const int lc = oc - grp * 2 * KK;
        const int t = lc / 2, bp_dir = lc % 2;
        const int i = t / g->kw, j = t % g->kw;
        const real *offg = off + (size_t)grp * 2 * KK * HoWo;
        const real *mskg = msk + (size_t)grp * KK * HoWo;
        for (int ho = 0; ho < g->Ho; ++ho)
            for (int wo = 0; wo < g->Wo; ++wo) {
                const int h_in = ho * g->sh - g->ph, w_in = wo * g->sw - g->pw;
                const real oh = offg[(size_t)(2 * t) * HoWo + ho * g->Wo + wo];
                const real ow = offg[(size_t)(2 * t + 1) * HoWo + ho * g->Wo + wo];
                const real m = mskg[(size_t)t * HoWo + ho * g->Wo + wo];
                real ih = h_in + i * g->dh + oh;
                real iw = w_in + j * g->dw + ow;
                const int outside = (ih <= -1 || iw <= -1 || ih >= g->H || iw >= g->W);
                if (outside) ih = iw = -2;
                real val = 0, mval = 0;
                for (int cc = 0; cc < cpg; ++cc) {
                    const int c = grp * cpg + cc;
                    const real *imc = im + (size_t)c * g->H * g->W;
                    const real cv = col[((size_t)c * KK + t) * HoWo + ho * g->Wo + wo];
                    if (!outside) mval += cv * SUF(bilinear)(imc, g->W, g->H, g->W, ih, iw);
                    val += SUF(coord_weight)(ih, iw, g->H, g->W, imc, g->W, bp_dir) * cv * m;
                }
                grad_off[(size_t)oc * HoWo + ho * g->Wo + wo] = val;
                if (bp_dir == 0)
                    grad_msk[((size_t)grp * KK + t) * HoWo + ho * g->Wo + wo] = mval;
            }
    }
}

/* C(MxN) = beta*C + A(MxK) * B(KxN), row major, transposition flags on A/B. */
static void SUF(gemm)(int ta, int tb, int M, int N, int K, const real *A, int lda, const real *B, int ldb,
                      real beta, real *C, int ldc)
{
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        real *c = C + (size_t)m * ldc;
        if (beta == 0) memset(c, 0, sizeof(real) * N);
        if (!tb) {
            for (int k = 0; k < K; ++k) {
                const real a = ta ? A[(size_t)k * lda + m] : A[(size_t)m * lda + k];
                const real *b = B + (size_t)k * ldb;
                for (int n = 0; n < N; ++n) c[n] += a * b[n];
            }
        } else {
            for (int n = 0; n < N; ++n) {
                const real *b = B + (size_t)n * ldb;
                real acc = 0;
                if (!ta) { const real *a = A + (size_t)m * lda; for (int k = 0; k < K; ++k) acc += a[k] * b[k]; }
                else for (int k = 0; k < K; ++k) acc += A[(size_t)k * lda + m] * b[k];
                c[n] += acc;
            }
        }
    }
}

static int SUF(setup)(SUF(geom) *g, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                      int dh, int dw, int dg)
{
    if (C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dg <= 0 || C % dg) return 1;
    g->C = C; g->H = H; g->W = W; g->kh = kh; g->kw = kw; g->sh = sh; g->sw = sw;
    g->ph = ph; g->pw = pw; g->dh = dh; g->dw = dw; g->dg = dg;
    g->Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1; /* dcn_v2_cpu.cpp:65-66 */
    g->Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    return (g->Ho <= 0 || g->Wo <= 0) ? 1 : 0;
}

/* dcn_v2_cpu.cpp:22-111.  output (B,Cout,Ho,Wo) = bias + W_flat * columns, per sample. */
int SUF(dcn_oracle_forward)(const real *input, const real *weight, const real *bias, const real *offset,
                            const real *mask, int B, int C, int H, int W, int Cout, int kh, int kw, int sh,
                            int sw, int ph, int pw, int dh, int dw, int dg, real *output)
{
    SUF(geom) g;
    if (SUF(setup)(&g, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return 1;
    const int HoWo = g.Ho * g.Wo, K = C * kh * kw, KK = kh * kw;
    real *col = (real *)malloc(sizeof(real) * (size_t)K * HoWo);
    if (!col) return 2;
    for (int b = 0; b < B; ++b) {
        real *out = output + (size_t)b * Cout * HoWo;
        for (int o = 0; o < Cout; ++o)
            for (int p = 0; p < HoWo; ++p) out[(size_t)o * HoWo + p] = bias[o];
        SUF(im2col)(input + (size_t)b * C * H * W, offset + (size_t)b * dg * 2 * KK * HoWo,
                    mask + (size_t)b * dg * KK * HoWo, &g, col);
        SUF(gemm)(0, 0, Cout, HoWo, K, weight, K, col, HoWo, (real)1, out, HoWo);
    }
    free(col);
    return 0;
}

/* dcn_v2_cpu.cpp:113-229.  Return order grad_input, grad_offset, grad_mask,
 * grad_weight, grad_bias (:226-228).  All five are overwritten. */
int SUF(dcn_oracle_backward)(const real *input, const real *weight, const real *bias, const real *offset,
                             const real *mask, const real *grad_output, int B, int C, int H, int W, int Cout,
                             int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                             real *grad_input, real *grad_offset, real *grad_mask, real *grad_weight,
                             real *grad_bias)
{
    (void)bias;
    SUF(geom) g;
    if (SUF(setup)(&g, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg)) return 1;
    const int HoWo = g.Ho * g.Wo, K = C * kh * kw, KK = kh * kw;
    real *col = (real *)malloc(sizeof(real) * (size_t)K * HoWo);
    if (!col) return 2;
    memset(grad_input, 0, sizeof(real) * (size_t)B * C * H * W);
    memset(grad_offset, 0, sizeof(real) * (size_t)B * dg * 2 * KK * HoWo);
    memset(grad_mask, 0, sizeof(real) * (size_t)B * dg * KK * HoWo);
    memset(grad_weight, 0, sizeof(real) * (size_t)Cout * K);
    memset(grad_bias, 0, sizeof(real) * (size_t)Cout);
    for (int b = 0; b < B; ++b) {
        const real *in_b = input + (size_t)b * C * H * W;
        const real *off_b = offset + (size_t)b * dg * 2 * KK * HoWo;
        const real *msk_b = mask + (size_t)b * dg * KK * HoWo;
        const real *gy_b = grad_output + (size_t)b * Cout * HoWo;
        /* columns = W_flat^T (K x Cout) * dY_b (Cout x HoWo)            :179-182 */
        SUF(gemm)(1, 0, K, HoWo, Cout, weight, K, gy_b, HoWo, (real)0, col, HoWo);
        /* gradient w.r.t. sampling coordinates and mask                :185-194 */
        SUF(col2im_coord)(col, in_b, off_b, msk_b, &g, grad_offset + (size_t)b * dg * 2 * KK * HoWo,
                          grad_mask + (size_t)b * dg * KK * HoWo);
        /* gradient w.r.t. input data                                   :196-203 */
        SUF(col2im)(col, off_b, msk_b, &g, grad_input + (size_t)b * C * H * W);
        /* columns recomputed for the weight gradient                   :206-213 */
        SUF(im2col)(in_b, off_b, msk_b, &g, col);
        /* grad_weight += dY_b (Cout x HoWo) * columns^T (HoWo x K)      :216-217 */
        SUF(gemm)(0, 1, Cout, K, HoWo, gy_b, HoWo, col, HoWo, (real)1, grad_weight, K);
        /* grad_bias += dY_b * ones                                     :221-223 */
        for (int o = 0; o < Cout; ++o) {
            real s = 0;
            for (int p = 0; p < HoWo; ++p) s += gy_b[(size_t)o * HoWo + p];
            grad_bias[o] += s;
        }
    }
    free(col);
    return 0;
}
