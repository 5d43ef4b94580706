// Per-object rows of the training loss on the GPU: one wave per object slot (include/dcd_hip.h, dcd_loss_rows_*).
// The arithmetic is in loss_rows_math.h; this file holds the wave primitives, the kernels and the C entry points.
//
// The section it replaces was ~550 ATen launches per step (detector_loss.py's per-term slices, masks, decodes and
// their autograd mirror images: profiles/step_r03_v2_kernels.csv), each a few microseconds of dependent latency on
// 320 slots of work.  Here a slot is one wave: the 73 dense keypoints and the 1500 pair depths are spread over the
// lanes and reduced with wave shuffles, everything else is a few hundred scalar operations.
#include <hip/hip_runtime.h>

#include "loss_rows_math.h"

namespace {

struct Wave64 {
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ int lanes() const { return 64; }
    __device__ __forceinline__ float sum(float v) const
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    }
    __device__ __forceinline__ int min_int(int v) const
    {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const int o = __shfl_xor(v, m, 64);
            v = o < v ? o : v;
        }
        return v;
    }
    __device__ __forceinline__ bool any(bool v) const { return __ballot(v) != 0ull; }
};

__global__ __launch_bounds__(64) void loss_rows_prepare_kernel(const dcd_loss_rows_args a)
{
    lr_prepare_row(a, blockIdx.x, Wave64());
}

template <bool BWD>
__global__ __launch_bounds__(64) void loss_rows_kernel(const dcd_loss_rows_args a)
{
    lr_row<BWD>(a, blockIdx.x, Wave64());
}

// sums[c] = sum over the slots of column c; the 3-D IoU column is filled in here from the per-slot metric (a non-finite
// IoU of an empty slot must not reach the sum, hence the select instead of a product with the mask).
__global__ __launch_bounds__(64) void loss_rows_sum_kernel(const dcd_loss_rows_args a)
{
    const int c = blockIdx.x, BM = a.B * a.M;
    float acc = 0.f;
    for (int s = threadIdx.x; s < BM; s += 64) {
        float v = a.cols[(size_t)c * BM + s];
        if (c == LR_IOU3D) {
            v = a.reg_mask[s] ? a.iou3d[s] : 0.f;
            a.cols[(size_t)c * BM + s] = v;
        }
        acc += v;
    }
    acc = Wave64().sum(acc);
    if (threadIdx.x == 0) a.sums[c] = acc;
}

// grad_pois += gradients that arrive through the edge solver: its 2-D inputs were (kpts + centre + offset) * 4 - pad
__global__ void loss_rows_finish_kernel(const dcd_loss_rows_args a)
{
    const int s = blockIdx.x, K = a.K;
    if (!a.reg_mask[s]) return;                         // the solver's gradient of an empty slot is zero anyway
    float *gp = a.grad_pois + (size_t)s * a.C;
    for (int i = threadIdx.x; i < K * 2; i += blockDim.x) gp[a.ch_kpts2d + i] += 4.f * a.grad_kps[(size_t)s * K * 2 + i];
    for (int i = threadIdx.x; i < K * 3; i += blockDim.x) gp[a.ch_kpts3d + i] += a.grad_kps3d[(size_t)s * K * 3 + i];
}

bool sizes_ok(const dcd_loss_rows_args *a)
{
    return a && a->B > 0 && a->M > 0 && a->C > 0 && a->K > 0 && a->K <= 128 && a->NP > 0 && a->num_classes > 0;
}

bool inputs_ok(const dcd_loss_rows_args *a)
{
    return a->pois && a->reg_mask && a->trunc_mask && a->find_pcl && a->ori_mask && a->cls_ids && a->centers && a->pad_size &&
           a->bboxes && a->locations && a->rotys && a->offset_3D && a->dimensions && a->orientations && a->keypoints &&
           a->kp_depth_mask && a->kpts2d && a->kpts3d && a->calib_P && a->calib && a->dim_mean;
}

bool channels_ok(const dcd_loss_rows_args *a)
{
    const int first[11] = {a->ch_box2d, a->ch_offset, a->ch_corner, a->ch_corner_unc, a->ch_dims, a->ch_ori_cls, a->ch_ori_off,
                           a->ch_depth, a->ch_depth_unc, a->ch_kpts2d, a->ch_kpts3d};
    const int width[11] = {4, 2, 2 * LR_NKP, 3, 3, 2 * LR_NBIN, 2 * LR_NBIN, 1, 1, 2 * a->K, 3 * a->K};
    int total = 0;
    for (int i = 0; i < 11; ++i) {
        if (first[i] < 0 || first[i] + width[i] > a->C) return false;
        for (int j = 0; j < i; ++j)
            if (first[i] < first[j] + width[j] && first[j] < first[i] + width[i]) return false;      // heads must not overlap
        total += width[i];
    }
    return total == a->C;      // the backward writes exactly these channels: there must be no others
}

}  // namespace

extern "C" {

int dcd_loss_rows_prepare(void *stream_, const dcd_loss_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!sizes_ok(a) || !inputs_ok(a) || !channels_ok(a)) return DCD_ERR_BAD_ARG;
    if (!a->kps_pred || !a->kps_tgt || !a->kps3d_pred || !a->kps3d_tgt || !a->rot || !a->P_rows || !a->kmask) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_rows_prepare_kernel, dim3(a->B * a->M), dim3(64), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_loss_rows_forward(void *stream_, const dcd_loss_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!sizes_ok(a) || !inputs_ok(a) || !channels_ok(a)) return DCD_ERR_BAD_ARG;
    if (!a->pair_depth || !a->pair_mask || !a->cols || !a->corners_pred || !a->corners_tgt || !a->iou3d || !a->sums)
        return DCD_ERR_BAD_ARG;
    const int BM = a->B * a->M;
    hipLaunchKernelGGL(loss_rows_kernel<false>, dim3(BM), dim3(64), 0, stream, *a);
    const int st = dcd_iou3d(stream_, a->corners_pred, a->corners_tgt, BM, a->iou3d);
    if (st != DCD_OK) return st;
    hipLaunchKernelGGL(loss_rows_sum_kernel, dim3(DCD_LOSS_ROWS_NCOL), dim3(64), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_loss_rows_backward(void *stream_, const dcd_loss_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!sizes_ok(a) || !inputs_ok(a) || !channels_ok(a)) return DCD_ERR_BAD_ARG;
    if (!a->pair_depth || !a->pair_mask || !a->grad_sums || !a->grad_pois || !a->grad_pair) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_rows_kernel<true>, dim3(a->B * a->M), dim3(64), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

int dcd_loss_rows_finish(void *stream_, const dcd_loss_rows_args *a)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!sizes_ok(a) || !channels_ok(a) || !a->reg_mask || !a->grad_pois || !a->grad_kps || !a->grad_kps3d) return DCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(loss_rows_finish_kernel, dim3(a->B * a->M), dim3(128), 0, stream, *a);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
