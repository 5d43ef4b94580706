// targets.hip -- training-target encoding on the device.
//
// Replaces the per-sample numpy work of the reference's data loader, `KITTIDataset.__getitem__`
// (DGDE/data/datasets/kitti.py:354-606) with its helpers `gaussian_radius` / `draw_umich_gaussian` / `draw_umich_gaussian_2D`
// (DGDE/model/heatmap_coder.py:37-124), `encode_alpha_multibin` (kitti.py:225-244), `approx_proj_center`
// (kitti_utils.py:1040-1078), `get_edge_utils` (kitti.py:165-223): the raw label values of a whole batch go in, every
// ParamsList field comes out as a batch tensor, in one launch per batch (+ one for the border walk).
// One workgroup per (object slot, image): 83 points (8 corners, 2 face centres, 63 + ... key points) are projected by one thread
// each, thread 0 takes the scalar decisions, all threads write the per-point rows and splat the Gaussian with an integer
// atomicMax (non-negative floats order like their bit patterns), which makes the heat map independent of the object order
// exactly like the reference's np.maximum.  Arithmetic follows the reference: float64, except the float32 arrays it keeps
// (`obj.t`, `obj.box2d`).  HBM-trivial (a few hundred KB per batch): latency-bound by nature.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"

namespace {

constexpr int TG_MAXK = 128;       // key points per object (63 + 10 = 73 on the DGDE path)
constexpr int TG_THREADS = 128;

struct TgOut {
    float *hm;                 // (B, n_classes, fh, fw), zero-filled by the caller
    int32_t *cls_ids;          // (B, M)
    int32_t *target_centers;   // (B, M, 2)
    float *gt_bboxes, *bboxes; // (B, M, 4)
    float *keypoints;          // (B, M, 10, 3)
    float *kdm;                // (B, M, 3)
    float *ek2d, *ek3d;        // (B, M, K, 3)
    float *ekdm;               // (B, M, K)
    float *calib_p;            // (B, M, 12)
    uint8_t *find_pcl;         // (B, M)
    float *dimensions, *locations;   // (B, M, 3)
    float *rotys, *alphas;     // (B, M)
    float *offset_3d;          // (B, M, 2)
    double *occlusions, *truncations;   // (B, M)
    float *orientations;       // (B, M, 8)
    uint8_t *reg_mask, *trunc_mask;     // (B, M)
    float *reg_weight;         // (B, M)
};

struct TgCfg {
    int B, M, n_extra, in_w, in_h, down, num_bin, n_cls;
    double filter_trunc, filter_size, edge_ratio;
};

__device__ __forceinline__ void project(const double *P, double x, double y, double z, double &u, double &v, double &d)
{
    const double hx = x * P[0] + y * P[1] + z * P[2] + P[3];
    const double hy = x * P[4] + y * P[5] + z * P[6] + P[7];
    d = x * P[8] + y * P[9] + z * P[10] + P[11];
    u = hx / d;
    v = hy / d;
}

__device__ double gaussian_radius(double height, double width)
{
    const double mo = 0.7;
    const double b1 = height + width, c1 = width * height * (1 - mo) / (1 + mo);
    const double r1 = (b1 + sqrt(b1 * b1 - 4 * c1)) / 2;
    const double b2 = 2 * (height + width), c2 = (1 - mo) * width * height;
    const double r2 = (b2 + sqrt(b2 * b2 - 16 * c2)) / 2;
    const double a3 = 4 * mo, b3 = -2 * mo * (height + width), c3 = (mo - 1) * width * height;
    const double r3 = (b3 + sqrt(b3 * b3 - 4 * a3 * c3)) / 2;
    return fmin(r1, fmin(r2, r3));
}

// objs row (16 doubles): trunc, occ, box x1 y1 x2 y2 (float32 values), h, w, l, t x y z (float32 values), ry, alpha, find_pcl, class id
__global__ __launch_bounds__(TG_THREADS) void target_encode_objects(const double *__restrict__ objs, const double *__restrict__ kpts3d,
                                                                    const double *__restrict__ Pm, const int32_t *__restrict__ img_size,
                                                                    const int32_t *__restrict__ n_obj, TgCfg c, TgOut o)
{
    const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int K = c.n_extra + 10;
    const size_t slot = (size_t)b * c.M + i;
    if (i >= n_obj[b]) return;
    __shared__ double P[12];
    __shared__ double su[TG_MAXK], sv[TG_MAXK], sz[TG_MAXK];     // image coordinates (before padding) and camera depth per point
    __shared__ double c3[10][3];                                  // 8 corners + 2 face centres, camera frame
    __shared__ int s_keep, s_tc[2], s_rx, s_ry, s_cls;
    __shared__ float s_vis[10];
    if (tid < 12) P[tid] = Pm[(size_t)b * 12 + tid];
    const double *ob = objs + slot * 16;
    const double h = ob[6], w = ob[7], l = ob[8], ry = ob[12];
    const double tx = ob[9], ty = ob[10], tz = ob[11];
    const double cr = cos(ry), sr = sin(ry);
    const int img_w = img_size[2 * b], img_h = img_size[2 * b + 1];
    const int pad_x = (c.in_w - img_w) / 2, pad_y = (c.in_h - img_h) / 2;
    const int fw = c.in_w / c.down, fh = c.in_h / c.down;
    __syncthreads();

    // ---- points: 0..n_extra-1 extra key points (object frame, already shifted by -h/2), then the 8 corners, then 2 centres
    if (tid < 8) {
        const double xs[8] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2};
        const double ys[8] = {0, 0, 0, 0, -h, -h, -h, -h};
        const double zs[8] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2};
        c3[tid][0] = cr * xs[tid] + sr * zs[tid] + tx;
        c3[tid][1] = ys[tid] + ty;
        c3[tid][2] = -sr * xs[tid] + cr * zs[tid] + tz;
    }
    __syncthreads();
    if (tid < 2)
        for (int k = 0; k < 3; ++k) c3[8 + tid][k] = (c3[4 * tid][k] + c3[4 * tid + 1][k] + c3[4 * tid + 2][k] + c3[4 * tid + 3][k]) / 4;
    __syncthreads();
    if (tid < K) {
        double x, y, z;
        if (tid < c.n_extra) {
            const double *e = kpts3d + (slot * c.n_extra + tid) * 3;
            x = cr * e[0] + sr * e[2] + tx;
            y = e[1] + ty;
            z = -sr * e[0] + cr * e[2] + tz;
        } else {
            x = c3[tid - c.n_extra][0]; y = c3[tid - c.n_extra][1]; z = c3[tid - c.n_extra][2];
        }
        double u, v, d;
        project(P, x, y, z, u, v, d);
        su[tid] = u; sv[tid] = v; sz[tid] = z;
    }
    __syncthreads();

    // ---- scalar decisions (kitti.py:398-516)
    if (tid == 0) {
        s_keep = 0;
        const float locs_y = (float)((double)(float)ty - h / 2);        // float32 array arithmetic of the reference (:410-411)
        const double lx = (float)tx, ly = locs_y, lz = (float)tz;
        bool keep = lz > 0;
        const int cb = c.n_extra;
        double pb[4] = {su[cb], sv[cb], su[cb], sv[cb]};
        for (int k = 1; k < 8; ++k) {
            pb[0] = fmin(pb[0], su[cb + k]); pb[1] = fmin(pb[1], sv[cb + k]);
            pb[2] = fmax(pb[2], su[cb + k]); pb[3] = fmax(pb[3], sv[cb + k]);
        }
        const bool proj_box = pb[0] >= 0 && pb[1] >= 0 && pb[2] <= img_w - 1 && pb[3] <= img_h - 1;
        double b2[4];
        if (proj_box) {
            for (int k = 0; k < 4; ++k) b2[k] = pb[k];
            if (ob[0] >= c.filter_trunc && fmin(b2[2] - b2[0], b2[3] - b2[1]) <= c.filter_size) keep = false;
        } else {
            const float f0 = (float)ob[2], f1 = (float)ob[3], f2 = (float)ob[4], f3 = (float)ob[5];
            if (ob[0] >= c.filter_trunc && fminf(f2 - f0, f3 - f1) <= (float)c.filter_size) keep = false;
            b2[0] = f0; b2[1] = f1; b2[2] = f2; b2[3] = f3;
        }
        double pcu, pcv, pcd;
        project(P, lx, ly, lz, pcu, pcv, pcd);
        const bool inside = pcu >= 0 && pcu <= img_w - 1 && pcv >= 0 && pcv <= img_h - 1;
        double tpu = pcu, tpv = pcv;
        int approx = 0;
        if (keep && !inside) {                                           // approx_proj_center, kitti_utils.py:1040-1078
            approx = 1;
            double sx_, sy_;
            if (proj_box) { sx_ = (b2[0] + b2[2]) / 2; sy_ = (b2[1] + b2[3]) / 2; }
            else { sx_ = ((float)b2[0] + (float)b2[2]) / 2.f; sy_ = ((float)b2[1] + (float)b2[3]) / 2.f; }
            if (!(sx_ >= 0 && sy_ >= 0 && sx_ <= img_w - 1 && sy_ <= img_h - 1)) keep = false;   // the reference fails here (None)
            const double a = (sy_ - pcv) / (sx_ - pcu), bb = pcv - a * pcu;
            double best = 1e300;
            auto cand = [&](double x, double y) {
                const double d2 = (x - pcu) * (x - pcu) + (y - pcv) * (y - pcv);
                if (d2 < best) { best = d2; tpu = x; tpv = y; }
            };
            const double left_y = bb, right_y = (img_w - 1) * a + bb, top_x = -bb / a, bottom_x = (img_h - 1 - bb) / a;
            if (left_y >= 0 && left_y <= img_h - 1) cand(0, left_y);
            if (right_y >= 0 && right_y <= img_h - 1) cand(img_w - 1, right_y);
            if (top_x >= 0 && top_x <= img_w - 1) cand(top_x, 0);
            if (bottom_x >= 0 && bottom_x <= img_w - 1) cand(bottom_x, img_h - 1);
        }
        // feature-map scale (:489-499); the float32 box keeps float32 arithmetic
        const double tcx_f = (tpu + pad_x) / c.down, tcy_f = (tpv + pad_y) / c.down;
        const double pcx = (pcu + pad_x) / c.down, pcy = (pcv + pad_y) / c.down;
        if (proj_box) {
            b2[0] = (b2[0] + pad_x) / c.down; b2[1] = (b2[1] + pad_y) / c.down;
            b2[2] = (b2[2] + pad_x) / c.down; b2[3] = (b2[3] + pad_y) / c.down;
        } else {
            b2[0] = ((float)b2[0] + (float)pad_x) / (float)c.down; b2[1] = ((float)b2[1] + (float)pad_y) / (float)c.down;
            b2[2] = ((float)b2[2] + (float)pad_x) / (float)c.down; b2[3] = ((float)b2[3] + (float)pad_y) / (float)c.down;
        }
        const int x_min = (pad_x + c.down - 1) / c.down, y_min = (pad_y + c.down - 1) / c.down;
        const int x_max = (pad_x + img_w - 1) / c.down, y_max = (pad_y + img_h - 1) / c.down;
        long long tcx = llrint(tcx_f), tcy = llrint(tcy_f);            // np.round: half to even
        tcx = tcx < x_min ? x_min : (tcx > x_max ? x_max : tcx);
        tcy = tcy < y_min ? y_min : (tcy > y_max ? y_max : tcy);
        const bool pred_2d = tcx >= b2[0] && tcy >= b2[1] && tcx <= b2[2] && tcy <= b2[3];
        const double bw_ = b2[2] - b2[0], bh_ = b2[3] - b2[1];
        keep = keep && bw_ > 0 && bh_ > 0 && tcx >= 0 && tcx <= fw - 1 && tcy >= 0 && tcy <= fh - 1;
        if (keep) {
            int rx, ryy;
            if (approx) {                                                // :521-528 (ADJUST_BOUNDARY_HEATMAP)
                const double ew = fmin((double)tcx - b2[0], b2[2] - (double)tcx), eh = fmin((double)tcy - b2[1], b2[3] - (double)tcy);
                rx = (int)(ew * c.edge_ratio); ryy = (int)(eh * c.edge_ratio);
                rx = rx < 0 ? 0 : rx; ryy = ryy < 0 ? 0 : ryy;
            } else {
                rx = ryy = (int)gaussian_radius(bh_, bw_);
                if (rx < 0) rx = ryy = 0;
            }
            const int cls = (int)ob[15];
            s_keep = 1; s_tc[0] = (int)tcx; s_tc[1] = (int)tcy; s_rx = rx; s_ry = ryy; s_cls = cls;
            o.cls_ids[slot] = cls;
            o.target_centers[slot * 2] = (int)tcx; o.target_centers[slot * 2 + 1] = (int)tcy;
            o.offset_3d[slot * 2] = (float)(pcx - (double)tcx); o.offset_3d[slot * 2 + 1] = (float)(pcy - (double)tcy);
            for (int k = 0; k < 4; ++k) {
                o.gt_bboxes[slot * 4 + k] = (float)ob[2 + k];
                o.bboxes[slot * 4 + k] = pred_2d ? (float)b2[k] : 0.f;
            }
            for (int k = 0; k < 12; ++k) o.calib_p[slot * 12 + k] = (float)P[k];
            o.find_pcl[slot] = ob[14] != 0.0;
            o.dimensions[slot * 3] = (float)l; o.dimensions[slot * 3 + 1] = (float)h; o.dimensions[slot * 3 + 2] = (float)w;
            o.locations[slot * 3] = (float)lx; o.locations[slot * 3 + 1] = locs_y; o.locations[slot * 3 + 2] = (float)lz;
            o.rotys[slot] = (float)ry; o.alphas[slot] = (float)ob[13];
            o.reg_mask[slot] = 1; o.reg_weight[slot] = 1.f; o.trunc_mask[slot] = (uint8_t)approx;
            o.occlusions[slot] = ob[1]; o.truncations[slot] = ob[0];
            // multi-bin orientation (:225-244)
            const double PI_ = 3.14159265358979323846;
            const double centers[4] = {0, PI_ / 2, PI_, -PI_ / 2};
            const double bin_size = 2 * PI_ / c.num_bin, range_size = bin_size / 2 + bin_size * (1.0 / 6);
            for (int k = 0; k < c.num_bin; ++k) {
                double off = ob[13] - centers[k];
                if (off > PI_) off -= 2 * PI_;
                if (off < -PI_) off += 2 * PI_;
                const bool in = fabs(off) < range_size;
                o.orientations[slot * 2 * c.num_bin + k] = in ? 1.f : 0.f;
                o.orientations[slot * 2 * c.num_bin + c.num_bin + k] = in ? (float)off : 0.f;
            }
        }
    }
    __syncthreads();
    if (!s_keep) return;
    const int tcx = s_tc[0], tcy = s_tc[1];

    // ---- per-point rows: visibility, local coordinates (:463-486, :544-554)
    bool vis = false;
    if (tid < K) vis = su[tid] >= 0 && su[tid] <= img_w - 1 && sv[tid] >= 0 && sv[tid] <= img_h - 1 && sz[tid] > 0;
    if (tid >= c.n_extra && tid < K) s_vis[tid - c.n_extra] = vis ? 1.f : 0.f;
    __syncthreads();
    if (tid < 10) {                                       // KEYPOINT_VISIBLE_MODIFY: corner k and k+4 share, the two centres share
        const int k = tid;
        float m;
        if (k < 8) m = (s_vis[k & 3] != 0.f || s_vis[(k & 3) + 4] != 0.f) ? 1.f : 0.f;
        else m = (s_vis[8] != 0.f || s_vis[9] != 0.f) ? 1.f : 0.f;
        const int p = c.n_extra + k;
        const float kx = (float)((su[p] + pad_x) / c.down - tcx), ky = (float)((sv[p] + pad_y) / c.down - tcy);
        float *kp = o.keypoints + (slot * 10 + k) * 3;
        kp[0] = kx; kp[1] = ky; kp[2] = m;
        float *e2 = o.ek2d + (slot * K + p) * 3;
        e2[0] = kx; e2[1] = ky; e2[2] = m;
        o.ekdm[slot * K + p] = m;
        // raw box points in the object frame (kitti_utils.py:147)
        const double xs[10] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2, 0, 0};
        const double ys[10] = {0, 0, 0, 0, -h, -h, -h, -h, 0, -h};
        const double zs[10] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2, 0, 0};
        float *e3 = o.ek3d + (slot * K + p) * 3;
        e3[0] = (float)xs[k]; e3[1] = (float)ys[k]; e3[2] = (float)zs[k];
    }
    if (tid < c.n_extra) {
        float *e2 = o.ek2d + (slot * K + tid) * 3;
        e2[0] = (float)((su[tid] + pad_x) / c.down - tcx);
        e2[1] = (float)((sv[tid] + pad_y) / c.down - tcy);
        e2[2] = vis ? 1.f : 0.f;
        o.ekdm[slot * K + tid] = vis ? 1.f : 0.f;
        const double *e = kpts3d + (slot * c.n_extra + tid) * 3;
        float *e3 = o.ek3d + (slot * K + tid) * 3;
        e3[0] = (float)e[0]; e3[1] = (float)e[1]; e3[2] = (float)e[2];
    }
    __syncthreads();
    if (tid == 0) {                                       // depth validity of the three key-point groups, after the modification
        auto mv = [&](int k) { return k < 8 ? (s_vis[k & 3] != 0.f || s_vis[(k & 3) + 4] != 0.f) : (s_vis[8] != 0.f || s_vis[9] != 0.f); };
        o.kdm[slot * 3 + 0] = (mv(8) && mv(9)) ? 1.f : 0.f;
        o.kdm[slot * 3 + 1] = (mv(0) && mv(2) && mv(4) && mv(6)) ? 1.f : 0.f;
        o.kdm[slot * 3 + 2] = (mv(1) && mv(3) && mv(5) && mv(7)) ? 1.f : 0.f;
    }

    // ---- heat map: max with the (elliptic) Gaussian patch (heatmap_coder.py:59-68, :84-124)
    const int rx = s_rx, ryy = s_ry;
    const double sgx = (2 * rx + 1) / 6.0, sgy = (2 * ryy + 1) / 6.0;
    const int left = min(tcx, rx), right = min(fw - tcx, rx + 1), top = min(tcy, ryy), bottom = min(fh - tcy, ryy + 1);
    const int pw = left + right, ph = top + bottom;
    unsigned *hm = reinterpret_cast<unsigned *>(o.hm + ((size_t)b * c.n_cls + s_cls) * fh * fw);
    for (int e = tid; e < pw * ph; e += TG_THREADS) {
        const int yy = e / pw, xx = e - yy * pw;
        const double dx = (double)(xx - left), dy = (double)(yy - top);
        double g = exp(-(dx * dx) / (2 * sgx * sgx) - (dy * dy) / (2 * sgy * sgy));
        if (g < 2.220446049250313e-16) g = 0.0;                         // h[h < eps * h.max()] = 0, h.max() = 1 at the centre
        const float gf = (float)g;
        atomicMax(hm + (size_t)(tcy - top + yy) * fw + (tcx - left + xx), __float_as_uint(gf));
    }
}

// border walk of the un-padded image area on the stride-`down` map (kitti.py:165-223): left column down, bottom row right,
// right column up, top row left; one workgroup per image
__global__ void target_edge_indices(const int32_t *__restrict__ img_size, TgCfg c, int64_t *__restrict__ pad_size,
                                    int64_t *__restrict__ edge_indices, int64_t *__restrict__ edge_len, int max_edge)
{
    const int b = blockIdx.x;
    const int img_w = img_size[2 * b], img_h = img_size[2 * b + 1];
    const int pad_x = (c.in_w - img_w) / 2, pad_y = (c.in_h - img_h) / 2;
    const int x_min = (pad_x + c.down - 1) / c.down, y_min = (pad_y + c.down - 1) / c.down;
    const int x_max = (pad_x + img_w - 1) / c.down, y_max = (pad_y + img_h - 1) / c.down;
    const int n_left = y_max - y_min, n_bottom = x_max - x_min, n_right = y_max - y_min, n_top = x_max - x_min + 1;
    const int n = n_left + n_bottom + n_right + n_top;
    int64_t *e = edge_indices + (size_t)b * max_edge * 2;
    for (int k = threadIdx.x; k < max_edge; k += blockDim.x) {
        int x = 0, y = 0;
        if (k < n_left) { x = x_min; y = y_min + k; }
        else if (k < n_left + n_bottom) { x = x_min + (k - n_left); y = y_max; }
        else if (k < n_left + n_bottom + n_right) { x = x_max; y = y_max - (k - n_left - n_bottom); }
        else if (k < n) { x = x_max - (k - n_left - n_bottom - n_right); y = y_min; }
        e[2 * k] = x; e[2 * k + 1] = y;
    }
    if (threadIdx.x == 0) {
        pad_size[2 * b] = pad_x; pad_size[2 * b + 1] = pad_y;
        edge_len[b] = n - 1;
    }
}

}  // namespace

extern "C" {

int dcd_encode_targets(void *stream_, const double *objs, const double *kpts3d, const double *P, const int32_t *img_size,
                       const int32_t *n_obj, int B, int M, int n_extra, int in_w, int in_h, int down_ratio, double filter_trunc,
                       double filter_size, double edge_heatmap_ratio, int num_bin, int n_classes, void *const *outputs, int n_outputs)
{
    hipStream_t stream = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (!objs || !kpts3d || !P || !img_size || !n_obj || !outputs || n_outputs != 26) return DCD_ERR_BAD_ARG;
    if (B <= 0 || M <= 0 || n_extra < 0 || n_extra + 10 > TG_MAXK || in_w <= 0 || in_h <= 0 || down_ratio <= 0 || num_bin != 4 || n_classes <= 0)
        return DCD_ERR_BAD_ARG;
    for (int k = 0; k < n_outputs; ++k)
        if (!outputs[k]) return DCD_ERR_BAD_ARG;
    TgCfg c = {B, M, n_extra, in_w, in_h, down_ratio, num_bin, n_classes, filter_trunc, filter_size, edge_heatmap_ratio};
    TgOut o;
    int k = 0;
    o.hm = (float *)outputs[k++]; o.cls_ids = (int32_t *)outputs[k++]; o.target_centers = (int32_t *)outputs[k++];
    o.gt_bboxes = (float *)outputs[k++]; o.bboxes = (float *)outputs[k++]; o.keypoints = (float *)outputs[k++];
    o.kdm = (float *)outputs[k++]; o.ek2d = (float *)outputs[k++]; o.ek3d = (float *)outputs[k++]; o.ekdm = (float *)outputs[k++];
    o.calib_p = (float *)outputs[k++]; o.find_pcl = (uint8_t *)outputs[k++]; o.dimensions = (float *)outputs[k++];
    o.locations = (float *)outputs[k++]; o.rotys = (float *)outputs[k++]; o.alphas = (float *)outputs[k++];
    o.offset_3d = (float *)outputs[k++]; o.occlusions = (double *)outputs[k++]; o.truncations = (double *)outputs[k++];
    o.orientations = (float *)outputs[k++]; o.reg_mask = (uint8_t *)outputs[k++]; o.trunc_mask = (uint8_t *)outputs[k++];
    o.reg_weight = (float *)outputs[k++];
    int64_t *pad_size = (int64_t *)outputs[k++], *edge_indices = (int64_t *)outputs[k++], *edge_len = (int64_t *)outputs[k++];
    const int max_edge = (in_w / down_ratio + in_h / down_ratio) * 2;
    hipLaunchKernelGGL(target_encode_objects, dim3(M, B), dim3(TG_THREADS), 0, stream, objs, kpts3d, P, img_size, n_obj, c, o);
    hipLaunchKernelGGL(target_edge_indices, dim3(B), dim3(256), 0, stream, img_size, c, pad_size, edge_indices, edge_len, max_edge);
    return hipGetLastError() == hipSuccess ? DCD_OK : DCD_ERR_LAUNCH;
}

}  // extern "C"
