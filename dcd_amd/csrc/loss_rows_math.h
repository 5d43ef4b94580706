// Arithmetic of one object slot of the training loss (include/dcd_hip.h, dcd_loss_rows_*).
//
// Written once for two compilations: loss_rows.hip instantiates it with a 64-lane wave (one wave per slot), and
// tests/host_rows.cpp instantiates it with a one-lane "wave" on the host so that the formulas and their hand-written
// derivatives can be checked against autograd without a GPU.  The host build is test infrastructure; the library has
// no host path.
//
// Follows DGDE/model/head/detector_loss.py:405-583 (terms) and :217-403 (decodes), DGDE/model/anno_encoder.py (cited per
// block), DGDE/model/layers/iou_loss.py:12-49 (GIoU), DGDE/model/head/depth_losses.py:50-67 (depth-weighted L1).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/dcd_hip.h"

#ifdef __HIPCC__
#define LR_HD __host__ __device__ __forceinline__
#else
#define LR_HD inline
#endif

enum {
    LR_OV, LR_GIOU, LR_IOU, LR_M2, LR_DEPTH_LOG, LR_DEPTH, LR_TRUNC, LR_OFF, LR_ORI, LR_DIMS, LR_IOU3D, LR_CORNER, LR_KP,
    LR_L2D, LR_M2D, LR_L3D, LR_M3D, LR_VALID_L, LR_INVALID_L, LR_N_VALID, LR_N_INVALID, LR_MAE, LR_KD_LOG, LR_KD_V, LR_KD_I
};
static_assert(LR_KD_I + 1 == DCD_LOSS_ROWS_NCOL, "column list and DCD_LOSS_ROWS_NCOL disagree");

#define LR_PI 3.14159265358979323846f
#define LR_NKP 10        // box keypoints: 8 corners + top / bottom centre
#define LR_NBIN 4        // orientation bins

LR_HD float lr_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
LR_HD float lr_abs(float x) { return x < 0.f ? -x : x; }
LR_HD float lr_min(float a, float b) { return a < b ? a : b; }
LR_HD float lr_max(float a, float b) { return a > b ? a : b; }
// gradient of min(a, b) / max(a, b) w.r.t. a (ties split evenly, as torch.min / torch.max do)
LR_HD float lr_dmin(float a, float b) { return a < b ? 1.f : (a > b ? 0.f : 0.5f); }
LR_HD float lr_dmax(float a, float b) { return a > b ? 1.f : (a < b ? 0.f : 0.5f); }
// gradient factor of clamp(x, lo, hi)
LR_HD float lr_dclamp(float x, float lo, float hi) { return (x >= lo && x <= hi) ? 1.f : 0.f; }
LR_HD float lr_clamp(float x, float lo, float hi) { return lr_min(lr_max(x, lo), hi); }

// Which annotated object a slot stands for (detector_loss.py:217-230 compacts the list; here every slot is evaluated and
// the empty ones read the first annotated object), its image, and f_u of the calibration the reference picks for the
// keypoint depths: it indexes the calibrations by the RANK of the image among the images that own objects
// (anno_encoder.py:206-207).
struct LrSlot {
    int r, b;
    bool ov;
    float fu_rank;
};

template <class W>
LR_HD LrSlot lr_slot(const dcd_loss_rows_args &a, int s, W w)
{
    const int BM = a.B * a.M;
    int first = BM;
    for (int i = w.lane(); i < BM; i += w.lanes())
        if (a.reg_mask[i] && i < first) first = i;
    first = w.min_int(first);
    if (first == BM) first = 0;
    LrSlot o;
    o.ov = a.reg_mask[s] != 0;
    o.r = o.ov ? s : first;
    o.b = o.r / a.M;
    int rank = 0;
    for (int bb = 0; bb < o.b; ++bb) {
        bool any = false;
        for (int i = w.lane(); i < a.M; i += w.lanes()) any = any || a.reg_mask[bb * a.M + i] != 0;
        rank += w.any(any) ? 1 : 0;
    }
    o.fu_rank = a.calib[rank * 6 + 2];
    return o;
}

// Inputs of the edge solver for slot s: dense keypoints in image pixels (decode_kpts_2d_img, anno_encoder.py:392-393:
// (kpts + centre + offset) * 4 - pad), the 3-D keypoints, yaw, projection matrix and the keypoint mask
// (detector_loss.py:356-381).
template <class W>
LR_HD void lr_prepare_row(const dcd_loss_rows_args &a, int s, W w)
{
    const LrSlot sl = lr_slot(a, s, w);
    const int r = sl.r, K = a.K, BM = a.B * a.M;
    const bool found = a.find_pcl[r] != 0 && sl.ov;
    const float bx = (float)a.centers[r * 2 + 0] + a.offset_3D[r * 2 + 0];
    const float by = (float)a.centers[r * 2 + 1] + a.offset_3D[r * 2 + 1];
    const float padx = (float)a.pad_size[sl.b * 2 + 0], pady = (float)a.pad_size[sl.b * 2 + 1];
    const float *p = a.pois + (size_t)r * a.C;
    for (int k = w.lane(); k < K; k += w.lanes()) {
        const float *t2 = a.kpts2d + ((size_t)r * K + k) * 3;
        const float *t3 = a.kpts3d + ((size_t)r * K + k) * 3;
        const size_t o2 = ((size_t)s * K + k) * 2, o3 = ((size_t)s * K + k) * 3;
        a.kps_pred[o2 + 0] = (p[a.ch_kpts2d + 2 * k + 0] + bx) * 4.f - padx;
        a.kps_pred[o2 + 1] = (p[a.ch_kpts2d + 2 * k + 1] + by) * 4.f - pady;
        a.kps_tgt[o2 + 0] = (t2[0] + bx) * 4.f - padx;
        a.kps_tgt[o2 + 1] = (t2[1] + by) * 4.f - pady;
        for (int i = 0; i < 3; ++i) {
            a.kps3d_pred[o3 + i] = p[a.ch_kpts3d + 3 * k + i];
            a.kps3d_tgt[o3 + i] = t3[i];
        }
        // yaw, projection and mask are stored twice, (2, B*M, ...): with kps_pred / kps_tgt and kps3d_pred / kps3d_tgt
        // as the halves of one buffer each, ONE solver call over 2*B*M rows serves predictions and targets
        const uint8_t m = (t2[2] != 0.f && found) ? 1 : 0;
        a.kmask[(size_t)s * K + k] = m;
        a.kmask[((size_t)BM + s) * K + k] = m;
    }
    if (w.lane() == 0) {
        a.rot[s] = a.rot[BM + s] = a.rotys[r];
        for (int i = 0; i < 12; ++i)
            a.P_rows[(size_t)s * 12 + i] = a.P_rows[((size_t)BM + s) * 12 + i] = a.calib_P[(size_t)r * 12 + i];
    }
}

// 8 corners of a box (encode_box3d, anno_encoder.py:93-128): x uses l/2, y uses h/2, z uses w/2 with the sign table of
// :119-123; R = rotation about the camera y axis.
LR_HD void lr_corner_signs(int k, float &sx, float &sy, float &sz)
{
    sx = (k == 2 || k == 3 || k == 6 || k == 7) ? 1.f : -1.f;
    sy = k < 4 ? 1.f : -1.f;
    sz = (k == 1 || k == 2 || k == 5 || k == 6) ? 1.f : -1.f;
}

LR_HD float lr_wrap(float a)
{
    if (a > LR_PI) a -= 2.f * LR_PI;
    if (a < -LR_PI) a += 2.f * LR_PI;
    return a;
}

// One slot.  BWD = false: writes the slot's column values, both corner sets.  BWD = true: recomputes the forward values it
// needs and writes d(sum_c grad_sums[c] * column c) / d(pois row) and / d(pair depths).
template <bool BWD, class W>
LR_HD void lr_row(const dcd_loss_rows_args &a, int s, W w)
{
    const LrSlot sl = lr_slot(a, s, w);
    const int r = sl.r, b = sl.b, K = a.K, NP = a.NP, BM = a.B * a.M;
    const bool ov = sl.ov;
    const bool lane0 = w.lane() == 0;
    float *gp = BWD ? a.grad_pois + (size_t)s * a.C : nullptr;
    float *gpair = BWD ? a.grad_pair + (size_t)s * NP : nullptr;
    if (BWD && !ov) {      // an empty slot contributes to no sum
        for (int c = w.lane(); c < a.C; c += w.lanes()) gp[c] = 0.f;
        for (int j = w.lane(); j < NP; j += w.lanes()) gpair[j] = 0.f;
        return;
    }
    const float *p = a.pois + (size_t)r * a.C;
    const float *g = a.grad_sums;
    float col[DCD_LOSS_ROWS_NCOL];
    for (int c = 0; c < DCD_LOSS_ROWS_NCOL; ++c) col[c] = 0.f;
    col[LR_OV] = ov ? 1.f : 0.f;

    const float cx = (float)a.centers[r * 2 + 0], cy = (float)a.centers[r * 2 + 1];
    const float td = a.locations[r * 3 + 2];                              // target depth
    const float padx = (float)a.pad_size[b * 2 + 0], pady = (float)a.pad_size[b * 2 + 1];
    const float *cal = a.calib + b * 6;
    const float c_u = cal[0], c_v = cal[1], f_u = cal[2], f_v = cal[3], b_x = cal[4], b_y = cal[5];

    // ---- 2-D box: GIoU on (l,t,r,b) distances of the objects with a non-degenerate box (:415-421; iou_loss.py:12-49)
    {
        const float x1 = a.bboxes[r * 4 + 0], y1 = a.bboxes[r * 4 + 1], x2 = a.bboxes[r * 4 + 2], y2 = a.bboxes[r * 4 + 3];
        const bool m2 = (y2 - y1 > 0.f) && (x2 - x1 > 0.f) && ov;
        const float tl = m2 ? cx - x1 : 1.f, tt = m2 ? cy - y1 : 1.f, tr = m2 ? x2 - cx : 1.f, tb = m2 ? y2 - cy : 1.f;
        const float rl = p[a.ch_box2d + 0], rt = p[a.ch_box2d + 1], rr = p[a.ch_box2d + 2], rb = p[a.ch_box2d + 3];
        const float pl = lr_max(rl, 0.f), pt = lr_max(rt, 0.f), pr = lr_max(rr, 0.f), pb = lr_max(rb, 0.f);
        const float ta = (tl + tr) * (tt + tb), pa = (pl + pr) * (pt + pb);
        const float wi = lr_min(pl, tl) + lr_min(pr, tr), gwi = lr_max(pl, tl) + lr_max(pr, tr);
        const float hi = lr_min(pb, tb) + lr_min(pt, tt), ghi = lr_max(pb, tb) + lr_max(pt, tt);
        const float ac = gwi * ghi + 1e-7f, ai = wi * hi, au = ta + pa - ai;
        const float iou = (ai + 1.f) / (au + 1.f);
        const float giou = iou - (ac - au) / ac;
        col[LR_GIOU] = m2 ? 1.f - giou : 0.f;
        col[LR_IOU] = m2 ? iou : 0.f;
        col[LR_M2] = m2 ? 1.f : 0.f;
        if (BWD && lane0) {
            const float G = m2 ? g[LR_GIOU] : 0.f;
            // loss = 2 - iou - au / ac
            const float g_ai = -1.f / (au + 1.f);
            const float g_au = (ai + 1.f) / ((au + 1.f) * (au + 1.f)) - 1.f / ac;
            const float g_ac = au / (ac * ac);
            const float g_pa = g_au, g_ai_t = g_ai - g_au;        // au = ta + pa - ai
            const float g_wi = g_ai_t * hi, g_hi = g_ai_t * wi, g_gwi = g_ac * ghi, g_ghi = g_ac * gwi;
            const float gl = g_pa * (pt + pb) + g_wi * lr_dmin(pl, tl) + g_gwi * lr_dmax(pl, tl);
            const float gr = g_pa * (pt + pb) + g_wi * lr_dmin(pr, tr) + g_gwi * lr_dmax(pr, tr);
            const float gt = g_pa * (pl + pr) + g_hi * lr_dmin(pt, tt) + g_ghi * lr_dmax(pt, tt);
            const float gb = g_pa * (pl + pr) + g_hi * lr_dmin(pb, tb) + g_ghi * lr_dmax(pb, tb);
            gp[a.ch_box2d + 0] = rl > 0.f ? G * gl : 0.f;          // through the relu
            gp[a.ch_box2d + 1] = rt > 0.f ? G * gt : 0.f;
            gp[a.ch_box2d + 2] = rr > 0.f ? G * gr : 0.f;
            gp[a.ch_box2d + 3] = rb > 0.f ? G * gb : 0.f;
        }
    }

    // ---- direct depth (inv_sigmoid decode, anno_encoder.py:130-145) with its uncertainty (:423-429)
    {
        const float x = p[a.ch_depth];
        const float sg = 1.f / (1.f + expf(-x));
        const float d_raw = 1.f / sg - 1.f;
        const float d = lr_clamp(d_raw, a.depth_lo, a.depth_hi);
        const float dl = a.depth_weight * lr_abs(d - td);
        const float u_raw = p[a.ch_depth_unc];
        const float u = lr_clamp(u_raw, a.unc_lo, a.unc_hi);
        const float eu = expf(-u);
        col[LR_DEPTH_LOG] = ov ? dl : 0.f;
        col[LR_DEPTH] = ov ? dl * eu + u * a.depth_weight : 0.f;
        if (BWD && lane0) {
            const float G = g[LR_DEPTH];
            const float g_d = G * a.depth_weight * lr_sign(d - td) * eu * lr_dclamp(d_raw, a.depth_lo, a.depth_hi);
            gp[a.ch_depth] = g_d * (-(1.f - sg) / sg);
            gp[a.ch_depth_unc] = G * (a.depth_weight - dl * eu) * lr_dclamp(u_raw, a.unc_lo, a.unc_hi);
        }
    }

    // ---- projected-centre offset; truncated objects use the log form (:431-440)
    const float o0 = p[a.ch_offset + 0], o1 = p[a.ch_offset + 1];
    const float t0 = a.offset_3D[r * 2 + 0], t1 = a.offset_3D[r * 2 + 1];
    float g_o0 = 0.f, g_o1 = 0.f;
    {
        const float off_l = lr_abs(o0 - t0) + lr_abs(o1 - t1);
        const bool tv = a.trunc_mask[r] != 0 && ov;
        const float t_l = a.trunc_log ? logf(1.f + off_l) : off_l;
        col[LR_TRUNC] = tv ? t_l : 0.f;
        col[LR_OFF] = (ov && !tv) ? off_l : 0.f;
        if (BWD) {
            const float coef = tv ? g[LR_TRUNC] * (a.trunc_log ? 1.f / (1.f + off_l) : 1.f) : g[LR_OFF];
            g_o0 = coef * lr_sign(o0 - t0);
            g_o1 = coef * lr_sign(o1 - t1);
        }
    }

    // ---- multi-bin orientation (Real_MultiBin_loss, detector_loss.py:644-666): per-bin 2-way cross entropy / 4 plus L1 on
    // the normalised (sin, cos) offsets of the bins that contain the angle
    float q1[LR_NBIN];                   // softmax probability of "angle in this bin" (also the decode's bin score)
    float g_oc[2 * LR_NBIN], g_oo[2 * LR_NBIN];
    {
        const bool mo = a.ori_mask[r] != 0 && ov;
        const float G = (BWD && mo) ? g[LR_ORI] : 0.f;
        float per_row = 0.f;
        for (int i = 0; i < LR_NBIN; ++i) {
            const float l0 = p[a.ch_ori_cls + 2 * i], l1 = p[a.ch_ori_cls + 2 * i + 1];
            const float mx = lr_max(l0, l1);
            const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
            const float lse = logf(e0 + e1);
            const float logp0 = l0 - mx - lse, logp1 = l1 - mx - lse;
            q1[i] = e1 / (e0 + e1);
            const float binf = a.orientations[r * 8 + i];
            const int bin = (int)binf;
            const float ce = -(bin ? logp1 : logp0);
            const float v0 = p[a.ch_ori_off + 2 * i], v1 = p[a.ch_ori_off + 2 * i + 1];
            const float nrm = sqrtf(v0 * v0 + v1 * v1);
            const float den = lr_max(nrm, 1e-12f);
            const float f0 = v0 / den, f1 = v1 / den;
            const float ang = a.orientations[r * 8 + LR_NBIN + i];
            const float ts = sinf(ang), tc = cosf(ang);
            const bool in = binf == 1.f;
            per_row += ce * (1.0f / LR_NBIN) + (in ? lr_abs(f0 - ts) + lr_abs(f1 - tc) : 0.f);
            if (BWD) {
                const float q0 = e0 / (e0 + e1);
                g_oc[2 * i + 0] = G * (1.0f / LR_NBIN) * (q0 - (bin == 0 ? 1.f : 0.f));
                g_oc[2 * i + 1] = G * (1.0f / LR_NBIN) * (q1[i] - (bin == 1 ? 1.f : 0.f));
                const float gf0 = in ? G * lr_sign(f0 - ts) : 0.f, gf1 = in ? G * lr_sign(f1 - tc) : 0.f;
                if (nrm > 1e-12f) {
                    const float dot = f0 * gf0 + f1 * gf1;
                    g_oo[2 * i + 0] = (gf0 - f0 * dot) / den;
                    g_oo[2 * i + 1] = (gf1 - f1 * dot) / den;
                } else {
                    g_oo[2 * i + 0] = gf0 / den;
                    g_oo[2 * i + 1] = gf1 / den;
                }
            }
        }
        col[LR_ORI] = mo ? per_row : 0.f;
    }

    // ---- dimensions: exp(offset) * class mean (decode_dimension, anno_encoder.py:226-252), weighted L1 (:451-454)
    float pd[3], tdim[3], g_pd[3] = {0.f, 0.f, 0.f};
    {
        int cls = a.cls_ids[r];
        cls = cls < 0 ? 0 : (cls >= a.num_classes ? a.num_classes - 1 : cls);
        float rows = 0.f;
        for (int i = 0; i < 3; ++i) {
            pd[i] = expf(p[a.ch_dims + i]) * a.dim_mean[cls * 3 + i];
            tdim[i] = a.dimensions[r * 3 + i];
            rows += lr_abs(pd[i] - tdim[i]) * a.dim_weight[i];
            if (BWD) g_pd[i] = g[LR_DIMS] * lr_sign(pd[i] - tdim[i]) * a.dim_weight[i];
        }
        col[LR_DIMS] = ov ? rows : 0.f;
    }

    // ---- pair depths, first pass: their mean is the depth the corners are decoded at (CORNER_LOSS_DEPTH 'edges', :383-387);
    // the pair-depth term itself (:188-204)
    const bool found = a.find_pcl[r] != 0 && ov;
    const float *pdp = a.pair_depth + (size_t)s * NP, *pmk = a.pair_mask + (size_t)s * NP;
    float zsum = 0.f;
    {
        float vl = 0.f, il = 0.f, nv = 0.f, ni = 0.f, mae = 0.f;
        for (int j = w.lane(); j < NP; j += w.lanes()) {
            const float v = pdp[j];
            const bool pm = pmk[j] > 0.f;
            const float reg = lr_abs(v - td);
            zsum += v;
            if (found) {
                if (pm) { vl += reg; nv += 1.f; mae += reg / td; } else { il += reg; ni += 1.f; }
            }
        }
        zsum = w.sum(zsum);
        if (!BWD) {
            col[LR_VALID_L] = w.sum(vl);
            col[LR_INVALID_L] = w.sum(il);
            col[LR_N_VALID] = w.sum(nv);
            col[LR_N_INVALID] = w.sum(ni);
            col[LR_MAE] = w.sum(mae);
        }
    }
    const float z = zsum / (float)NP;

    // ---- corners of the predicted box: location from (centre + offset) at depth z (decode_location_flatten,
    // anno_encoder.py:147-161), yaw from the best bin plus the viewing ray (decode_axes_orientation, :254-304)
    float g_z = 0.f;
    {
        const float u = (cx + o0) * a.down_ratio - padx, v = (cy + o1) * a.down_ratio - pady;
        const float X = ((u - c_u) * z) / f_u + b_x, Y = ((v - c_v) * z) / f_v + b_y;
        int best = 0;
        for (int i = 1; i < LR_NBIN; ++i) if (q1[i] > q1[best]) best = i;
        const float of0 = p[a.ch_ori_off + 2 * best], of1 = p[a.ch_ori_off + 2 * best + 1];
        const float centre = best == 0 ? 0.f : (best == 1 ? 0.5f * LR_PI : (best == 2 ? LR_PI : -0.5f * LR_PI));
        const float roty = lr_wrap(atan2f(of0, of1) + centre + atan2f(X, z));
        const float cr = cosf(roty), sr = sinf(roty);
        // target box: annotated yaw and size, location decoded from the annotated offset and depth (:290-292)
        const float tu = (cx + t0) * a.down_ratio - padx, tv_ = (cy + t1) * a.down_ratio - pady;
        const float TX = ((tu - c_u) * td) / f_u + b_x, TY = ((tv_ - c_v) * td) / f_v + b_y;
        const float troty = a.rotys[r];
        const float tcr = cosf(troty), tsr = sinf(troty);
        float corner_l = 0.f, gX = 0.f, gY = 0.f, gZ = 0.f, g_roty = 0.f;
        const float G = BWD ? g[LR_CORNER] : 0.f;
        for (int k = 0; k < 8; ++k) {
            float sx, sy, sz;
            lr_corner_signs(k, sx, sy, sz);
            const float ox = 0.5f * pd[0] * sx, oy = 0.5f * pd[1] * sy, oz = 0.5f * pd[2] * sz;
            const float px = cr * ox + sr * oz + X, py = oy + Y, pz = -sr * ox + cr * oz + z;
            const float qx = 0.5f * tdim[0] * sx, qy = 0.5f * tdim[1] * sy, qz = 0.5f * tdim[2] * sz;
            const float tx = tcr * qx + tsr * qz + TX, ty = qy + TY, tz = -tsr * qx + tcr * qz + td;
            corner_l += lr_abs(px - tx) + lr_abs(py - ty) + lr_abs(pz - tz);
            if (!BWD && lane0) {
                float *cp = a.corners_pred + ((size_t)s * 8 + k) * 3, *ct = a.corners_tgt + ((size_t)s * 8 + k) * 3;
                cp[0] = px; cp[1] = py; cp[2] = pz;
                ct[0] = tx; ct[1] = ty; ct[2] = tz;
            }
            if (BWD) {
                const float hx = G * lr_sign(px - tx), hy = G * lr_sign(py - ty), hz = G * lr_sign(pz - tz);
                gX += hx; gY += hy; gZ += hz;
                g_pd[0] += 0.5f * sx * (cr * hx - sr * hz);
                g_pd[1] += 0.5f * sy * hy;
                g_pd[2] += 0.5f * sz * (sr * hx + cr * hz);
                g_roty += hx * (-sr * ox + cr * oz) + hz * (-cr * ox - sr * oz);
            }
        }
        col[LR_CORNER] = ov ? corner_l : 0.f;
        if (BWD) {
            // yaw = atan2(of0, of1) + centre + atan2(X, z); the wrap has slope 1
            const float den = X * X + z * z;
            gX += g_roty * z / den;
            gZ += -g_roty * X / den;
            const float den2 = of0 * of0 + of1 * of1;
            g_oo[2 * best + 0] += g_roty * of1 / den2;
            g_oo[2 * best + 1] += -g_roty * of0 / den2;
            g_o0 += gX * z / f_u * a.down_ratio;
            g_o1 += gY * z / f_v * a.down_ratio;
            g_z = gZ + gX * (u - c_u) / f_u + gY * (v - c_v) / f_v;
        }
    }

    // ---- box keypoints: masked L1 (:459-463) and the three depths from projected heights
    // (decode_depth_from_keypoints_batch, anno_encoder.py:193-224) with their uncertainties (:505-541)
    {
        float ky[LR_NKP], g_ky[LR_NKP];
        float kl = 0.f;
        const float Gk = BWD ? g[LR_KP] : 0.f;
        for (int k = 0; k < LR_NKP; ++k) {
            const float kx = p[a.ch_corner + 2 * k];
            ky[k] = p[a.ch_corner + 2 * k + 1];
            const float *t = a.keypoints + ((size_t)r * LR_NKP + k) * 3;
            kl += (lr_abs(kx - t[0]) + lr_abs(ky[k] - t[1])) * t[2];
            if (BWD) {
                if (lane0) gp[a.ch_corner + 2 * k] = Gk * t[2] * lr_sign(kx - t[0]);
                g_ky[k] = Gk * t[2] * lr_sign(ky[k] - t[1]);
            }
        }
        col[LR_KP] = ov ? kl : 0.f;

        const int ia[5] = {LR_NKP - 2, 0, 2, 1, 3}, ib[5] = {LR_NKP - 1, 4, 6, 5, 7};
        const float fh = sl.fu_rank * pd[1];
        float dep[5], den[5], h[5];
        for (int i = 0; i < 5; ++i) {
            h[i] = ky[ia[i]] - ky[ib[i]];
            den[i] = lr_max(h[i], 0.f) * a.down_ratio + a.kd_eps;
            dep[i] = fh / den[i];
        }
        const float kdr[3] = {dep[0], (dep[1] + dep[2]) / 2.f, (dep[3] + dep[4]) / 2.f};
        float g_kdr[3];
        for (int j = 0; j < 3; ++j) {
            const float kd = lr_clamp(kdr[j], a.depth_lo, a.depth_hi);
            const bool km = a.kp_depth_mask[r * 3 + j] != 0.f;
            const float vj = lr_abs(kd - td);
            const float cu_raw = p[a.ch_corner_unc + j];
            const float cu = lr_clamp(cu_raw, a.unc_lo, a.unc_hi);
            const float ecu = expf(-cu);
            if (ov) {
                if (km) { col[LR_KD_LOG] += vj; col[LR_KD_V] += vj * ecu + cu; } else { col[LR_KD_I] += vj * ecu; }
            }
            if (BWD) {
                g_kdr[j] = km ? g[LR_KD_V] * ecu * lr_sign(kd - td) * lr_dclamp(kdr[j], a.depth_lo, a.depth_hi) : 0.f;
                const float g_cu = km ? g[LR_KD_V] * (1.f - vj * ecu) : -g[LR_KD_I] * vj * ecu;
                if (lane0) gp[a.ch_corner_unc + j] = g_cu * lr_dclamp(cu_raw, a.unc_lo, a.unc_hi);
            }
        }
        if (BWD) {
            const float g_dep[5] = {g_kdr[0], 0.5f * g_kdr[1], 0.5f * g_kdr[1], 0.5f * g_kdr[2], 0.5f * g_kdr[2]};
            float g_fh = 0.f;
            for (int i = 0; i < 5; ++i) {
                g_fh += g_dep[i] / den[i];
                const float g_h = h[i] > 0.f ? -g_dep[i] * fh * a.down_ratio / (den[i] * den[i]) : 0.f;
                g_ky[ia[i]] += g_h;
                g_ky[ib[i]] -= g_h;
            }
            g_pd[1] += g_fh * sl.fu_rank;
            if (lane0)
                for (int k = 0; k < LR_NKP; ++k) gp[a.ch_corner + 2 * k + 1] = g_ky[k];
        }
    }

    // ---- dense keypoints: depth-weighted L1 in 2-D (depth_losses.py:50-67), L1 in 3-D (:176-186)
    {
        const float wgt = td < 5.f ? td * 0.01f : log10f(lr_max(td, 5.f) - 4.f) + 0.1f;
        float l2 = 0.f, n2 = 0.f, l3 = 0.f, n3 = 0.f;
        for (int k = w.lane(); k < K; k += w.lanes()) {
            const float *t2 = a.kpts2d + ((size_t)r * K + k) * 3, *t3 = a.kpts3d + ((size_t)r * K + k) * 3;
            const bool m2d = t2[2] != 0.f && found;
            const float ex = p[a.ch_kpts2d + 2 * k], ey = p[a.ch_kpts2d + 2 * k + 1];
            if (m2d) { l2 += (lr_abs(ex - t2[0]) + lr_abs(ey - t2[1])) * wgt; n2 += 1.f; }
            if (BWD) {
                gp[a.ch_kpts2d + 2 * k + 0] = m2d ? g[LR_L2D] * wgt * lr_sign(ex - t2[0]) : 0.f;
                gp[a.ch_kpts2d + 2 * k + 1] = m2d ? g[LR_L2D] * wgt * lr_sign(ey - t2[1]) : 0.f;
            }
            for (int i = 0; i < 3; ++i) {
                const float e = p[a.ch_kpts3d + 3 * k + i];
                if (found) l3 += lr_abs(e - t3[i]);
                if (BWD) gp[a.ch_kpts3d + 3 * k + i] = found ? g[LR_L3D] * lr_sign(e - t3[i]) : 0.f;
            }
            if (found) n3 += 1.f;
        }
        if (!BWD) {
            col[LR_L2D] = w.sum(l2);
            col[LR_M2D] = w.sum(n2);
            col[LR_L3D] = w.sum(l3);
            col[LR_M3D] = w.sum(n3);
        }
    }

    if (BWD) {
        // pair depths, second pass: valid pairs carry the L1 gradient, every pair a share of the mean's
        const float share = g_z / (float)NP;
        for (int j = w.lane(); j < NP; j += w.lanes()) {
            const float v = pdp[j];
            const bool valid = pmk[j] > 0.f && found;
            gpair[j] = share + (valid ? g[LR_VALID_L] * lr_sign(v - td) : 0.f);
        }
        if (lane0) {
            gp[a.ch_offset + 0] = g_o0;
            gp[a.ch_offset + 1] = g_o1;
            for (int i = 0; i < 3; ++i) gp[a.ch_dims + i] = g_pd[i] * pd[i];          // d exp(x) * mean / dx = the value
            for (int i = 0; i < 2 * LR_NBIN; ++i) {
                gp[a.ch_ori_cls + i] = g_oc[i];
                gp[a.ch_ori_off + i] = g_oo[i];
            }
        }
    } else if (lane0) {
        for (int c = 0; c < DCD_LOSS_ROWS_NCOL; ++c) a.cols[(size_t)c * BM + s] = col[c];
    }
}
