// Host instantiation of csrc/loss_rows_math.h with a one-lane "wave": lets tests/test_host_rows.py check the loss-row
// formulas and their hand-written derivatives against autograd on a machine without a GPU.  Test infrastructure only --
// nothing under dcd_amd/ links or loads this.
#include "../../dcd_amd/csrc/loss_rows_math.h"

namespace {
struct Wave1 {
    int lane() const { return 0; }
    int lanes() const { return 1; }
    float sum(float v) const { return v; }
    int min_int(int v) const { return v; }
    bool any(bool v) const { return v; }
};
}  // namespace

extern "C" {

void host_rows_prepare(const dcd_loss_rows_args *a)
{
    for (int s = 0; s < a->B * a->M; ++s) lr_prepare_row(*a, s, Wave1());
}

// the 3-D IoU column (a separate kernel on the GPU) is left at zero
void host_rows_forward(const dcd_loss_rows_args *a)
{
    const int BM = a->B * a->M;
    for (int s = 0; s < BM; ++s) lr_row<false>(*a, s, Wave1());
    for (int c = 0; c < DCD_LOSS_ROWS_NCOL; ++c) {
        double acc = 0.0;
        for (int s = 0; s < BM; ++s) acc += a->cols[(size_t)c * BM + s];
        a->sums[c] = (float)acc;
    }
}

void host_rows_backward(const dcd_loss_rows_args *a)
{
    for (int s = 0; s < a->B * a->M; ++s) lr_row<true>(*a, s, Wave1());
}

void host_rows_finish(const dcd_loss_rows_args *a)
{
    const int K = a->K;
    for (int s = 0; s < a->B * a->M; ++s) {
        if (!a->reg_mask[s]) continue;
        float *gp = a->grad_pois + (size_t)s * a->C;
        for (int i = 0; i < K * 2; ++i) gp[a->ch_kpts2d + i] += 4.f * a->grad_kps[(size_t)s * K * 2 + i];
        for (int i = 0; i < K * 3; ++i) gp[a->ch_kpts3d + i] += a->grad_kps3d[(size_t)s * K * 3 + i];
    }
}

}  // extern "C"
