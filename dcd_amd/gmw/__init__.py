"""GMW (graph matching weighting) train step -- SURVEY.md section 8(f) rank 1: the direct consumer of the edge-constraint
depth solver.  Mirrors GMW/model/model.py, GMW/lib/optimal_transport.py and the step of GMW/main.py:447-466."""
from .model import GMW, pairwise_l2_dist                      # noqa: F401
from .optimal_transport import RegularisedTransport          # noqa: F401
from .step import compute_reg_loss, correspondence_loss, gmw_losses, gmw_train_step, gmw_val_step    # noqa: F401
