"""Train-step pieces the benchmark and a `tools/plain_train_net.py`-style driver need.

Mirrors the semantics of the reference's harness without copying it:
  build_optimizer  <- DGDE/solver/__init__.py:10-62   AdamW(betas=(0.9,0.99), wd), bias parameters at lr x BIAS_LR_FACTOR
  build_scheduler  <- DGDE/solver/__init__.py:64-92   step decay (LambdaLR) + cosine warm-up from lr/DIV_FACTOR
  wrap_distributed <- DGDE/tools/plain_train_net.py:54-62   SyncBN conversion + DDP (one process per GPU, RCCL)
  train_step       <- DGDE/engine/trainer.py:121-155  forward, sum of losses, backward, clip, optimizer step
"""
import math
import weakref

import os

import torch
from torch import nn

from dcd_amd import ops
from dcd_amd.utils import comm


_OWN_ADAMW = os.environ.get("DCD_OWN_ADAMW", "1") != "0"      # 0: clip with _foreach ops + the library's fused AdamW (A/B timing)


class ClipAdamW(torch.optim.AdamW):
    """`torch.optim.AdamW` (same constructor, same `state` / `state_dict` layout, same `step()`) with one more method:
    `clip_and_step(max_norm)` = `clip_grad_norm_(params, max_norm)` + the non-finite guard + `step()` on csrc/optim.hip -- the gradient
    norm in one pass, the clip coefficient applied inside the AdamW kernel, 2 048 elements per block: 7 launches and 0.14 ms per step
    where `_foreach_norm` / `_foreach_mul_` / the library's fused kernel took 29 and 0.43 ms (two blocks per CU at 2 TB/s).  The
    arithmetic is the library's fused kernel's, operation for operation (tests/test_gpu_optim.py).  Needs what `build_optimizer`
    sets up on the device: fused + capturable groups with tensor learning rates, fp32 contiguous parameters; anything else falls
    back to the three library calls."""

    def own_kernels_ok(self):
        if not _OWN_ADAMW or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return False
        for g in self.param_groups:
            if not (g.get("fused") and g.get("capturable") and torch.is_tensor(g["lr"]) and g["lr"].is_cuda and g["lr"].dtype == torch.float32
                    and not g["amsgrad"] and not g["maximize"] and not g.get("differentiable")):
                return False
        return True

    @torch.no_grad()
    def clip_and_step(self, max_norm):
        """Returns the total gradient norm (0-dim device tensor), like `clip_grad_norm_`.  A non-finite norm leaves parameters, moments and
        step counters as they were."""
        from .. import _lib
        import ctypes
        per_group, all_grads = [], []
        for group in self.param_groups:
            params, grads, m, v, mx, steps = [], [], [], [], [], []
            self._init_group(group, params, grads, m, v, mx, steps)
            per_group.append((group, params, grads, m, v, steps))
            all_grads += grads
        tensors = [t for _, ps, gs, ms, vs, ss in per_group for t in ps + gs + ms + vs + ss]
        if not all_grads or any(t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda for t in tensors):
            total = clip_grad_norm([p for g in self.param_groups for p in g["params"]], max_norm)      # the library path
            guard_nonfinite_step(self, total)
            self.step()
            self.found_inf = None
            return total
        L = _lib.lib()
        dev = all_grads[0].device
        stream = _lib.stream_of(all_grads[0])

        def table(ts):
            return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

        def counts(ts):
            return (ctypes.c_int64 * len(ts))(*[t.numel() for t in ts])

        n_all = counts(all_grads)
        nbytes = L.dcd_clip_adamw_workspace_bytes(len(all_grads), n_all)
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
        scal = torch.empty(4, dtype=torch.float32, device=dev)
        _lib.check(L.dcd_clip_grad_norm_scalars(stream, len(all_grads), table(all_grads), n_all, float(max_norm or 0.0), ws.data_ptr(),
                                                int(nbytes), scal.data_ptr()), "dcd_clip_grad_norm_scalars")
        for group, params, grads, m, v, steps in per_group:
            if not params:
                continue
            beta1, beta2 = group["betas"]
            _lib.check(L.dcd_adamw_apply(stream, len(params), table(params), table(grads), table(m), table(v), table(steps), counts(params),
                                         group["lr"].data_ptr(), float(beta1), float(beta2), float(group["eps"]),
                                         float(group["weight_decay"]), scal.data_ptr()), "dcd_adamw_apply")
        self._opt_called = True                                  # (what the schedulers' call-order check looks at)
        return scal[0]


def build_optimizer(model, cfg):
    """Same per-parameter learning rates as the reference (one group per parameter there); parameters are pooled
    into two groups (weights, biases) so the optimizer runs as a handful of fused kernels instead of 290 x k."""
    s = cfg.SOLVER
    weights, biases = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (biases if "bias" in name else weights).append(p)
    groups = [{"params": weights, "lr": s.BASE_LR},
              {"params": biases, "lr": max(s.BASE_LR, s.BASE_LR * s.BIAS_LR_FACTOR)}]
    fused = all(p.is_cuda for p in weights + biases)
    # On the device the step may be replayed from a HIP graph (GraphedTrainStep): the step counters then live on the device
    # (capturable) and the learning rates are 0-dim device tensors the schedulers fill in place -- a Python float would be
    # frozen into the captured launch.
    kw = dict(weight_decay=s.WEIGHT_DECAY, betas=(0.9, 0.99), fused=fused)
    if fused:
        kw["capturable"] = True
        dev = weights[0].device if weights else biases[0].device
        for g_ in groups:
            g_["lr"] = torch.tensor(float(g_["lr"]), dtype=torch.float32, device=dev)
    if s.OPTIMIZER == "adamw":
        return (ClipAdamW if fused else torch.optim.AdamW)(groups, lr=s.BASE_LR, **kw)
    if s.OPTIMIZER == "adam":
        return torch.optim.Adam(groups, lr=s.BASE_LR, **kw)
    raise NotImplementedError("SOLVER.OPTIMIZER=%s" % s.OPTIMIZER)


class CosineWarmupLR(torch.optim.lr_scheduler.LRScheduler):
    """lr(t) = eta_min + (base - eta_min) * (1 - cos(pi t / T_max)) / 2 (DGDE/solver/learning_schedules_fastai.py)."""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.T_max, self.eta_min = T_max, eta_min
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return [self.eta_min + (b - self.eta_min) * (1 - math.cos(math.pi * self.last_epoch / self.T_max)) / 2
                for b in self.base_lrs]


def build_scheduler(optimizer, cfg):
    s = cfg.SOLVER

    def decay(it):
        f = 1.0
        for step in s.STEPS:
            if it >= step:
                f *= s.LR_DECAY
        return max(f, s.LR_CLIP / s.BASE_LR)
    sched = torch.optim.lr_scheduler.LambdaLR(optimizer, decay)
    warm = CosineWarmupLR(optimizer, T_max=s.WARMUP_STEPS, eta_min=s.BASE_LR / s.DIV_FACTOR) if s.LR_WARMUP else None
    return sched, warm


def enable_sync_bn(model, group=None):
    """MODEL.USE_SYNC_BN (DGDE/tools/plain_train_net.py:56-57).  The fused BatchNorm2d modules all-reduce their fp64
    per-channel sums themselves (one small RCCL call per BN per pass, no module swap, the ReLU / residual fusion stays);
    the two BatchNorm1d of the edge-fusion branch become torch SyncBatchNorm (GPU only -- torch has no CPU SyncBN)."""
    import torch.distributed as dist
    from dcd_amd.model.layers.norm import BatchNorm2d
    group = dist.group.WORLD if group is None else group
    for name, m in list(model.named_modules()):
        if isinstance(m, BatchNorm2d):
            m.sync_group = group
    if next(model.parameters()).is_cuda:
        for parent in list(model.modules()):
            for cname, child in list(parent.named_children()):
                if isinstance(child, nn.BatchNorm1d):
                    setattr(parent, cname, nn.SyncBatchNorm.convert_sync_batchnorm(child, group))
    return model


def step_schedulers(scheduler, warmup_scheduler, iteration, cfg):
    """The reference's per-iteration rule (DGDE/engine/trainer.py:90-95,152-155): the cosine warm-up drives the learning rate
    for the first SOLVER.WARMUP_STEPS iterations (when SOLVER.LR_WARMUP), the step-decay LambdaLR afterwards; both are stepped
    with the absolute iteration number."""
    warmup_iters = cfg.SOLVER.WARMUP_STEPS if cfg.SOLVER.LR_WARMUP else -1
    if iteration < warmup_iters:
        warmup_scheduler.step(iteration)
    else:
        scheduler.step(iteration)


def _reference_order(model, optimizer):
    """For every trainable parameter in `named_parameters` order -- the order in which the reference creates its one-parameter
    groups (DGDE/solver/__init__.py:10-25) -- (our group index, our flat state index).  Parameters that are frozen here but
    trainable in the reference (the dead `Tree.project` convs under DDP) get None: they never receive a gradient there either,
    so the reference keeps no optimizer state for them."""
    m = model.module if isinstance(model, nn.parallel.DistributedDataParallel) else model
    where, flat = {}, 0
    for gi, g in enumerate(optimizer.param_groups):
        for p in g["params"]:
            where[id(p)] = (gi, flat)
            flat += 1
    frozen_here = {id(p) for mod in m.modules() if getattr(mod, "dead_project", False) for p in mod.project.parameters()}
    order = []
    for _, p in m.named_parameters():
        if id(p) in where:
            order.append(where[id(p)])
        elif id(p) in frozen_here or p.requires_grad:
            order.append(None)
    return order


def _host_state_value(key, v):
    """A host copy of one optimizer-state entry (never the live tensor); device step counters are left for the caller's
    single stacked copy."""
    if not torch.is_tensor(v):
        return v
    if key == "step":
        return v if v.is_cuda else v.detach().clone().to(torch.float32)
    return v.detach().cpu().clone() if not v.is_cuda else v.detach().cpu()


def optimizer_state_to_reference(model, optimizer):
    """Our pooled AdamW state (two groups: weights, biases) re-expressed in the reference's layout: one group per parameter in
    `named_parameters` order, state keyed by that index -- what `utils/check_point.py:31-43` saves and
    `SOLVER.LOAD_OPTIMIZER_SCHEDULER` loads."""
    sd = optimizer.state_dict()
    groups, state = [], {}
    for ref_idx, loc in enumerate(_reference_order(model, optimizer)):
        if loc is None:                                   # frozen here: hyper-parameters of a weight group, no state
            g = {k: v for k, v in sd["param_groups"][0].items() if k != "params"}
        else:
            g = {k: v for k, v in sd["param_groups"][loc[0]].items() if k != "params"}
            if loc[1] in sd["state"]:
                # Optimizer.state_dict() hands out the LIVE per-parameter dicts: build new ones (host copies, float step
                # counter) -- writing into them would turn the running optimizer's moments into CPU tensors (advisor r3)
                state[ref_idx] = {k: _host_state_value(k, v) for k, v in sd["state"][loc[1]].items()}
        g["params"] = [ref_idx]
        # The reference's optimizer is the plain for-loop AdamW and adopts the saved groups verbatim after
        # torch.load(map_location=cpu) (DGDE/utils/check_point.py:138): device learning-rate tensors, `capturable` and the
        # fused flag of OUR optimizer must not travel -- plain floats, capturable off, host step counters.
        g["fused"] = None
        g["foreach"] = None
        g["capturable"] = False
        for k in ("lr", "initial_lr", "weight_decay", "eps"):
            if torch.is_tensor(g.get(k)):
                g[k] = float(g[k])
        groups.append(g)
    # step counters: device tensors under capturable AdamW -> ONE stacked copy to the host instead of a sync per parameter
    dev_steps = [(i, st["step"]) for i, st in state.items() if torch.is_tensor(st.get("step")) and st["step"].is_cuda]
    if dev_steps:
        host = torch.stack([t.detach().reshape(()).to(torch.float32) for _, t in dev_steps]).cpu()
        for (i, _), v in zip(dev_steps, host):
            state[i]["step"] = v.clone()
    return {"state": state, "param_groups": groups}


def optimizer_state_from_reference(ref_sd, model, optimizer):
    """Inverse of optimizer_state_to_reference: a reference checkpoint's 'optimizer' entry -> a state dict this optimizer
    loads.  Learning rates are taken per pooled group from the first member (all members share it by construction)."""
    order = _reference_order(model, optimizer)
    if len(ref_sd["param_groups"]) != len(order):
        raise ValueError("optimizer state has %d parameter groups, the model has %d trainable parameters"
                         % (len(ref_sd["param_groups"]), len(order)))
    ours = optimizer.state_dict()
    state, seen = {}, set()
    for ref_idx, loc in enumerate(order):
        if loc is None:
            continue
        rg = ref_sd["param_groups"][ref_idx]
        if loc[0] not in seen:
            seen.add(loc[0])
            og = ours["param_groups"][loc[0]]
            for k, v in rg.items():
                if k in ("params", "fused", "foreach", "capturable"):
                    continue                               # how THIS optimizer runs is not the checkpoint's business
                if torch.is_tensor(og.get(k)):
                    # our learning rates are device tensors (capturable fused AdamW; GraphedTrainStep addresses them): keep
                    # the tensor, take the value -- a float here would freeze the rate inside a captured step
                    og[k] = og[k].clone().fill_(float(v))
                else:
                    og[k] = float(v) if torch.is_tensor(v) else v
        if ref_idx in ref_sd["state"]:
            state[loc[1]] = ref_sd["state"][ref_idx]
    ours["state"] = state
    return ours


def _scheduler_state(sd, pick):
    """Per-group lists of an LR scheduler's state (base_lrs, _last_lr, lr_lambdas) re-indexed by `pick`."""
    out = dict(sd)
    for k in ("base_lrs", "_last_lr", "lr_lambdas"):
        if isinstance(out.get(k), (list, tuple)):
            out[k] = [out[k][i] for i in pick]
    return out


def checkpoint_state(model, optimizer=None, scheduler=None, **extra):
    """Checkpoint dict in the reference's layout (DGDE/utils/check_point.py:31-43): {'model', 'optimizer', 'scheduler'} plus
    the trainer's extras ('iteration', 'iter_per_epoch', DGDE/engine/trainer.py:167-171).  State-dict keys of the model equal
    the reference's; the optimizer and scheduler entries are converted to the reference's one-group-per-parameter layout, so a
    file written here resumes there (SOLVER.LOAD_OPTIMIZER_SCHEDULER) and the other way round."""
    m = model.module if isinstance(model, nn.parallel.DistributedDataParallel) else model
    data = {"model": m.state_dict()}
    if optimizer is not None:
        data["optimizer"] = optimizer_state_to_reference(model, optimizer)
    if scheduler is not None and hasattr(scheduler, "state_dict"):
        sd = scheduler.state_dict()
        if optimizer is not None:
            order = _reference_order(model, optimizer)
            sd = _scheduler_state(sd, [0 if loc is None else loc[0] for loc in order])
        data["scheduler"] = sd
    data.update(extra)
    return data


def load_checkpoint_state(data, model, optimizer=None, scheduler=None):
    """Inverse of checkpoint_state; accepts files written by the reference (per-parameter groups).  Returns the extras
    (iteration, iter_per_epoch, ...)."""
    m = model.module if isinstance(model, nn.parallel.DistributedDataParallel) else model
    data = dict(data)
    m.load_state_dict(data.pop("model"))
    if optimizer is not None and "optimizer" in data:
        # Optimizer.load_state_dict deep-copies the groups: a device learning-rate tensor would come back as a NEW tensor, while a
        # captured step (GraphedTrainStep) and the scheduler keep addressing the old one.  Keep the objects, move the values.
        keep = [{k: v for k, v in g.items() if torch.is_tensor(v) and k != "params"} for g in optimizer.param_groups]
        optimizer.load_state_dict(optimizer_state_from_reference(data.pop("optimizer"), model, optimizer))
        for g, old in zip(optimizer.param_groups, keep):
            for k, t in old.items():
                t.fill_(float(g[k]))
                g[k] = t
    if scheduler is not None and "scheduler" in data:
        sd = data.pop("scheduler")
        if optimizer is not None:
            order = _reference_order(model, optimizer)
            first = {}
            for ref_idx, loc in enumerate(order):
                if loc is not None:
                    first.setdefault(loc[0], ref_idx)
            sd = _scheduler_state(sd, [first[g] for g in range(len(optimizer.param_groups))])
        scheduler.load_state_dict(sd)
    return data


def wrap_distributed(model, cfg, local_rank):
    """SyncBN (when MODEL.USE_SYNC_BN) + DistributedDataParallel over RCCL.  Unlike the reference no unused-parameter
    search is needed (the ImageNet `fc` is never attached), and gradients live in the all-reduce buckets."""
    import os
    if comm.get_world_size() == 1 and os.environ.get("DCD_FORCE_DDP", "0") != "1":
        return model      # DCD_FORCE_DDP=1: wrap anyway (exercises SyncBN + DDP + RCCL on a single GPU)
    # parameters that structurally never get a gradient (the reason the reference needs find_unused_parameters=True,
    # besides the ImageNet `fc`): freeze them so DDP does not wait for their all-reduce.  Equivalent to the reference,
    # whose optimizer skips parameters whose grad is None.
    for m in model.modules():
        if getattr(m, "dead_project", False):
            for p in m.project.parameters():
                p.requires_grad_(False)
    if cfg.MODEL.USE_SYNC_BN:
        enable_sync_bn(model)
    kw = dict(broadcast_buffers=False, find_unused_parameters=False, gradient_as_bucket_view=True, bucket_cap_mb=32)
    if next(model.parameters()).is_cuda:
        kw.update(device_ids=[local_rank], output_device=local_rank)
    return nn.parallel.DistributedDataParallel(model, **kw)


def prepare_data_parallel(model, cfg):
    """What `wrap_distributed` does to the model itself -- SyncBN, frozen dead projections -- WITHOUT the DistributedDataParallel
    wrapper: `GraphedTrainStep(..., distributed=True)` all-reduces the gradients itself, as one flat buffer inside the captured
    step (at one image per rank there is nothing for DDP's bucket overlap to hide behind: the backward is 12 ms, the 85 MB
    all-reduce < 1.5 ms, and the reducer's hooks are host work the graph exists to remove)."""
    for m in model.modules():
        if getattr(m, "dead_project", False):
            for p in m.project.parameters():
                p.requires_grad_(False)
    if cfg.MODEL.USE_SYNC_BN:
        enable_sync_bn(model)
    return model


_PARAM_LISTS = weakref.WeakKeyDictionary()


def _parameters_of(model):
    """`list(model.parameters())`, built once per model: the module-tree walk costs 1.2 ms per step, which shows at one
    image per GPU where the step is launch-bound (tools/cpu_profile_step.py)."""
    params = _PARAM_LISTS.get(model)
    if params is None:
        params = _PARAM_LISTS[model] = list(model.parameters())
    return params


def clip_grad_norm(params, max_norm):
    """`torch.nn.utils.clip_grad_norm_(params, max_norm)` (2-norm, the reference's call, DGDE/engine/trainer.py:144)
    without its per-call grouping by device / dtype: three multi-tensor launches, no host sync."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return torch.zeros(())
    total = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
    torch._foreach_mul_(grads, (max_norm / (total + 1e-6)).clamp(max=1.0))
    return total


def guard_nonfinite_step(optimizer, total_norm):
    """A batch without a single annotated object (or any other source of Inf/NaN) must not reach the weights; the reference
    stops with an exception there (its num_reg_3D guard is followed by an UnboundLocalError).  On the GPU the fused AdamW
    kernel takes a device flag (`found_inf`, the GradScaler hook) and skips the whole update when it is set -- no host sync,
    no extra launch; the log dict still raises when somebody reads it.  On the host we simply raise."""
    if total_norm.is_cuda:
        if optimizer.defaults.get("fused"):
            optimizer.found_inf = (~torch.isfinite(total_norm)).to(torch.float32).reshape(())
    elif not bool(torch.isfinite(total_norm)):
        raise FloatingPointError("non-finite gradient norm: %r" % float(total_norm))


def clip_and_step(model, optimizer, grad_norm_clip):
    """The tail of a step: clip_grad_norm_(parameters, clip), the non-finite guard, optimizer.step() (DGDE/engine/trainer.py:144-147)."""
    if grad_norm_clip and grad_norm_clip > 0:
        if isinstance(optimizer, ClipAdamW) and optimizer.own_kernels_ok():
            return optimizer.clip_and_step(grad_norm_clip)
        guard_nonfinite_step(optimizer, clip_grad_norm(_parameters_of(model), grad_norm_clip))
    optimizer.step()


def train_step(model, optimizer, images, targets, grad_norm_clip=15.0, scheduler=None, iteration=None):
    """One optimisation step; returns (loss_dict, log_loss_dict)."""
    optimizer.zero_grad(set_to_none=True)        # before the forward: nothing on the host between the loss and its backward
    ops.refresh_conv_weights()                   # Winograd-domain weights of all 3x3 convolutions: one launch per step
    loss_dict, log_loss_dict = model(images, targets)
    losses = getattr(loss_dict, "total", None)
    if losses is None:
        losses = sum(loss_dict.values())
    losses.backward()
    clip_and_step(model, optimizer, grad_norm_clip)
    if scheduler is not None:
        scheduler.step(iteration) if iteration is not None else scheduler.step()
    return loss_dict, log_loss_dict


class GraphedTrainStep:
    """`train_step` replayed from ONE HIP graph per input signature: forward, 13-term loss, backward, clip, AdamW -- ~2 600
    launches at bs 8, ~60 ms of host work -- become a single `hipGraphLaunch`.  What makes that possible is already in place:
    no host synchronisation anywhere in the step (masked object slots, lazy log dict, device-side non-finite guard), caller-
    sized workspaces from the caching allocator, fused capturable AdamW with tensor learning rates.

    Inputs are copied into static buffers before each replay (images, every tensor field of the targets, and the per-image
    intrinsics as a (6,) row: Calibration objects are host data).  Before capturing, `warmup` eager steps run on a side
    stream so that lazily created handles / MIOpen selections exist; model, BN buffers and optimizer state are restored
    afterwards, so the captured step is the first one that counts.  Data parallel: `distributed=True` (see __init__) captures the
    SyncBN and gradient collectives with the kernels; a DDP-wrapped model is not accepted.  Returned loss tensors are the graph's
    static outputs (read them before the next call).

    Opt-in (bench.py: DCD_STEP_GRAPH=1).  Checked at 96x320 against eager steps (tests/test_gpu_golden.py, opt-in) and at
    384x1280 over eight steps (tools/check_step_graph.py: same loss trajectory, parameter distance equal to the distance between
    two eager runs).  Earlier in round 2 the second replay faulted at full size; it stopped when the last per-step
    host-to-device constants left the step.  Flat multi-block ATen reductions inside a captured graph did return wrong sums on
    this stack (profiles/r02_graph_memset_hazard.txt), so the graphed loss uses two-stage sums and our kernels zero-fill with
    kernels (csrc/zero_fill.h)."""

    def __init__(self, model, optimizer, grad_norm_clip=15.0, warmup=2, distributed=False, group=None, recapture_every=None):
        """distributed=True (round 3): data parallel INSIDE the graph -- the model is the bare module prepared by
        `prepare_data_parallel` (SyncBN on), its SyncBN all-reduces and ONE all-reduce of the flat gradient buffer are captured
        with the kernels (RCCL collectives are stream work like any other; DGDE/tools/plain_train_net.py:54-62 is DDP + SyncBN
        around the same step).  Every rank must call the step the same number of times.

        recapture_every=N: drop the graphs after every N replays, so the next call captures again.  A captured step freezes what
        the library decides on the host per call -- above all the DCNv2 launch policy (include/dcd_hip.h, dcd_dcn_v2_forget: a
        layer whose offsets were small at capture time replays the "never hand over" sequence, correct for any offsets but slower
        once many samples are displaced by 3 px or more).  A training run whose learned offsets grow should re-capture now and then
        (every few thousand steps costs nothing measurable: a capture is ~3 eager steps); `recapture()` does it on demand.  The
        count is per call of this object, so data-parallel ranks re-capture in the same call."""
        self.model, self.optimizer, self.clip, self.warmup = model, optimizer, grad_norm_clip, warmup
        self._s2d_prev = None
        if any(p.is_cuda for p in model.parameters()):
            # no MIOpen backward-data solver (memset + accumulate) in the captured step; handed back when a capture fails and no
            # graph of this object is left (the eager fallback then runs the faster stock solver again)
            self._s2d_prev = ops.stride2_on_own_kernels()
        self.distributed, self.group = distributed, group
        self.recapture_every, self._replays = recapture_every, 0
        self._graphs = {}
        self._flat = None
        self._side = None
        self.capture_error = None
        self._fail_capture = False
        if distributed:
            if isinstance(model, nn.parallel.DistributedDataParallel):
                raise ValueError("GraphedTrainStep(distributed=True) takes the bare module (prepare_data_parallel), not a DDP wrapper")
            params = [p for p in model.parameters() if p.requires_grad]
            self._dp_params = params
            self._flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
            views, o = [], 0
            for p in params:
                views.append(self._flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            self._dp_views = views
            import torch.distributed as dist
            self._world = dist.get_world_size(group)

    def _reduce_gradients(self):
        """grads -> flat buffer (multi-tensor copy), ONE all-reduce, mean; the parameters' .grad then ARE the flat views, so the
        clip and the fused AdamW read the reduced values.  A parameter that got no gradient in this step keeps `.grad = None`
        -- the optimizer skips it, as eager DDP and the reference do (no weight decay / moment update on it; advisor r3); its
        slice of the flat buffer stays zero.  Which parameters those are is a property of the model, the same on every rank."""
        import torch.distributed as dist
        have = [(v, p) for v, p in zip(self._dp_views, self._dp_params) if p.grad is not None]
        if len(have) != len(self._dp_params):
            self._flat.zero_()
        torch._foreach_copy_([v for v, _ in have], [p.grad for _, p in have])
        dist.all_reduce(self._flat, group=self.group)
        if self._world > 1:
            self._flat.mul_(1.0 / self._world)
        for v, p in have:
            p.grad = v

    @staticmethod
    def _calib_row(c):
        return [float(c.c_u), float(c.c_v), float(c.f_u), float(c.f_v), float(c.b_x), float(c.b_y)]

    @staticmethod
    def _signature(images, targets):
        sig = [tuple(images.shape), images.dtype]
        for t in targets:
            sig.append(tuple((k, tuple(v.shape), v.dtype) for k, v in t.extra_fields.items() if torch.is_tensor(v)))
        return tuple(sig)

    def _static_copy(self, images, targets):
        from dcd_amd.structures.params_3d import ParamsList
        dev = images.device
        st_images = images.detach().clone()
        st_targets, calib = [], torch.zeros((len(targets), 6), dtype=torch.float32, device=dev)
        for i, t in enumerate(targets):
            c = ParamsList(t.size, t.is_train)
            for k, v in t.extra_fields.items():
                if torch.is_tensor(v):
                    c.extra_fields[k] = v.detach().to(dev).clone()
                elif k == "calib":
                    c.extra_fields[k] = calib[i]                      # a view: filled by _copy_in, consumed as a table row
                else:
                    c.extra_fields[k] = v
            st_targets.append(c)
        host = torch.zeros((len(targets), 6), dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros((len(targets), 6))
        return st_images, st_targets, calib, host

    @staticmethod
    def _copy_in(entry, images, targets):
        st_images, st_targets, calib, host = entry["static"]
        if st_images.data_ptr() != images.data_ptr():
            st_images.copy_(images, non_blocking=True)
        rows = []
        for c, t in zip(st_targets, targets):
            for k, v in t.extra_fields.items():
                if torch.is_tensor(v):
                    dst = c.extra_fields[k]
                    if dst.data_ptr() != v.data_ptr():
                        dst.copy_(v, non_blocking=True)
                elif k == "calib":
                    rows.append(GraphedTrainStep._calib_row(v))
        if rows and rows != entry.get("rows"):
            entry["rows"] = rows
            host.copy_(torch.tensor(rows, dtype=torch.float32))
            calib.copy_(host, non_blocking=True)

    def _eager(self, images, targets):
        self.optimizer.zero_grad(set_to_none=True)
        ops.refresh_conv_weights()
        loss_dict, log = self.model(images, targets)
        total = getattr(loss_dict, "total", None)
        if total is None:
            total = sum(loss_dict.values())
        total.backward()
        if self.distributed:
            self._reduce_gradients()
        clip_and_step(self.model, self.optimizer, self.clip)
        return loss_dict, log

    def _capture(self, images, targets):
        import copy
        entry = {"static": self._static_copy(images, targets)}
        self._copy_in(entry, images, targets)
        st_images, st_targets = entry["static"][0], entry["static"][1]
        for m in self.model.modules():                   # the loss section's own graphs would nest inside this one
            if hasattr(getattr(m, "loss_evaluator", None), "use_graph"):
                m.loss_evaluator.use_graph = False
        on_gpu = st_images.is_cuda
        model_state = copy.deepcopy(self.model.state_dict())
        saved = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in self.optimizer.state.items()}
        try:
            # warm-up: every rank runs the same eager steps (same collectives, in the same order) whatever happens later
            n_warm = max(self.warmup, 3 if self.distributed else 1)   # >= 1: the optimizer's state tensors must exist before the
            if on_gpu:                                                # capture; collectives: communicators set up eagerly first
                # ONE side stream for the warm-up and the capture: per-stream scratch our ops keep (zeroed arrival counters)
                # is then created by the warm-up, not inside the graph
                side = self._side = self._side or torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(n_warm):
                        self._eager(st_images, st_targets)
                torch.cuda.current_stream().wait_stream(side)
            else:
                for _ in range(n_warm):
                    self._eager(st_images, st_targets)
        finally:
            self.model.load_state_dict(model_state)            # in place: parameters and BN buffers keep their storage
            with torch.no_grad():
                for p_, st in self.optimizer.state.items():    # in place too: the graph will address these very tensors
                    old = saved.get(id(p_))
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(old[k]) if old is not None and k in old else v.zero_()
            self.optimizer.zero_grad(set_to_none=True)
        if on_gpu:
            # The DCN calls choose their launch sequence per layer from what the layer's previous call reported to the host
            # (csrc/dcn_v2.hip, hand-over policy); the graph freezes that choice, so let the warm-up's reports arrive first.
            torch.cuda.synchronize()
        # from here on nothing is executed: a failure below leaves every rank with the same collective history
        if self._fail_capture:                               # test hook: this rank's capture "fails" (after the common warm-up)
            raise RuntimeError("capture failure requested (test hook)")
        entry["graph"], entry["out"] = self._record_graph(st_images, st_targets)
        return entry

    def _record_graph(self, st_images, st_targets):
        if not st_images.is_cuda:
            raise RuntimeError("a HIP graph needs a GPU")
        graph = torch.cuda.CUDAGraph()
        if self.distributed:
            # the process group's watchdog thread polls the events of the warm-up collectives; in the default (global) capture
            # mode such a query from ANOTHER thread aborts the capture ("operation not permitted when stream is capturing",
            # seen 1 run in 3) -> let the queue drain and restrict the capture checks to this thread
            import time
            torch.cuda.synchronize()
            time.sleep(0.2)
            ctx = torch.cuda.graph(graph, stream=self._side, capture_error_mode="thread_local")
        else:
            ctx = torch.cuda.graph(graph, stream=self._side)
        with ctx:
            loss_dict, log = self._eager(st_images, st_targets)
        return graph, (loss_dict, log)

    def capture(self, images, targets):
        """Warm up and capture the step for this input signature WITHOUT replaying it; returns True when a graph is ready.
        Nothing a failed capture did survives: model, BN buffers and optimizer state are the pre-call values either way, and no
        collective has been issued by the capture itself (captured collectives are recorded, not run) -- so under data
        parallelism the ranks can still agree on what to do next (`agree`) before any of them replays."""
        key = self._signature(images, targets)
        if key in self._graphs:
            return True
        if len(self._graphs) >= 2:                     # e.g. the last, smaller batch of an epoch
            self._graphs.pop(next(iter(self._graphs)))
        try:
            self._graphs[key] = self._capture(images, targets)
            return True
        except Exception as e:                          # noqa: BLE001 -- whatever it was, the caller falls back to the eager step
            self.capture_error = e
            self._graphs.pop(key, None)
            self.optimizer.zero_grad(set_to_none=True)  # the flat views must not stay behind as .grad for an eager / DDP step
            if not self._graphs and self._s2d_prev is not None:
                ops.restore_stride2_mode(self._s2d_prev)
            return False

    def agree(self, ok):
        """Data parallel: True only if EVERY rank captured (one eager all-reduce of a flag, issued by all ranks at the same
        point: after `capture`, before any replay).  A rank whose capture failed must not meet its peers inside the replayed
        SyncBN all-reduces with a different collective -- that is a hang, not a fallback (advisor r3)."""
        if not self.distributed:
            return bool(ok)
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._flat.device)   # issued by every rank, capture or not
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        agreed = bool(flag.item())
        if not agreed:
            self._graphs.clear()
            if self._s2d_prev is not None:
                ops.restore_stride2_mode(self._s2d_prev)
        return agreed

    def recapture(self):
        """Forget the captured graphs: the next call warms up and captures again, with the launch decisions of that moment (see
        `recapture_every`).  Data parallel: call it on every rank before the same step."""
        self._graphs.clear()
        self._replays = 0

    def replay(self, images, targets):
        entry = self._graphs[self._signature(images, targets)]
        self._copy_in(entry, images, targets)
        entry["graph"].replay()
        self._replays += 1
        # the replay stepped the optimizer without touching the parameters' version counters: an eager forward that follows must
        # not take the transformed 3x3 weights of the step before for current (the graph itself re-transforms them every replay)
        ops.invalidate_conv_weights()
        return entry["out"]

    def __call__(self, images, targets):
        """capture (first call per signature) + cross-rank agreement + replay.  Under data parallelism every rank must meet an
        unseen signature in the SAME call (fixed shapes per rank: the synthetic batches, a drop-last loader): a rank that
        captures while its peers replay issues different collectives.  Raises when the ranks did not all capture; the caller
        then owns the fallback (bench.py: eager DDP step)."""
        if self.recapture_every and self._replays >= self.recapture_every:
            self.recapture()
        key = self._signature(images, targets)
        if key not in self._graphs:
            if self.distributed and self._graphs:
                # a second signature on THIS rank: its peers may be replaying the first one right now -- a capture here (warm-up
                # collectives, the vote) would meet their replayed collectives.  Refuse instead of hanging.
                raise RuntimeError("GraphedTrainStep(distributed=True): the input signature changed after the capture "
                                   "(%r); data-parallel ranks must keep one fixed signature (pad or drop the last batch), "
                                   "or call recapture() on every rank before the step that changes it" % (key[0],))
            ok = self.agree(self.capture(images, targets))
            if not ok:
                raise RuntimeError("whole-step graph unavailable on at least one rank: %r" % (self.capture_error,))
        return self.replay(images, targets)


def init_like_trained(model, std=0.01, seed=0):
    """The reference zero-initialises `conv_offset_mask`, which turns every DCN into a plain conv x 0.5; benchmarks use
    N(0, std^2) offset/mask weights so sampling positions are non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "conv_offset_mask.weight" in name:
                p.copy_(torch.randn(p.shape, generator=g) * std)
