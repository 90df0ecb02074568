"""Training-driver pieces: the Ranger optimizer as ONE fused device step, gradient-norm clipping without a
host round trip, and the flat-and-anneal learning-rate schedule.

Mirrors (same names, arguments and state_dict layout):
  Ranger                         tools/torch_utils/solver/ranger2020.py:44-246
  flat_and_anneal_lr_scheduler   tools/torch_utils/solver/lr_scheduler.py:177-263
  build_optimizer, build_lr_rate tools/training_utils.py:13-56
  clip_grad_norm_                torch.nn.utils.clip_grad_norm_ as used in engine/train.py:99,104

The reference's ``Ranger.step`` walks the parameter tensors in Python (~10 elementwise launches each).  Here
the parameters of a group are re-seated as views of one flat fp32 buffer, and so are their gradients and the
three state tensors; ``step()`` is a single ``hsp_ranger_step`` launch per group (csrc/optim.hip) driven by a
device table of rows.  ``state_dict()`` still holds per-parameter ``step / exp_avg / exp_avg_sq / slow_buffer``
entries (views), so checkpoints written by either implementation load into the other.
"""
import ctypes
import math

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

from . import ops
from ._lib import HspError, lib
from .config import FLAGS

_ROW_CHUNK = 4096          # 1-D tensors are cut into rows of at most this many elements


class _FlatGroup:
    """flat storage of one param group: parameters, gradients, exp_avg, exp_avg_sq, slow weights + the row table."""

    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise HspError("Ranger: empty parameter group")
        dev = self.params[0].device
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or p.device != dev:
                raise HspError("Ranger: parameters must be fp32 tensors on one GPU (hs_pose_amd has no CPU path)")
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) & ~3                      # 16-byte aligned starts
        self.offsets, self.total = offs, total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_s = torch.zeros(total, dtype=torch.float32, device=dev)
        rows = []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.view(self.flat_p, p, o).copy_(p.data)
            p.data = self.view(self.flat_p, p, o)              # the parameter now lives in the flat buffer
            g = self.view(self.flat_g, p, o)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g                                         # autograd accumulates in place into the flat buffer
            if p.dim() > 1:
                rl = n // p.shape[0]
                rows += [(o + r * rl, rl, 1) for r in range(p.shape[0])]
            else:
                rows += [(o + c, min(_ROW_CHUNK, n - c), 0) for c in range(0, n, _ROW_CHUNK)]
        # slow_buffer is filled from the weights at the FIRST step(), like the reference creates it lazily
        # (ranger2020.py:168-170): weights loaded between construction and the first step must be what Lookahead
        # interpolates towards, not the random init that was live when the optimizer was built
        self.slow_ready = False
        tab = np.zeros(len(rows), dtype=np.dtype([("offset", np.int64), ("len", np.int32), ("gc", np.int32)]))
        for i, (o, l, gc) in enumerate(rows):
            tab[i] = (o, l, gc)
        self.gc_rows = tab["gc"].copy()
        self.rows_host = tab
        self.rows = torch.from_numpy(tab.view(np.uint8)).to(dev)
        self.nrows = len(rows)

    @staticmethod
    def view(flat, p, off):
        return flat[off:off + p.numel()].view(p.shape)

    def rows_for(self, use_gc, gc_conv_only):
        """row table with the gc flag resolved (conv-only: tensors with more than 3 dims, ranger2020.py:34-36)."""
        if use_gc and not gc_conv_only:
            return self.rows
        key = (use_gc, gc_conv_only)
        cache = self.__dict__.setdefault("_rows_cache", {})
        if key not in cache:
            tab = self.rows_host.copy()
            if not use_gc:
                tab["gc"] = 0
            else:
                i = 0
                for p in self.params:
                    nr = p.shape[0] if p.dim() > 1 else (p.numel() + _ROW_CHUNK - 1) // _ROW_CHUNK
                    if p.dim() <= 3:
                        tab["gc"][i:i + nr] = 0
                    i += nr
            cache[key] = torch.from_numpy(tab.view(np.uint8)).to(self.rows.device)
        return cache[key]


def _radam_step_size(step, beta1, beta2, thresh):
    """(N_sma > threshold, step_size) of ranger2020.py:194-212 (host scalars, cached per step by the reference)."""
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > thresh:
        ss = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (1 - beta1 ** step)
        return True, ss
    return False, 1.0 / (1 - beta1 ** step)


class Ranger(Optimizer):
    """RAdam + Lookahead + gradient centralisation; arguments and defaults of ranger2020.py:45-58."""

    def __init__(self, params, lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(0.95, 0.999), eps=1e-5,
                 weight_decay=0, use_gc=True, gc_conv_only=False, gc_loc=True):
        if not 0.0 <= alpha <= 1.0:
            raise ValueError(f"Invalid slow update rate: {alpha}")
        if not 1 <= k:
            raise ValueError(f"Invalid lookahead steps: {k}")
        if not lr > 0:
            raise ValueError(f"Invalid Learning Rate: {lr}")
        if not eps > 0:
            raise ValueError(f"Invalid eps: {eps}")
        defaults = dict(lr=lr, alpha=alpha, k=k, step_counter=0, betas=betas, N_sma_threshhold=N_sma_threshhold,
                        eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.N_sma_threshhold = N_sma_threshhold
        self.alpha, self.k = alpha, k
        self.gc_loc, self.use_gc, self.gc_conv_only = gc_loc, use_gc, gc_conv_only
        self._flat = [_FlatGroup(g["params"]) for g in self.param_groups]
        self._gnorm_sq = None            # device scalar written by clip_grad_norm_, consumed by the next step()
        self._max_norm = 0.0
        for fg in self._flat:
            for p, o in zip(fg.params, fg.offsets):
                self.state[p] = dict(step=0, exp_avg=fg.view(fg.flat_m, p, o), exp_avg_sq=fg.view(fg.flat_v, p, o),
                                     slow_buffer=fg.view(fg.flat_s, p, o))

    # -- gradients -------------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        """zero the flat gradient buffers (one memset per group); the .grad views stay in place."""
        for fg in self._flat:
            fg.flat_g.zero_()
            for p, o in zip(fg.params, fg.offsets):
                if p.grad is None or p.grad.data_ptr() != fg.flat_g.data_ptr() + 4 * o:
                    p.grad = fg.view(fg.flat_g, p, o)

    def _collect_grads(self, fg):
        """a backward that REPLACED .grad (first accumulation into a None grad) is copied back into the flat buffer."""
        for p, o in zip(fg.params, fg.offsets):
            if p.grad is None:
                raise HspError("Ranger: every parameter of a group needs a gradient (fused step)")
            if p.grad.data_ptr() != fg.flat_g.data_ptr() + 4 * o:
                v = fg.view(fg.flat_g, p, o)
                v.copy_(p.grad)
                p.grad = v

    def sync_grads(self):
        """make the flat gradient buffers current: a backward that replaced a ``.grad`` (first accumulation into a None
        grad) is copied back into its flat view.  Call before reading ``flat_g`` directly (data-parallel all-reduce)."""
        for fg in self._flat:
            self._collect_grads(fg)

    def scale_grads_by_clip_(self):
        """apply the pending clip coefficient to the gradients IN PLACE now (instead of inside the next step()): what the
        reference's clip_grad_norm_ does on iterations that accumulate without stepping (engine/train.py:103-104).
        Device-side, no host synchronisation."""
        if self._gnorm_sq is None:
            return
        coef = torch.clamp(self._max_norm / (self._gnorm_sq.sqrt() + 1e-6), max=1.0)
        for fg in self._flat:
            fg.flat_g.mul_(coef)
        self._gnorm_sq = None

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_(all parameters, max_norm) (engine/train.py:99): the squared norm is reduced
        on the device and the coefficient min(1, max_norm / (norm + 1e-6)) is applied INSIDE the next step() --
        no host synchronisation.  Returns the (device) total norm."""
        L = lib()
        parts = []
        for fg in self._flat:
            self._collect_grads(fg)
            out = torch.empty(1, dtype=torch.float32, device=fg.flat_g.device)
            wsb = L.hsp_sumsq_workspace_bytes(fg.total)
            ws = ops._ws(wsb, fg.flat_g.device)
            ops._run("hsp_sumsq_f32", (ops._p(fg.flat_g), fg.total, ops._p(out), ops._p(ws), wsb, ops._stream()),
                     key=f"n{fg.total}", abytes=4 * fg.total)
            parts.append(out)
        self._gnorm_sq = parts[0] if len(parts) == 1 else torch.stack(parts).sum(dim=0)
        self._max_norm = float(max_norm)
        return self._gnorm_sq.sqrt()

    # -- the step --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        for group, fg in zip(self.param_groups, self._flat):
            self._collect_grads(fg)
            if not fg.slow_ready:                                  # first step of this group: slow weights = weights now
                fg.flat_s.copy_(fg.flat_p)
                fg.slow_ready = True
            st0 = self.state[fg.params[0]]
            step = int(st0["step"]) + 1
            beta1, beta2 = group["betas"]
            adaptive, step_size = _radam_step_size(step, beta1, beta2, self.N_sma_threshhold)
            look = step % group["k"] == 0
            rows = fg.rows_for(self.use_gc, self.gc_conv_only)
            ops._run("hsp_ranger_step",
                     (ops._p(fg.flat_p), ops._p(fg.flat_g), ops._p(fg.flat_m), ops._p(fg.flat_v), ops._p(fg.flat_s),
                      ops._p(rows), fg.nrows, beta1, beta2, group["eps"], group["weight_decay"],
                      step_size * group["lr"], int(adaptive), int(look), self.alpha, int(not self.gc_loc),
                      ops._p(self._gnorm_sq), self._max_norm, ops._stream()),
                     key=f"n{fg.total}", abytes=fg.total * (44 + (8 if look else 0)))
            for p in fg.params:
                self.state[p]["step"] = step
        self._gnorm_sq = None
        return loss

    # -- checkpoints -----------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """accepts checkpoints of the reference Ranger (per-parameter tensors): values are copied into the flat state."""
        super().load_state_dict(state_dict)
        for fg in self._flat:
            # a checkpoint written after >= 1 step carries every slow_buffer; one without them -- or one this optimizer wrote BEFORE
            # its first step, whose slow_buffer views exist but still hold zeros -- leaves the lazy fill armed
            fg.slow_ready = all("slow_buffer" in self.state[p] and int(self.state[p].get("step", 0)) > 0 for p in fg.params)
            for p, o in zip(fg.params, fg.offsets):
                st = self.state[p]
                for name, flat in (("exp_avg", fg.flat_m), ("exp_avg_sq", fg.flat_v), ("slow_buffer", fg.flat_s)):
                    v = fg.view(flat, p, o)
                    if name in st:
                        v.copy_(st[name])
                    st[name] = v
                st["step"] = int(st.get("step", 0))


def clip_grad_norm_(optimizer, max_norm):
    """drop-in for ``torch.nn.utils.clip_grad_norm_(network.parameters(), max_norm)`` in the train loop when the
    optimizer is the fused Ranger (the scaling itself happens in ``optimizer.step()``)."""
    return optimizer.clip_grad_norm_(max_norm)


# ------------------------------------------------------------------------------------------------------------
# learning-rate schedule
# ------------------------------------------------------------------------------------------------------------

def flat_and_anneal_factor(x, total_iters, warmup_iters=0, warmup_factor=0.1, warmup_method="linear", anneal_point=0.72,
                           anneal_method="cosine", target_lr_factor=0, poly_power=1.0, step_gamma=0.1,
                           steps=(2 / 3.0, 8 / 9.0)):
    """lr factor at iteration x (lr_scheduler.py:219-261): warm-up, flat, then anneal from anneal_point*total_iters."""
    from bisect import bisect_right
    anneal_start = (steps[0] if anneal_method == "step" else anneal_point) * total_iters
    if x < warmup_iters:
        if warmup_method == "linear":
            a = float(x) / warmup_iters
            return warmup_factor * (1 - a) + a
        return warmup_factor
    if x >= anneal_start:
        if anneal_method == "step":
            return step_gamma ** bisect_right([s * total_iters for s in steps], float(x))
        frac = (float(x) - anneal_start) / (total_iters - anneal_start)
        if anneal_method == "cosine":
            return target_lr_factor + 0.5 * (1 - target_lr_factor) * (1 + math.cos(math.pi * frac))
        if anneal_method == "linear":
            return target_lr_factor + (1 - target_lr_factor) * (total_iters - float(x)) / (total_iters - anneal_start)
        if anneal_method == "poly":
            return target_lr_factor + (1 - target_lr_factor) * ((total_iters - float(x)) / (total_iters - anneal_start)) ** poly_power
        if anneal_method == "exp":
            return max(target_lr_factor, 5e-3) ** frac
        return 1
    return 1


def flat_and_anneal_lr_scheduler(optimizer, total_iters, warmup_iters=0, warmup_factor=0.1, warmup_method="linear",
                                 anneal_point=0.72, anneal_method="cosine", target_lr_factor=0, poly_power=1.0,
                                 step_gamma=0.1, steps=(2 / 3.0, 8 / 9.0)):
    if warmup_method not in ("constant", "linear"):
        raise ValueError("Only 'constant' or 'linear' warmup_method accepted, got {}".format(warmup_method))
    if anneal_method not in ("cosine", "linear", "poly", "exp", "step", "none"):
        raise ValueError("Only 'cosine', 'linear', 'poly', 'exp', 'step' or 'none' anneal_method accepted, got {}".format(anneal_method))
    if anneal_method != "step" and not 0 <= anneal_point <= 1:
        raise ValueError("anneal_point should be in [0,1], got {}".format(anneal_point))
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda x: flat_and_anneal_factor(x, total_iters, warmup_iters, warmup_factor, warmup_method, anneal_point,
                                                    anneal_method, target_lr_factor, poly_power, step_gamma, steps))


def build_optimizer(params):
    """tools/training_utils.py:38-56 with the reference's flag defaults: Ranger(lr=FLAGS.lr, weight_decay=0)."""
    kind = getattr(FLAGS, "optimizer_type", "Ranger")
    if kind != "Ranger":
        raise NotImplementedError(f"optimizer_type {kind}: only Ranger (the reference's default) is built")
    groups = list(params)
    return Ranger(groups, lr=float(getattr(FLAGS, "lr", 1e-4)), weight_decay=0)


def build_lr_rate(optimizer, total_iters):
    """tools/training_utils.py:13-35: flat_and_anneal with the flag defaults (warm-up 1000 iters from 0.001, cosine from 72 %)."""
    name = getattr(FLAGS, "lr_scheduler_name", "flat_and_anneal")
    if name != "flat_and_anneal":
        raise NotImplementedError(f"lr_scheduler_name {name}: only flat_and_anneal (the reference's default) is built")
    return flat_and_anneal_lr_scheduler(
        optimizer, total_iters, warmup_iters=getattr(FLAGS, "warmup_iters", 1000),
        warmup_factor=getattr(FLAGS, "warmup_factor", 0.001), warmup_method=getattr(FLAGS, "warmup_method", "linear"),
        anneal_point=getattr(FLAGS, "anneal_point", 0.72), anneal_method=getattr(FLAGS, "anneal_method", "cosine"),
        target_lr_factor=0, poly_power=getattr(FLAGS, "poly_power", 1.0), step_gamma=getattr(FLAGS, "gamma", 0.1))
