"""The 19 loss terms of the PoseNet_only stage through libhsp's fused kernels (csrc/losses.hip, include/hsp.h
``hsp_pose_losses_fwd / _bwd``): five launches per step instead of the ~2 200 ATen kernels that the torch-op composition in
``losses.py`` issues for the reference's ``losses/*.py`` (network/HSPose.py:84-160).  Same dictionary layout, keys, weights and
shapes as the four loss modules return, so ``engine/train.py:84-90`` sums them unchanged.

``losses.py`` stays the readable statement of the same formulas (and what the CPU tests pin against the reference's
fixtures); ``tests/test_gpu_fused_losses.py`` holds this path against the same fixtures and against ``losses.py``'s
autograd on random batches.
"""
import ctypes

import torch

from ._lib import HspLossCfg, lib
from .config import FLAGS
from . import ops

TERMS = (("fsnet_loss", ("Rot1", "Rot1_cos", "Rot2", "Rot2_cos", "Rot_r_a", "Tran", "Size", "R_con")),
         ("recon_loss", ("recon_per_p", "recon_p_f", "recon_point_vote", "recon_point_r", "recon_point_t",
                         "recon_point_s", "recon_point_self")),
         ("geo_loss", ("geo_point",)),
         ("prop_loss", ("Prop_pm", "Prop_sym_recon", "Prop_sym_rt")))
N_TERMS = sum(len(k) for _, k in TERMS)

# order of the differentiable inputs / of the gradients _bwd writes
_NET = ("recon", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s")
_GT = ("PC", "gt_R", "gt_t", "gt_s", "mean_shape", "sym", "obj_id")


def _cfg():
    kind = getattr(FLAGS, "fsnet_loss_type", "l1")
    if kind not in ("l1", "smoothl1"):
        raise NotImplementedError(kind)
    c = HspLossCfg()
    for name, _ in HspLossCfg._fields_:
        if name != "smooth_l1":
            setattr(c, name, float(getattr(FLAGS, name)))
    c.smooth_l1 = 1 if kind == "smoothl1" else 0
    return c


def _f32(t, shape, name):
    t = ops._req(t.detach(), torch.float32, "pose_losses." + name)
    if tuple(t.shape) != tuple(shape):
        if t.numel() != int(torch.Size(shape).numel()):
            raise ops.HspError(f"pose_losses.{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        t = t.reshape(shape)
    return t


class LossDict(dict):
    """the reference's loss_dict (four sub-dictionaries of 0-dim terms) that also carries their sum (``total``) and the identity of
    the 19 term tensors it was built with (``terms``): the sum is only valid for exactly those"""
    total = None
    terms = None

    def untouched(self):
        """True while every group still holds exactly the tensors the loss kernels returned (no term dropped, replaced,
        re-weighted or added)"""
        if self.terms is None or set(self.keys()) != {g for g, _ in TERMS}:
            return False
        k = 0
        for group, keys in TERMS:
            d = self[group]
            if len(d) != len(keys):
                return False
            for key in keys:
                if d.get(key) is not self.terms[k]:
                    return False
                k += 1
        return True


def total_loss(loss_dict):
    """sum of every term of a loss_dict, i.e. what engine/train.py:84-90 computes with
    ``sum(fsnet_loss.values()) + sum(recon_loss.values()) + sum(geo_loss.values()) + sum(prop_loss.values())`` -- taken from the
    fused loss kernels' own reduction when the dict came from them (one launch forward, one in backward, instead of ~28)."""
    t = getattr(loss_dict, "total", None)
    if t is not None and loss_dict.untouched():          # an edited dict (an ablation's dropped / re-weighted / extra term): sum it as is
        return t
    return sum(sum(d.values()) for d in (loss_dict['fsnet_loss'], loss_dict['recon_loss'], loss_dict['geo_loss'],
                                         loss_dict['prop_loss']))


class _PoseLosses(torch.autograd.Function):
    """(network outputs..., ground truth...) -> the 19 terms as separate 0-dim tensors (views of one buffer)."""

    @staticmethod
    def forward(ctx, recon, fn, fd, ff, pg, pr, fg, fr, T, s, PC, gt_R, gt_t, gt_s, ms, sym, obj):
        B, N, _ = PC.shape
        args = [_f32(PC, (B, N, 3), "PC"), _f32(gt_R, (B, 3, 3), "gt_R"), _f32(gt_t, (B, 3), "gt_t"),
                _f32(gt_s, (B, 3), "gt_s"), _f32(ms, (B, 3), "mean_shape"), _f32(sym, (B, 4), "sym"),
                _f32(obj, (B,), "obj_id"), _f32(recon, (B, N, 3), "recon"), _f32(fn, (B, N, 6, 3), "face_normal"),
                _f32(fd, (B, N, 6), "face_dis"), _f32(ff, (B, N, 6), "face_f"), _f32(pg, (B, 3), "p_green_R"),
                _f32(pr, (B, 3), "p_red_R"), _f32(fg, (B,), "f_green_R"), _f32(fr, (B,), "f_red_R"),
                _f32(T, (B, 3), "Pred_T"), _f32(s, (B, 3), "Pred_s")]
        cfg = _cfg()
        ws = ops._ws(lib().hsp_pose_losses_workspace_bytes(B), PC.device)
        terms = torch.empty(N_TERMS, dtype=torch.float32, device=PC.device)
        ops._run("hsp_pose_losses_fwd", [ops._p(a) for a in args] + [B, N, ctypes.byref(cfg), ops._p(terms), ops._p(ws),
                                                                    ws.numel(), ops._stream()])
        ctx.args, ctx.cfg, ctx.ws, ctx.dims = args, cfg, ws, (B, N)
        ctx.shapes = [t.shape for t in (recon, fn, fd, ff, pg, pr, fg, fr, T, s)]
        ctx.set_materialize_grads(False)                   # an unused output's gradient arrives as None, not as a zero tensor
        out = tuple(terms[k:k + 1] if k == 2 else terms[k] for k in range(N_TERMS))        # Rot2 keeps the reference's (1,)
        # + their sum (engine/train.py:84-90 adds the 19 terms one by one: 18 launches forward and ten more in backward for
        # what is one reduction over 19 floats; ``total_loss(loss_dict)`` hands this one out)
        return out + (terms.sum(),)

    @staticmethod
    def backward(ctx, *grads):
        B, N = ctx.dims
        dev = ctx.args[0].device
        if all(g is None for g in grads):
            return (None,) * 17
        g_total, grads = grads[N_TERMS], grads[:N_TERMS]
        if all(g is None for g in grads):                       # only the fused total was used: every term's weight is its gradient
            gw = g_total.reshape(1).float().expand(N_TERMS).contiguous()
        else:
            zero = None
            parts = []
            for g in grads:
                if g is None:
                    if zero is None:
                        zero = torch.zeros(1, dtype=torch.float32, device=dev)
                    parts.append(zero)
                else:
                    parts.append(g.reshape(1))
            gw = torch.cat(parts).float().contiguous()
            if g_total is not None:
                gw = gw + g_total.reshape(1).float()
        shapes = [(B, N, 3), (B, N, 6, 3), (B, N, 6), (B, N, 6), (B, 3), (B, 3), (B,), (B,), (B, 3), (B, 3)]
        outs = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]
        scratch = torch.empty(B * 54, dtype=torch.float32, device=dev)
        ops._run("hsp_pose_losses_bwd", [ops._p(a) for a in ctx.args] + [B, N, ctypes.byref(ctx.cfg), ops._p(gw), ops._p(ctx.ws),
                                                                        ctx.ws.numel(), ops._p(scratch)]
                 + [ops._p(o) for o in outs] + [ops._stream()])
        outs = [o.reshape(s) for o, s in zip(outs, ctx.shapes)]
        need = ctx.needs_input_grad
        return tuple(o if need[i] else None for i, o in enumerate(outs)) + (None,) * 7


def pose_losses(net_out, PC, gt_R, gt_t, gt_s, mean_shape, sym, obj_id):
    """net_out: dict with the ten network outputs (HSPose._NET_OUTPUTS); returns the reference's loss_dict
    {'fsnet_loss': {...}, 'recon_loss': {...}, 'geo_loss': {...}, 'prop_loss': {...}}."""
    vals = _PoseLosses.apply(*[net_out[k] for k in _NET], PC, gt_R, gt_t, gt_s, mean_shape, sym, obj_id.reshape(-1).float())
    out, k = LossDict(), 0
    out.total = vals[N_TERMS]
    out.terms = tuple(vals[:N_TERMS])
    for group, keys in TERMS:
        out[group] = {}
        for key in keys:
            out[group][key] = vals[k]
            k += 1
    return out
