"""On-device point-cloud augmentation of the training step; mirrors ``HSPose.data_augment`` (network/HSPose.py:185-256)
and the batched deformations of ``datasets/data_augmentation.py:70-150``.

Four independent augmentations, each applied to the samples whose uniform draw falls under its probability flag
(``aug_bb_pro, aug_rt_pro, aug_bc_pro, aug_pc_pro``): anisotropic box scaling in the object frame, a rigid
perturbation, a linear "box cage" taper along y for bowls / mugs, and per-point radial jitter.  The random draws are
made in the reference's order on the tensors' device (``torch.rand``), so a seeded run consumes the generator
identically.
"""
import contextlib
import os

import torch

from . import staging

from .config import FLAGS

_noise_feed = None


@contextlib.contextmanager
def jitter_noise_feed(noise):
    """Within the scope ``defor_3D_pc`` takes its per-point jitter factors from ``noise`` (a device tensor the caller
    filled from ``torch.rand(pc.shape) * r`` on the CPU generator, as the reference draws it) instead of drawing: a
    captured hipGraph cannot contain the host draw + upload (hs_pose_amd.graph.GraphedTrainStep)."""
    global _noise_feed
    prev, _noise_feed = _noise_feed, noise
    try:
        yield
    finally:
        _noise_feed = prev


def _object_frame(pc, R, t):
    return torch.matmul(pc - t.unsqueeze(-2), R)                     # R^T (p - t) per point


def _camera_frame(pts, R, t):
    return torch.matmul(pts, R.transpose(-2, -1)) + t.unsqueeze(-2)


def defor_3D_bb_in_batch(pc, model_point, R, t, s, sym=None, aug_bb=None):
    """scale the object along its own axes by aug_bb (x and z share the mean factor when it is rotationally symmetric);
    data_augmentation.py:70-79."""
    sym_bb = (aug_bb + aug_bb.flip(-1)) / 2.0                         # (x, y, z) + (z, y, x); no index upload: graph-capturable
    k = torch.where((sym[:, 0] == 1).unsqueeze(-1), sym_bb, aug_bb)
    pc_new = _camera_frame(_object_frame(pc, R, t) * k.unsqueeze(-2), R, t)
    return pc_new, s * k, model_point * k.unsqueeze(-2)


def defor_3D_rt_in_batch(pc, R, t, aug_rt_t, aug_rt_r):
    """shift by aug_rt_t, then rotate everything by aug_rt_r about the camera origin; data_augmentation.py:183-190."""
    pc_new = torch.matmul(pc + aug_rt_t.unsqueeze(-2), aug_rt_r.transpose(-2, -1))
    return pc_new, torch.matmul(aug_rt_r, R), torch.matmul(aug_rt_r, (t + aug_rt_t).unsqueeze(-1)).squeeze(-1)


def defor_3D_bc_in_batch(pc, R, t, s, model_point, nocs_scale):
    """taper: x and z scaled linearly in y between ey_down (bottom) and ey_up (top), both drawn in [0.8, 1.2); the new
    size comes from the tapered model; data_augmentation.py:108-129."""
    bs = pc.size(0)
    ey_up = torch.rand((bs, 1), device=pc.device) * (1.2 - 0.8) + 0.8
    ey_down = torch.rand((bs, 1), device=pc.device) * (1.2 - 0.8) + 0.8
    s_y = s[..., 1].unsqueeze(-1)

    def taper(pts):
        f = (pts[..., 1] + s_y / 2.0) / s_y * (ey_up - ey_down) + ey_down
        return torch.stack([pts[..., 0] * f, pts[..., 1], pts[..., 2] * f], dim=-1)
    pc_new = _camera_frame(taper(_object_frame(pc, R, t)), R, t)
    m = taper(model_point)
    s_new = (torch.max(m, dim=1)[0] - torch.min(m, dim=1)[0]) * nocs_scale.unsqueeze(-1)
    return pc_new, s_new, ey_up, ey_down


def _host_rand(like):
    """``torch.rand(shape).to(device)`` of the reference (CPU default generator), uploaded through the pinned staging ring
    (hs_pose_amd/staging.py) so the host does not wait for the previous training step."""
    if not like.is_cuda:
        return torch.rand(like.shape)
    return staging.upload(lambda buf: torch.rand(like.shape, out=buf), like.shape, torch.float32, like.device)


def defor_3D_pc(pc, gt_t, r=0.2, points_defor=None, return_defor=False):
    """every coordinate moves away from the object centre by a uniform fraction in [0, r); data_augmentation.py:137-144
    (the draw is made on the CPU generator and moved, like the reference's ``torch.rand(shape).to(device)``)."""
    if points_defor is None:
        points_defor = _noise_feed if _noise_feed is not None else _host_rand(pc) * r
    new_pc = pc + points_defor * (pc - gt_t.unsqueeze(1))
    return (new_pc, points_defor) if return_defor else new_pc


FUSED = True      # device batches: one libhsp launch (hsp_pose_augment); False: the torch composition above (tests)


def _data_augment_fused(PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t, aug_rt_r, model_point, nocs_scale, obj_ids):
    """the same augmentation through ``hsp_pose_augment`` (csrc/losses.hip): the six uniform draws are made here, on the
    device generator and in the composition's order (flag, flag, flag, ey_up, ey_down, flag; then the jitter noise on the
    CPU generator), so a seeded run consumes both generators exactly as the torch composition below does."""
    from . import ops
    bs, N, _ = PC.shape
    dev = PC.device
    draws = torch.cat([torch.rand((bs, 1), device=dev) for _ in range(6)], dim=1).t().contiguous()      # (6, bs)
    noise = _noise_feed if _noise_feed is not None else _host_rand(PC) * FLAGS.aug_pc_r
    f = lambda t, shape: ops._req(t.detach().float().reshape(shape), torch.float32, "data_augment")
    M = model_point.shape[1]
    out_pc = torch.empty(bs, N, 3, dtype=torch.float32, device=dev)
    out_R = torch.empty(bs, 3, 3, dtype=torch.float32, device=dev)
    out_t = torch.empty(bs, 3, dtype=torch.float32, device=dev)
    out_s = torch.empty(bs, 3, dtype=torch.float32, device=dev)
    args = [f(PC, (bs, N, 3)), f(gt_R, (bs, 3, 3)), f(gt_t, (bs, 3)), f(gt_s, (bs, 3)), f(mean_shape, (bs, 3)), f(sym, (bs, 4)),
            f(aug_bb, (bs, 3)), f(aug_rt_t, (bs, 3)), f(aug_rt_r, (bs, 3, 3)), f(model_point, (bs, M, 3)), f(nocs_scale, (bs,)),
            f(obj_ids, (bs,)), draws, f(noise, (bs, N, 3))]
    ops._run("hsp_pose_augment", [ops._p(a) for a in args] + [bs, N, M, float(FLAGS.aug_bb_pro), float(FLAGS.aug_rt_pro),
                                                            float(FLAGS.aug_bc_pro), float(FLAGS.aug_pc_pro), ops._p(out_pc),
                                                            ops._p(out_R), ops._p(out_t), ops._p(out_s), ops._stream()])
    return out_pc, out_R, out_t, out_s


def data_augment(PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t, aug_rt_r, model_point, nocs_scale, obj_ids):
    """HSPose.data_augment (HSPose.py:185-256): -> (PC, gt_R, gt_t, gt_s) with each augmentation applied where its
    draw says so.  gt_s is the size RESIDUAL to mean_shape, as everywhere in the network."""
    if FUSED and PC.is_cuda:
        return _data_augment_fused(PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t, aug_rt_r, model_point, nocs_scale,
                                   obj_ids)
    bs = PC.shape[0]

    flag = torch.rand((bs, 1), device=PC.device) < FLAGS.aug_bb_pro
    pc_new, s_new, model_new = defor_3D_bb_in_batch(PC, model_point, gt_R, gt_t, gt_s + mean_shape, sym, aug_bb)
    PC = torch.where(flag.unsqueeze(-1), pc_new, PC)
    gt_s = torch.where(flag, s_new - mean_shape, gt_s)
    model_point = torch.where(flag.unsqueeze(-1), model_new, model_point)

    flag = torch.rand((bs, 1), device=PC.device) < FLAGS.aug_rt_pro
    pc_new, R_new, t_new = defor_3D_rt_in_batch(PC, gt_R, gt_t, aug_rt_t, aug_rt_r)
    PC = torch.where(flag.unsqueeze(-1), pc_new, PC)
    gt_R = torch.where(flag.unsqueeze(-1), R_new, gt_R)
    gt_t = torch.where(flag, t_new, gt_t)

    # box-cage taper only for mug (5) and bowl (1)
    flag = torch.logical_and(torch.rand((bs, 1), device=PC.device) < FLAGS.aug_bc_pro,
                             torch.logical_or(obj_ids == 5, obj_ids == 1).unsqueeze(-1))
    pc_new, s_new, _, _ = defor_3D_bc_in_batch(PC, gt_R, gt_t, gt_s + mean_shape, model_point, nocs_scale)
    PC = torch.where(flag.unsqueeze(-1), pc_new, PC)
    gt_s = torch.where(flag, s_new - mean_shape, gt_s)

    flag = torch.rand((bs, 1), device=PC.device) < FLAGS.aug_pc_pro
    pc_new, _ = defor_3D_pc(PC, gt_t, FLAGS.aug_pc_r, return_defor=True)
    PC = torch.where(flag.unsqueeze(-1), pc_new, PC)
    return PC, gt_R, gt_t, gt_s
