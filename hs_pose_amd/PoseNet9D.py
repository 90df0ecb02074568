"""Host-side mirror of the reference's ``network/fs_net_repo/PoseNet9D.py`` (class name, sub-module names and the
10-tuple it returns are the interface ``HSPose.forward`` relies on, HSPose.py:64-65)."""
import os

import torch
import torch.nn as nn

from . import ops
from .config import FLAGS
from .FaceRecon import FaceRecon
from .PoseR import Rot_green, Rot_red
from .PoseTs import Pose_Ts


FUSED_FACE_SPLIT = True      # (tests flip it to compare the one-launch forms with the torch compositions below)


def _axis_and_confidence(head_out):
    """(B,4) head output -> (unit axis (B,3) from columns 1:4 with the reference's 1e-6 guard, sigmoid of column 0)
    (PoseNet9D.py:40-46)"""
    if (head_out.is_cuda and head_out.dtype == torch.float32 and head_out.dim() == 2 and head_out.shape[1] == 4
            and FUSED_FACE_SPLIT):
        return ops.axis_conf(head_out)                  # one launch each way instead of 4 forward + ~14 in autograd's backward
    v = head_out[:, 1:]
    return v / (v.norm(dim=1, keepdim=True) + 1e-6), head_out[:, 0].sigmoid()


def _split_face_head(face, n_faces=6):
    """(B,N,30) face-head output -> per-face unit normals (B,N,6,3), distances (B,N,6), confidences (B,N,6)
    (PoseNet9D.py:31-35)"""
    b, n, c = face.shape
    if face.is_cuda and face.dtype == torch.float32 and n_faces == 6 and c == 30 and FUSED_FACE_SPLIT:
        return ops.face_split(face)                     # one launch each way instead of ~8 forward + ~20 in autograd's backward
    normals = face[..., :3 * n_faces].view(b, n, n_faces, 3)
    return (normals / normals.norm(dim=-1, keepdim=True), face[..., 3 * n_faces:4 * n_faces],
            face[..., 4 * n_faces:].sigmoid())


class PoseNet9D(nn.Module):
    """reference PoseNet9D.py:14-52.  forward(points (B,N,3), obj_id (B,1)) -> recon, face_normal, face_dis, face_f,
    p_green_R, p_red_R, f_green_R, f_red_R, Pred_T, Pred_s (the first four are None unless FLAGS.train)."""

    def __init__(self):
        super().__init__()
        # registration order == the reference's (state_dict order)
        self.rot_green = Rot_green()
        self.rot_red = Rot_red()
        self.face_recon = FaceRecon()
        self.ts = Pose_Ts()

    def forward(self, points, obj_id):
        if self.face_recon._x3 is None:
            self.face_recon._x3 = ops.X3Planes()
        with ops.x3_scope(self.face_recon._x3):          # the heads' weight planes live in the backbone's registry
            return self._forward(points, obj_id)

    def _fan_first_layers(self, rows, xyz):
        """the first Conv1d of every consumer of feat's rows (the three pose heads, PoseR.py:27 / PoseTs.py:32, and -- training --
        the reconstruction block, FaceRecon.py:38) as one ``ops.fan_linear_rows`` node; keeps the heads' results for
        ``_forward`` and returns the reconstruction block's.  None (the caller then runs its own layer) when the shapes /
        GEMM mode are not the fused kernels'."""
        heads = (self.rot_green, self.rot_red, self.ts)
        layers = [(h.conv1.weight.squeeze(-1), h.conv1.bias) for h in heads]
        blk = self.face_recon.conv1d_block[0] if FLAGS.train else None
        if blk is not None:
            layers.append((blk.weight.squeeze(-1), blk.bias))
        if not ops.fan_linear_rows_ok(rows, xyz, [w for w, _ in layers]):
            self._first = None
            return None
        outs = ops.fan_linear_rows(rows, xyz, layers)
        self._first = outs[:3]
        return outs[3] if blk is not None else None

    def _forward(self, points, obj_id):
        if points.is_cuda and points.dtype == torch.float32 and not points.requires_grad:
            # the mean in the reference's summation order (one launch), training included: the coordinate searches then see the
            # reference's bits, so their tie-ridden rows (tiled clouds) come out as the reference's lists in both modes
            local, centre = ops.center_cloud(points)
        else:
            centre = points.mean(dim=1, keepdim=True)
            local = points - centre                              # the network sees clouds centred on their mean
        self._first = None
        self.face_recon.feat_consumers = self._fan_first_layers if FLAGS.train else None
        try:
            recon, face, feat = self.face_recon(local, obj_id)
        finally:
            self.face_recon.feat_consumers = None
        if not FLAGS.train:
            self._fan_first_layers(feat.reshape(-1, feat.shape[-1]), local)
        first, self._first = self._first, None
        fg, fr, ft = first if first is not None else (None, None, None)
        face_normal = face_dis = face_f = None
        if FLAGS.train:                                          # training-only heads (reconstruction, box faces)
            recon = recon + centre
            face_normal, face_dis, face_f = _split_face_head(face)
        else:
            recon = None
        p_green_R, f_green_R = _axis_and_confidence(self.rot_green.forward_rows(feat, fg))
        p_red_R, f_red_R = _axis_and_confidence(self.rot_red.forward_rows(feat, fr))
        # (fused: conv1 of the translation / size head read cat[feat, local] as the two sources of its product)
        shift, size = self.ts.forward_rows(feat if ft is not None else ops.cat_rows_pitched([feat, local]), ft)
        return (recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R,
                shift + centre.squeeze(1), size)
