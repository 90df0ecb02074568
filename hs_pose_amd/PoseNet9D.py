"""Host-side mirror of the reference's ``network/fs_net_repo/PoseNet9D.py`` (class name, sub-module names and the
10-tuple it returns are the interface ``HSPose.forward`` relies on, HSPose.py:64-65)."""
import torch
import torch.nn as nn

from . import ops
from .config import FLAGS
from .FaceRecon import FaceRecon
from .PoseR import Rot_green, Rot_red
from .PoseTs import Pose_Ts


def _axis_and_confidence(head_out):
    """(B,4) head output -> (unit axis (B,3) from columns 1:4 with the reference's 1e-6 guard, sigmoid of column 0)
    (PoseNet9D.py:40-46)"""
    v = head_out[:, 1:]
    return v / (v.norm(dim=1, keepdim=True) + 1e-6), head_out[:, 0].sigmoid()


def _split_face_head(face, n_faces=6):
    """(B,N,30) face-head output -> per-face unit normals (B,N,6,3), distances (B,N,6), confidences (B,N,6)
    (PoseNet9D.py:31-35)"""
    b, n, _ = face.shape
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

    def _forward(self, points, obj_id):
        centre = points.mean(dim=1, keepdim=True)
        local = points - centre                                  # the network sees clouds centred on their mean
        recon, face, feat = self.face_recon(local, obj_id)
        face_normal = face_dis = face_f = None
        if FLAGS.train:                                          # training-only heads (reconstruction, box faces)
            recon = recon + centre
            face_normal, face_dis, face_f = _split_face_head(face)
        else:
            recon = None
        p_green_R, f_green_R = _axis_and_confidence(self.rot_green.forward_rows(feat))
        p_red_R, f_red_R = _axis_and_confidence(self.rot_red.forward_rows(feat))
        shift, size = self.ts.forward_rows(ops.cat_rows_pitched([feat, local]))
        return (recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R,
                shift + centre.squeeze(1), size)
