"""Drop-in mirror of the reference's ``network/fs_net_repo/PoseNet9D.py``."""
import torch
import torch.nn as nn

from .config import FLAGS
from .FaceRecon import FaceRecon
from .PoseR import Rot_green, Rot_red
from .PoseTs import Pose_Ts


class PoseNet9D(nn.Module):
    """reference PoseNet9D.py:14-52.  forward(points (B,N,3), obj_id (B,1)) -> the same 10-tuple:
    recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R, Pred_T, Pred_s."""

    def __init__(self):
        super(PoseNet9D, self).__init__()
        self.rot_green = Rot_green()
        self.rot_red = Rot_red()
        self.face_recon = FaceRecon()
        self.ts = Pose_Ts()

    def forward(self, points, obj_id):
        bs, p_num = points.shape[0], points.shape[1]
        mean = points.mean(dim=1, keepdim=True)
        centred = points - mean
        recon, face, feat = self.face_recon(centred, obj_id)

        if FLAGS.train:
            recon = recon + mean
            face_normal = face[:, :, :18].view(bs, p_num, 6, 3)
            face_normal = face_normal / torch.norm(face_normal, dim=-1, keepdim=True)
            face_dis = face[:, :, 18:24]
            face_f = torch.sigmoid(face[:, :, 24:])
        else:
            face_normal, face_dis, face_f, recon = [None] * 4

        green_R_vec = self.rot_green.forward_rows(feat)
        red_R_vec = self.rot_red.forward_rows(feat)
        p_green_R = green_R_vec[:, 1:] / (torch.norm(green_R_vec[:, 1:], dim=1, keepdim=True) + 1e-6)
        p_red_R = red_R_vec[:, 1:] / (torch.norm(red_R_vec[:, 1:], dim=1, keepdim=True) + 1e-6)
        f_green_R = torch.sigmoid(green_R_vec[:, 0])
        f_red_R = torch.sigmoid(red_R_vec[:, 0])

        T, s = self.ts.forward_rows(torch.cat([feat, centred], dim=2))
        Pred_T = T + mean.squeeze(1)
        Pred_s = s
        return recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R, Pred_T, Pred_s
