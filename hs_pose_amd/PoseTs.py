"""Drop-in mirror of the reference's ``network/fs_net_repo/PoseTs.py`` (translation + size head)."""
import torch.nn as nn

from .config import FLAGS
from .PoseR import _PointMLPHead


class Pose_Ts(_PointMLPHead):
    """reference PoseTs.py:12-45: returns (xt (B,3), xs (B,3))."""

    def __init__(self):
        super().__init__(FLAGS.feat_c_ts, FLAGS.Ts_c)
        self.relu1 = nn.ReLU()
        self.relu2 = nn.ReLU()
        self.relu3 = nn.ReLU()

    def forward_rows(self, x, first=None):
        out = super().forward_rows(x, first)
        return out[:, 0:3], out[:, 3:6]
