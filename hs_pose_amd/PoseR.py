"""Drop-in mirror of the reference's ``network/fs_net_repo/PoseR.py`` (rotation heads).

Same parameter names (conv1..4, bn1..3).  The reference feeds (B,C,N) into Conv1d(k=1); here the
trunk runs point-major as (B*N,C) products on the hand-written kernels (``ops.linear_rows``: csrc/gemm_x3.hip, no BLAS
library), which is the layout the HS stack already produces -- ``forward`` still accepts the reference's (B,C,N), ``forward_rows`` takes (B,N,C).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import FLAGS


class _PointMLPHead(nn.Module):
    """Conv1d(f,1024)-BN-ReLU, Conv1d(1024,256)-BN-ReLU, max over points, Conv1d(256,256)-BN-ReLU,
    Dropout(0.2), Conv1d(256,k)   (reference PoseR.py:16-39, PoseTs.py:18-45)."""

    def __init__(self, f, k):
        super().__init__()
        self.f = f
        self.k = k
        self.conv1 = torch.nn.Conv1d(self.f, 1024, 1)
        self.conv2 = torch.nn.Conv1d(1024, 256, 1)
        self.conv3 = torch.nn.Conv1d(256, 256, 1)
        self.conv4 = torch.nn.Conv1d(256, self.k, 1)
        self.drop1 = nn.Dropout(0.2)
        self.bn1 = nn.BatchNorm1d(1024)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(256)

    def forward_rows(self, x: "(B, N, f)", first=None):
        """first: (conv1's output rows (B*N, 1024), BatchNorm first-pass buffer) when the caller computed it with the other
        layers that read the same rows (``ops.fan_linear_rows``, PoseNet9D); x then only gives the shape"""
        b, n, c = x.shape
        if first is None:
            first = ops.linear_rows(x.reshape(b * n, c), self.conv1.weight.squeeze(-1), self.conv1.bias, bn_partials=True)
        # fused BatchNorm + ReLU; the products leave the BatchNorms' first pass (per-tile column sums) in their epilogues
        h = ops.bn_relu(first[0], self.bn1, partial=first[1])
        y2, p2 = ops.linear_rows(h, self.conv2.weight.squeeze(-1), self.conv2.bias, bn_partials=True)
        h = ops.bn_relu(y2, self.bn2, partial=p2)
        h = ops.points_max(h.view(b, n, -1))                                     # (B,256)
        h = ops.bn_relu(ops.linear_rows(h, self.conv3.weight.squeeze(-1), self.conv3.bias), self.bn3)
        h = self.drop1(h)
        return ops.linear_rows(h, self.conv4.weight.squeeze(-1), self.conv4.bias).contiguous()

    def forward(self, x: "(B, f, N)"):
        return self.forward_rows(x.transpose(1, 2))


class Rot_green(_PointMLPHead):
    """reference PoseR.py:10-39"""

    def __init__(self):
        super().__init__(FLAGS.feat_c_R, FLAGS.R_c)


class Rot_red(_PointMLPHead):
    """reference PoseR.py:42-70"""

    def __init__(self):
        super().__init__(FLAGS.feat_c_R, FLAGS.R_c)
