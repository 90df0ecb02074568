"""Drop-in mirror of the reference's ``tools/pyTorchChamferDistance/chamfer_distance.py``."""
import torch

from . import ops


class ChamferDistanceFunction:
    """reference chamfer_distance.py:12-55: ``apply(xyz1, xyz2) -> (dist1, dist2)`` (differentiable)."""

    @staticmethod
    def apply(xyz1, xyz2):
        d1, d2, _, _ = ops.chamfer(xyz1, xyz2)
        return d1, d2


class ChamferDistance(torch.nn.Module):
    """reference chamfer_distance.py:58-60"""

    def forward(self, xyz1, xyz2):
        return ChamferDistanceFunction.apply(xyz1, xyz2)
