"""hs_pose_amd -- MI355X (gfx950) native hybrid-scope point-cloud feature extractor for HS-Pose.

Only the hot path of the reference (KNN -> receptive-field graph-conv stack -> pool / up-sample ->
per-point heads, plus Chamfer / FPS kernels) lives here, behind the reference's own Python operator
surface:

    hs_pose_amd.gcn3d       <-> network/fs_net_repo/gcn3d.py
    hs_pose_amd.FaceRecon   <-> network/fs_net_repo/FaceRecon.py
    hs_pose_amd.PoseR/PoseTs/PoseNet9D <-> network/fs_net_repo/{PoseR,PoseTs,PoseNet9D}.py
    hs_pose_amd.chamfer     <-> tools/pyTorchChamferDistance/chamfer_distance.py
    hs_pose_amd.config.FLAGS<-> the absl FLAGS object the reference modules read

Device work goes through the C-ABI shared library libhsp.so (include/hsp.h).  There is no CPU or
eager-PyTorch fallback: ops raise if the library is missing or a tensor is not on the GPU.
"""
from .config import FLAGS  # noqa: F401

__version__ = "0.1.0"
