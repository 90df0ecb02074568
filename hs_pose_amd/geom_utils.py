"""Pose-matrix assembly after the network; mirror of tools/geom_utils.py:232-244 (generate_RT) with
tools/rot_utils.py:39-100 (to_R_matrices and helpers) folded into one kernel (csrc/frontend.hip)."""
import torch

from . import ops


def generate_RT(R, f, T, mode, sym):
    """mode 'vec': R = [p_green (bs,3), p_red (bs,3)], f = [f_green (bs), f_red (bs)], T (bs,3), sym (bs,>=1)
    -> (bs,4,4) [[R, T],[0,1]]; mode 'gt' (anything else, geom_utils.py:241-242): R is (bs,3,3)."""
    if mode == "vec":
        return ops.generate_rt(R[0], R[1], f[0].reshape(-1), f[1].reshape(-1), T, sym)
    bs = T.shape[0]
    res = torch.eye(4, dtype=T.dtype, device=T.device).unsqueeze(0).repeat(bs, 1, 1)
    res[:, :3, :3] = R
    res[:, :3, 3] = T
    return res


def to_R_matrices(f_g_vec, f_r_vec, p_g_vec, p_r_vec):
    """rot_utils.py:97-100: (bs,3,3) rotation from two predicted axes and their confidences."""
    bs = p_g_vec.shape[0]
    zeros = torch.zeros(bs, 3, device=p_g_vec.device)
    nosym = torch.zeros(bs, 1, device=p_g_vec.device)
    return ops.generate_rt(p_g_vec, p_r_vec, f_g_vec.reshape(-1), f_r_vec.reshape(-1), zeros, nosym)[:, :3, :3]
