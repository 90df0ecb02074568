"""The four loss modules of the training step, mirrors of the reference's ``losses/`` package:

  fs_net_loss          losses/fs_net_loss.py:11-235      pose-vector / translation / size / confidence terms
  recon_6face_loss     losses/recon_loss.py:12-649       per-point face normals / distances / confidences and the
                                                         weighted plane fit ("voting") bounding-box terms
  geo_transform_loss   losses/geometry_loss.py:10-150    point re-projection consistency
  prop_rot_loss        losses/prop_loss.py:11-277        point matching under the predicted pose, symmetry terms
  control_loss         engine/organize_loss.py:1-14      which terms a training stage uses

Same constructor / ``forward(name_list, pred_list, gt_list, sym[, obj_ids])`` signatures, the same dictionary keys
and the same weights (``FLAGS``), so ``HSPose.forward(do_loss=True)`` returns what ``engine/train.py:84-90`` sums.
Only the batched code paths the reference actually runs are implemented (its ``*_old`` per-sample loops and the
terms no training stage selects -- 'Recon', 'Geo_face', 'Point_sampling', 'Point_c_reg', 'Prop_r_reg' -- are
not).  Everything here is (B,3)- or (B,N,6,3)-sized device math on torch ops: the per-batch cost is a few hundred
microseconds next to the backbone's 2 ms, which is why SURVEY ranks fusing it behind the hot path.

The symmetry conventions, shared by all four: ``sym[:,0] == 1`` rotational symmetry about y (bottle, bowl, can:
the red / x axis is undetermined), ``sym[:,1] == 1`` reflection across the x-y plane (laptop, mug with handle),
category id 5 = mug (its x faces are skipped).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import FLAGS


def control_loss(Train_stage):
    if Train_stage == 'PoseNet_only':
        return (['Rot1', 'Rot2', 'Rot1_cos', 'Rot2_cos', 'Rot_regular', 'Tran', 'Size', 'R_con'],
                ['Per_point', 'Point_voting'], ['Geo_point'], ['Prop_pm', 'Prop_sym'])
    if Train_stage == 'FSNet_only':
        return ['Rot1', 'Rot2', 'Tran', 'Size', 'Recon'], [], [], []
    raise NotImplementedError


# ------------------------------------------------------------------------------------------------------------
# small shared pieces
# ------------------------------------------------------------------------------------------------------------

def _dot(a, b):
    return (a * b).sum(dim=-1)


def _to_object_frame(points, R, t):
    """R^T (p - t) for every point: (B,N,3)."""
    return torch.matmul(points - t.unsqueeze(1), R)


_CONSTS = {}


def _const(name, ref, build):
    """small constant tensors, built once per (device, dtype) -- outside any graph capture (first eager call)"""
    key = (name, ref.device, ref.dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = build().to(device=ref.device, dtype=ref.dtype)
    return t


def _skew_basis():
    b = torch.zeros(3, 3, 3)                      # [k]x = sum_a axis[a] * b[a]:  [[0,-z,y],[z,0,-x],[-y,x,0]]
    b[2, 0, 1], b[1, 0, 2], b[2, 1, 0], b[0, 1, 2], b[1, 2, 0], b[0, 2, 1] = -1, 1, 1, -1, -1, 1
    return b.reshape(3, 9)


def _rodrigues(axis, s, c):
    """rotation matrices (B,3,3) about unit axes (B,3) with sin / cos (B,1) (tools/rot_utils.py:67-75):
    R = (1-c) k k^T + c I + s [k]x, assembled from three (B,3,3) terms instead of nine scalar expressions.  Entry by
    entry this is the reference's arithmetic -- (x*y)*(1-c) (+ c | - z*s | + y*s), the other terms being exact zeros --
    in 8 kernels instead of ~75."""
    t = (1 - c).unsqueeze(-1)
    outer = axis.unsqueeze(-1) * axis.unsqueeze(-2)
    K = torch.matmul(axis, _const("skew", axis, _skew_basis)).view(-1, 3, 3)
    return (outer * t + _const("eye3", axis, lambda: torch.eye(3)) * c.unsqueeze(-1)) + K * s.unsqueeze(-1)


def vertical_axes(c1, c2, y, z):
    """get_vertical_rot_vec_in_batch (tools/rot_utils.py:39-65): turn the two predicted axes about their common normal
    by confidence-weighted shares of (angle - 90 deg) so that they become perpendicular."""
    c1, c2 = c1.unsqueeze(-1), c2.unsqueeze(-1)
    axis = torch.cross(y, z, dim=-1)
    axis = axis / (torch.norm(axis, dim=-1, keepdim=True) + 1e-8)
    theta = torch.acos(torch.clamp(_dot(y, z).unsqueeze(-1), -1 + 1e-6, 1 - 1e-6))
    excess = theta - math.pi / 2
    th_y = c2 / (c1 + c2) * excess
    th_z = c1 / (c1 + c2) * excess
    new_y = torch.matmul(_rodrigues(axis, torch.sin(th_y), torch.cos(th_y)), y.unsqueeze(-1)).squeeze(-1)
    new_z = torch.matmul(_rodrigues(axis, torch.sin(-th_z), torch.cos(-th_z)), z.unsqueeze(-1)).squeeze(-1)
    return new_y, new_z


_va_memo = None


def vertical_axes_shared(c1, c2, y, z):
    """vertical_axes for the call the property loss and the voting loss both make on the same network outputs
    (prop_loss.py:163, recon_loss.py:640): computed once per forward, the second caller gets the same autograd nodes.
    The entry is keyed on the identity of the axis tensors (kept referenced, so an id cannot be reused) and the storage of
    the detached confidences; a new forward replaces it."""
    global _va_memo
    key = (id(y), id(z), c1.data_ptr(), c2.data_ptr(), y._version, z._version, torch.is_grad_enabled())
    if _va_memo is not None and _va_memo[0] == key:
        return _va_memo[2]
    out = vertical_axes(c1, c2, y, z)
    _va_memo = (key, (y, z, c1, c2), out)
    return out


def rot_mat_y_first(y, x):
    """get_rot_mat_y_first (tools/rot_utils.py:77-86): orthonormal frame with y kept, columns (x, y, z)."""
    y = F.normalize(y, p=2, dim=-1)
    z = F.normalize(torch.cross(x, y, dim=-1), p=2, dim=-1)
    x = torch.cross(y, z, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def _rescale(res, kept, batch):
    """res * batch / kept when any sample is kept, else res -- the reference's ``if valid_num > 0`` (a host round trip
    there) as a select on the device."""
    return res * torch.where(kept > 0, batch / kept.clamp(min=1).to(res.dtype), torch.ones_like(res))


def _masked_mean_rescaled(values, keep):
    """mean over the batch of ``values`` with the dropped samples zeroed, rescaled by B / #kept when any is kept
    (the reference's way of averaging over the non-symmetric samples only)."""
    res = torch.where(keep, values, torch.zeros_like(values)).mean()
    return _rescale(res, keep.sum(), values.size(0))


def _inverse3(m):
    """inverse of (...,3,3) matrices.  On the GPU by cofactors (nine fused element-wise kernels): torch.inverse goes
    through a batched LU in the solver library with a host synchronisation -- milliseconds for 48 matrices -- and so
    does its backward.  On the CPU torch.inverse (the reference's call; keeps the CPU parity tests at 1e-6)."""
    if not m.is_cuda:
        return torch.inverse(m)
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    A, B, C = e * i - f * h, c * h - b * i, b * f - c * e
    D, E, Fc = f * g - d * i, a * i - c * g, c * d - a * f
    G, H, I = d * h - e * g, b * g - a * h, a * e - b * d
    det = a * A + b * D + c * G
    adj = torch.stack([torch.stack([A, B, C], dim=-1), torch.stack([D, E, Fc], dim=-1), torch.stack([G, H, I], dim=-1)], dim=-2)
    return adj / det.unsqueeze(-1).unsqueeze(-1)


# ------------------------------------------------------------------------------------------------------------
# fs_net_loss
# ------------------------------------------------------------------------------------------------------------

class fs_net_loss(nn.Module):
    def __init__(self):
        super(fs_net_loss, self).__init__()
        kind = getattr(FLAGS, "fsnet_loss_type", "l1")
        if kind == 'l1':
            self.loss_func_t = self.loss_func_s = self.loss_func_Rot1 = self.loss_func_Rot2 = nn.L1Loss()
            self.loss_func_r_con = self.loss_func_Recon = nn.L1Loss()
        elif kind == 'smoothl1':
            self.loss_func_t = self.loss_func_s = self.loss_func_Rot1 = self.loss_func_Rot2 = nn.SmoothL1Loss(beta=0.5)
            self.loss_func_r_con = nn.SmoothL1Loss(beta=0.5)
            self.loss_func_Recon = nn.SmoothL1Loss(beta=0.3)
        else:
            raise NotImplementedError

    def forward(self, name_list, pred_list, gt_list, sym):
        out = {}
        nonsym = sym[:, 0] == 0
        if "Rot1" in name_list:
            out["Rot1"] = FLAGS.rot_1_w * self.loss_func_Rot1(pred_list["Rot1"], gt_list["Rot1"])
        if "Rot1_cos" in name_list:
            out["Rot1_cos"] = FLAGS.rot_1_w * ((1.0 - _dot(pred_list["Rot1"], gt_list["Rot1"])) * 2.0).mean()
        if "Rot2" in name_list:
            out["Rot2"] = FLAGS.rot_2_w * self.cal_loss_Rot2(pred_list["Rot2"], gt_list["Rot2"], sym)
        if "Rot2_cos" in name_list:
            out["Rot2_cos"] = FLAGS.rot_2_w * _masked_mean_rescaled(
                (1.0 - _dot(pred_list["Rot2"], gt_list["Rot2"])) * 2.0, nonsym)
        if "Rot_regular" in name_list:
            out["Rot_r_a"] = FLAGS.rot_regular * _masked_mean_rescaled(
                torch.abs(_dot(pred_list["Rot1"], pred_list["Rot2"])), nonsym)
        if "Recon" in name_list:
            raise NotImplementedError("fs_net_loss 'Recon' (FSNet_only stage): the reference itself exits there "
                                      "(fs_net_loss.py:55-61)")
        if "Tran" in name_list:
            out["Tran"] = FLAGS.tran_w * self.loss_func_t(pred_list["Tran"], gt_list["Tran"])
        if "Size" in name_list:
            out["Size"] = FLAGS.size_w * self.loss_func_s(pred_list["Size"], gt_list["Size"])
        if "R_con" in name_list:
            out["R_con"] = FLAGS.r_con_w * self.cal_loss_R_con(pred_list["Rot1"], pred_list["Rot2"], gt_list["Rot1"],
                                                               gt_list["Rot2"], pred_list["Rot1_f"], pred_list["Rot2_f"], sym)
        return out

    def cal_loss_Rot2(self, pred_v, gt_v, sym):
        """red-axis regression over the non-symmetric samples (fs_net_loss.py:146-154; the result keeps the
        reference's shape (1,))."""
        keep = (sym[:, 0] == 0).unsqueeze(-1)
        res = self.loss_func_Rot2(torch.where(keep, pred_v, torch.zeros_like(pred_v)),
                                  torch.where(keep, gt_v, torch.zeros_like(gt_v)))
        return _rescale(res, keep.sum(dim=0), pred_v.size(0))

    def cal_loss_R_con(self, p_rot_g, p_rot_r, g_rot_g, g_rot_r, p_g_con, p_r_con, sym):
        """the confidences should equal exp(-13.7 |axis error|^2); the red one only where the axis is defined
        (fs_net_loss.py:99-114: plain batch mean, no rescaling)."""
        def target(p, g):
            d = torch.norm(p - g, dim=-1)
            return torch.exp(-13.7 * d * d)
        res_g = self.loss_func_r_con(target(p_rot_g, g_rot_g), p_g_con)
        keep = sym[:, 0] == 0
        tr = target(p_rot_r, g_rot_r)
        res_r = self.loss_func_r_con(torch.where(keep, tr, torch.zeros_like(tr)),
                                     torch.where(keep, p_r_con, torch.zeros_like(p_r_con)))
        return res_r + res_g


# ------------------------------------------------------------------------------------------------------------
# geo_transform_loss
# ------------------------------------------------------------------------------------------------------------

class geo_transform_loss(nn.Module):
    def __init__(self):
        super(geo_transform_loss, self).__init__()
        self.loss_func = nn.L1Loss()

    def forward(self, name_list, pred_list, gt_list, sym):
        out = {}
        if 'Geo_point' in name_list:
            out['geo_point'] = FLAGS.geo_p_w * self.cal_geo_loss_point(gt_list['Points'], pred_list['Rot1'], pred_list['Rot2'],
                                                                      pred_list['Tran'], gt_list['R'], gt_list['T'], sym)
        if 'Geo_face' in name_list:
            raise NotImplementedError("geo_transform_loss 'Geo_face' is not used by any training stage")
        return out

    def cal_geo_loss_point(self, points, p_rot_g, p_rot_r, p_t, g_R, g_t, sym):
        """the points' y (and, without rotational symmetry, x) coordinate in the object frame must come out the
        same from the predicted axis + translation as from the ground-truth pose (geometry_loss.py:123-150)."""
        canon = _to_object_frame(points, g_R, g_t)
        centred = points - p_t.unsqueeze(1)
        res_y = self.loss_func(_dot(centred, p_rot_g.unsqueeze(1)), canon[:, :, 1])
        keep = sym[:, 0] == 0
        x_pred = torch.where(keep.view(-1, 1), _dot(centred, p_rot_r.unsqueeze(1)), torch.zeros_like(canon[:, :, 0]))
        x_gt = torch.where(keep.view(-1, 1), canon[:, :, 0], torch.zeros_like(canon[:, :, 0]))
        res_x = _rescale(self.loss_func(x_pred, x_gt), keep.sum(), points.size(0))
        return res_y + res_x


# ------------------------------------------------------------------------------------------------------------
# prop_rot_loss
# ------------------------------------------------------------------------------------------------------------

class prop_rot_loss(nn.Module):
    def __init__(self):
        super(prop_rot_loss, self).__init__()
        self.loss_func = nn.L1Loss()

    def forward(self, namelist, pred_list, gt_list, sym):
        out = {}
        if "Prop_pm" in namelist:
            out["Prop_pm"] = FLAGS.prop_pm_w * self.prop_point_matching_loss(
                gt_list['Points'], pred_list['Rot1'], pred_list['Rot1_f'], pred_list['Rot2'], pred_list['Rot2_f'],
                pred_list['Tran'], gt_list['R'], gt_list['T'], sym)
        if "Prop_r_reg" in namelist:
            out["Prop_r_reg"] = FLAGS.prop_r_reg_w * torch.mean(torch.abs(1.0 - (pred_list['Rot1_f'] + pred_list['Rot2_f'])))
        if "Prop_sym" in namelist and (FLAGS.prop_sym_w > 0):
            recon, rt = self.prop_sym_matching_loss(gt_list['Points'], pred_list['Recon'], pred_list['Rot1'], pred_list['Rot2'],
                                                    pred_list['Tran'], gt_list['R'], gt_list['T'], sym)
            out["Prop_sym_recon"] = FLAGS.prop_sym_w * recon
            out["Prop_sym_rt"] = FLAGS.prop_sym_w * rt
        else:
            out["Prop_occ"] = 0.0
        return out

    def prop_point_matching_loss(self, points, p_g_vec, f_g_vec, p_r_vec, f_r_vec, p_t, g_R, g_t, sym):
        """points mapped to the object frame by the PREDICTED pose (axes made perpendicular by their confidences;
        for rotationally symmetric objects the ground-truth x axis stands in for the red one with weight 1e-5)
        against the same under the ground-truth pose (prop_loss.py:156-189)."""
        canon = _to_object_frame(points, g_R, g_t)
        ys, xs = vertical_axes(f_g_vec, torch.full_like(f_g_vec, 1e-5), p_g_vec, g_R[..., 0])
        yn, xn = vertical_axes_shared(f_g_vec, f_r_vec, p_g_vec, p_r_vec)
        symmetric = (sym[:, 0] == 1).unsqueeze(-1)
        p_R = rot_mat_y_first(torch.where(symmetric, ys, yn), torch.where(symmetric, xs, xn))
        return self.loss_func(torch.matmul(points - p_t.unsqueeze(1), p_R), canon)

    def prop_sym_matching_loss(self, PC, PC_re, p_g_vec, p_r_vec, p_t, gt_R, gt_t, sym):
        """(recon, rt) of prop_loss.py:258-276.  Three object classes: rotational symmetry with a mirror plane (can,
        bowl, bottle), x-y mirror plane only (laptop, mug with handle), none (camera); a handle-less mug
        (sym = [1,0,0,0]) is left out of the reconstruction term."""
        rot_sym = sym[:, 0] == 1
        mirrors = sym[:, 1:].sum(dim=-1)
        cls_y = torch.logical_and(rot_sym, mirrors > 0).view(-1, 1, 1)           # 180 deg about y
        cls_yx = torch.logical_and(~rot_sym, sym[:, 1] == 1).view(-1, 1, 1)      # mirror z
        cls_none = torch.logical_and(~rot_sym, sym[:, 1] != 1).view(-1, 1, 1)
        skip = torch.logical_and(rot_sym, mirrors == 0).view(-1, 1, 1)
        zero = torch.zeros_like(PC)
        canon = _to_object_frame(PC, gt_R, gt_t)

        def back(pts):
            return torch.matmul(pts, gt_R.transpose(-2, -1)) + gt_t.unsqueeze(1)
        mirror_z = torch.cat([canon[..., :2], -canon[..., 2:]], dim=-1)                       # (x, y, -z)
        turn_y = torch.cat([-canon[..., 0:1], canon[..., 1:2], -canon[..., 2:]], dim=-1)      # (-x, y, -z)
        target = torch.where(cls_yx, back(mirror_z), zero) + torch.where(cls_y, back(turn_y), zero) \
            + torch.where(cls_none, PC, zero)
        recon = self.loss_func(target, torch.where(skip, torch.zeros_like(PC_re), PC_re))

        # the reconstruction should also be the mirror image of the input under the PREDICTED axes / translation
        centred = PC - p_t.unsqueeze(1)
        along = _dot(centred, p_g_vec.unsqueeze(1)).unsqueeze(-1) * p_g_vec.unsqueeze(1)
        mirrored_y = PC + 2.0 * (along - centred)
        n = torch.cross(p_r_vec, p_g_vec, dim=-1)
        n = n / (torch.norm(n, dim=-1, keepdim=True) + 1e-8)
        dist = -(_dot(PC, n.unsqueeze(1)) - _dot(n, p_t).view(-1, 1))
        mirrored_yx = PC + 2.0 * dist.unsqueeze(-1) * n.unsqueeze(1)
        rt = self.loss_func(torch.where(cls_y, mirrored_y, zero) + torch.where(cls_yx, mirrored_yx, zero),
                            torch.where(cls_yx, PC_re, zero) + torch.where(cls_y, PC_re, zero))
        return recon, rt


# ------------------------------------------------------------------------------------------------------------
# recon_6face_loss
# ------------------------------------------------------------------------------------------------------------

def _reorder_faces(x):
    """network order (y+, x+, z+, x-, z-, y-) -> (x+, y+, z+, x-, y-, z-) along dim 2, i.e. x[:, :, [1, 0, 2, 3, 5, 4]],
    by slices (a list index would upload an index tensor: not capturable in a hipGraph)."""
    return torch.cat([x[:, :, 1:2], x[:, :, 0:1], x[:, :, 2:4], x[:, :, 5:6], x[:, :, 4:5]], dim=2)


def _axis_mask(sym_flag, obj_ids):
    """(B,3) bool: which axis residuals count -- y always, z without rotational symmetry, x without rotational
    symmetry and not for the mug (recon_loss.py:545-553).  Built once per loss call."""
    no_rot = sym_flag == 0
    return torch.stack([torch.logical_and(no_rot, obj_ids != 5), torch.ones_like(no_rot), no_rot], dim=-1)


def _axis_sum(res, mask, xz_only=False):
    """sum over the batch of the per-axis residuals res (B,3) that count (same order as the reference: per-axis sums over
    the batch, then x + y + z)."""
    s = torch.where(mask, res, 0.0).sum(dim=0)
    return s[0] + s[2] if xz_only else s[0] + s[1] + s[2]


def fit_planes(points, weights):
    """confidence-weighted least-squares plane z = a x + b y + c through each point set (tools/plane_utils.py:24-49).
    points (...,N,3), weights (...,N) -> unit normal (...,3), foot point of the origin dn (...,3), signed
    offset c / sqrt(a^2 + b^2 + 1) (...,1).  (The N x N diagonal weight matrix of the reference is never formed.)"""
    A = torch.cat([points[..., :2], torch.ones_like(points[..., :1])], dim=-1)
    At = A.transpose(-1, -2)
    w = weights.unsqueeze(-1)
    X = torch.matmul(_inverse3(torch.matmul(At, w * A)), torch.matmul(At, w * points[..., 2:3]))
    a, b, c = X[..., 0, :], X[..., 1, :], X[..., 2, :]
    norm2 = a * a + b * b + 1.0
    dn = torch.cat([a * c, b * c, -c], dim=-1) / (norm2 + 1e-8)
    return dn / torch.norm(dn, dim=-1, keepdim=True), dn, c / torch.sqrt(norm2)


class recon_6face_loss(nn.Module):
    def __init__(self):
        super(recon_6face_loss, self).__init__()
        self.loss_func = nn.L1Loss()

    def forward(self, name_list, pred_list, gt_list, sym, obj_ids, save_path=None):
        out = {}
        if 'Per_point' in name_list:
            res_normal, res_dis, res_f = self.cal_recon_loss_point(
                gt_list['Points'], pred_list['F_n'], pred_list['F_d'], pred_list['F_c'], gt_list['R'], gt_list['T'],
                gt_list['Size'], gt_list['Mean_shape'], sym, obj_ids)
            out['recon_per_p'] = FLAGS.recon_n_w * res_normal + FLAGS.recon_d_w * res_dis
            out['recon_p_f'] = FLAGS.recon_f_w * res_f
        if 'Point_voting' in name_list:
            vote, r, t, s, self_cal = self.cal_recon_loss_vote(
                gt_list['Points'], pred_list['F_n'], pred_list['F_d'], pred_list['F_c'].detach(), pred_list['Rot1'],
                pred_list['Rot1_f'], pred_list['Rot2'], pred_list['Rot2_f'], pred_list['Tran'], pred_list['Size'],
                gt_list['R'], gt_list['T'], gt_list['Size'], gt_list['Mean_shape'], sym, obj_ids, save_path)
            out['recon_point_vote'] = FLAGS.recon_v_w * vote
            out['recon_point_r'] = FLAGS.recon_bb_r_w * r
            out['recon_point_t'] = FLAGS.recon_bb_t_w * t
            out['recon_point_s'] = FLAGS.recon_bb_s_w * s
            out['recon_point_self'] = FLAGS.recon_bb_self_w * self_cal
        if 'Point_sampling' in name_list or 'Point_c_reg' in name_list:
            raise NotImplementedError("recon_6face_loss 'Point_sampling' / 'Point_c_reg' are not used by any training stage")
        return out

    def cal_recon_loss_point(self, pc, face_normal, face_dis, face_f, gt_R, gt_t, gt_s, mean_shape, sym, obj_ids):
        """per-point supervision of the six box faces (recon_loss.py:464-543): the predicted normal of face +-a must
        be the +-a axis of the ground-truth rotation (1 - cos), the predicted distance must be size_a / 2 -+ the point's
        coordinate a, and the confidence must be exp(-303.5 |n d - n_gt d_gt|^2).  All three are summed over the
        faces that are defined for the object, / 6 / B."""
        bs = pc.shape[0]
        fn = _reorder_faces(face_normal)                        # (B,N,6,3)
        fd = _reorder_faces(face_dis)                           # (B,N,6)
        ff = _reorder_faces(face_f)
        coord = _to_object_frame(pc, gt_R, gt_t)                # (B,N,3)
        half = (gt_s + mean_shape).reshape(-1, 1, 3) / 2.0
        sym_flag = sym[:, 0]
        axis_mask = _axis_mask(sym_flag, obj_ids)
        axes = gt_R.transpose(-1, -2).unsqueeze(1)              # (B,1,3,3): row a = axis a in the camera frame
        res_n = res_d = res_c = 0.0
        for sign, sl in ((1.0, slice(0, 3)), (-1.0, slice(3, 6))):
            n_gt = sign * axes
            d_gt = half - sign * coord
            cos = (fn[:, :, sl] * n_gt).sum(dim=-1)             # (B,N,3)
            r = torch.mean(1.0 - cos, dim=1)                    # (B,3)
            xz = torch.where(sym_flag == 0, r[:, 0] + r[:, 2], torch.zeros_like(r[:, 0]))
            res_n = res_n + r[:, 1].sum() + xz.sum()
            res_d = res_d + _axis_sum(torch.mean(torch.abs(fd[:, :, sl] - d_gt), dim=1), axis_mask)
            err = torch.norm(fn[:, :, sl] * fd[:, :, sl].unsqueeze(-1) - n_gt * d_gt.unsqueeze(-1), dim=-1)
            conf = torch.exp(-303.5 * err * err)
            res_c = res_c + _axis_sum(torch.mean(torch.abs(conf - ff[:, :, sl]), dim=1), axis_mask)
        return res_n / 6 / bs, res_d / 6 / bs, res_c / 6 / bs

    def _voted_planes(self, on_plane, conf, gt_t, gt_R_signed, re_s, axis_mask):
        """fit the three +a (or -a) faces through the points pushed onto them, orient the normals like the ground
        truth, and compare the fitted foot points with the true ones (recon_loss.py:555-577)."""
        n, dn, c = fit_planes(on_plane.transpose(1, 2), conf.transpose(-1, -2))      # (B,3,3), (B,3,3), (B,3,1)
        axes = gt_R_signed.transpose(-1, -2)
        flip = (n * axes).sum(dim=-1, keepdim=True) < 0
        n = torch.where(flip, -n, n)
        c = torch.where(flip, -c, c)
        face_centre = gt_t.unsqueeze(-2) + axes * re_s.unsqueeze(-1) / 2.0
        dn_gt = axes * (-(axes * face_centre).sum(dim=-1, keepdim=True))
        return _axis_sum(torch.mean(torch.abs(dn - dn_gt), dim=-1), axis_mask), n, c

    def cal_recon_loss_vote(self, pc, face_normal, face_dis, face_c, p_rot_g, f_rot_g, p_rot_r, f_rot_r, p_t, p_s,
                            gt_R, gt_t, gt_s, mean_shape, sym, obj_ids, save_path=None):
        """bounding-box "voting" (recon_loss.py:616-649): every point votes a point on each face; planes fitted to
        the votes (weights = detached confidences) give a box whose faces are compared with the ground truth (vote),
        with the predicted rotation (r), translation (t) and size (s), and with itself (opposite faces parallel, x/z
        faces perpendicular to y).  NaNs in the fits propagate as NaN losses, as in the reference."""
        bs = pc.shape[0]
        re_s = gt_s + mean_shape
        pre_s = p_s + mean_shape
        fn = _reorder_faces(face_normal)
        fd = _reorder_faces(face_dis)
        fc = _reorder_faces(face_c)
        votes = pc.unsqueeze(-2) + fd.unsqueeze(-1) * fn                              # (B,N,6,3)
        axis_mask = _axis_mask(sym[:, 0], obj_ids)
        vote_up, n_up, c_up = self._voted_planes(votes[:, :, :3], fc[:, :, :3], gt_t, gt_R, re_s, axis_mask)
        vote_dn, n_dn, c_dn = self._voted_planes(votes[:, :, 3:], fc[:, :, 3:], gt_t, -gt_R, re_s, axis_mask)
        # a NaN in any fitted plane turns all five terms into NaN (recon_loss.py:632-639), which train.py then skips;
        # done as a select: no host round trip per step
        bad = torch.isnan(n_up).any() | torch.isnan(n_dn).any() | torch.isnan(c_up).any() | torch.isnan(c_dn).any()
        poison = torch.where(bad, torch.full_like(vote_up, float('nan')), torch.zeros_like(vote_up))
        res_vote = (vote_dn + vote_up) / 6.0 / bs
        # r: the fitted normals against the frame built from the predicted axes
        new_y, new_x = vertical_axes_shared(f_rot_g, f_rot_r, p_rot_g, p_rot_r)
        frame = torch.stack([new_x, new_y, torch.cross(new_x, new_y, dim=-1)], dim=-2)
        res_r = (_axis_sum(torch.mean(torch.abs(n_up - frame), dim=-1), axis_mask)
                 + _axis_sum(torch.mean(torch.abs(n_dn + frame), dim=-1), axis_mask)) / 6.0 / bs
        # t: the predicted centre is equally far from opposite faces
        dis_up = torch.abs((n_up * p_t.unsqueeze(-2)).sum(dim=-1, keepdim=True) + c_up).squeeze(-1)
        dis_dn = torch.abs((n_dn * p_t.unsqueeze(-2)).sum(dim=-1, keepdim=True) + c_dn).squeeze(-1)
        res_t = _axis_sum(torch.abs(dis_dn - dis_up), axis_mask) / 6.0 / bs
        # s: and half the predicted size away from each
        res_s = (_axis_sum(torch.abs(pre_s / 2.0 - dis_up), axis_mask)
                 + _axis_sum(torch.abs(pre_s / 2.0 - dis_dn), axis_mask)) / 6.0 / bs
        # self-consistency of the fitted box
        parallel = _axis_sum(torch.mean(torch.abs(n_up + n_dn), dim=-1), axis_mask)
        perp_up = _axis_sum(torch.abs((n_up[:, 1:2] * n_up).sum(dim=-1)), axis_mask, xz_only=True)
        perp_dn = _axis_sum(torch.abs((n_dn[:, 1:2] * n_dn).sum(dim=-1)), axis_mask, xz_only=True)
        res_self = (parallel + perp_up + perp_dn) / 6.0 / bs
        return res_vote + poison, res_r + poison, res_t + poison, res_s + poison, res_self + poison
