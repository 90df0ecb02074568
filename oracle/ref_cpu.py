"""
oracle/ref_cpu.py -- TEST INFRASTRUCTURE ONLY (CPU oracle).

A functional, torch-fp32, CPU restatement of the HS-Pose hybrid-scope feature extractor, written
from the behaviour of the reference (all paths relative to /root/reference).  It works on a flat
``state`` dict (reference state_dict key names -> tensors) instead of nn.Modules, so the same
tensors can be leaf variables for autograd-derived oracle gradients.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (hs_pose_amd/) never does.

Parity status: PINNED.  oracle/gen_golden.py imports the reference in the build container, checks
every function here against it (bit-equal for indices, exact/allclose(0) for floats where the op
sequence is identical) and writes the fixtures under tests/golden/ that
tests/test_oracle_golden.py re-checks everywhere.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------------------------
# neighbour search                                                   gcn3d.py:15-36
# ---------------------------------------------------------------------------------------------

def knn_index(x: Tensor, k: int) -> Tensor:
    """k nearest rows of ``x`` (B,N,C) to each row, expanded-form fp32 distance, rank 0 dropped.

    Follows gcn3d.py:19-23: ``inner*(-2) + quad[:,None,:] + quad[:,:,None]`` evaluated left to
    right, ``topk(k+1, smallest)`` then ``[:, :, 1:]``.  Returns int64 (B,N,k), ascending distance.
    """
    inner = torch.bmm(x, x.transpose(1, 2))
    quad = (x ** 2).sum(dim=2)
    dist = inner * (-2) + quad.unsqueeze(1) + quad.unsqueeze(2)
    return torch.topk(dist, k + 1, dim=-1, largest=False)[1][:, :, 1:]


def knn_dist(x: Tensor) -> Tensor:
    """The (B,N,N) distance matrix knn_index ranks by (for near-tie diagnostics in tests)."""
    inner = torch.bmm(x, x.transpose(1, 2))
    quad = (x ** 2).sum(dim=2)
    return inner * (-2) + quad.unsqueeze(1) + quad.unsqueeze(2)


def nearest_index(target: Tensor, source: Tensor) -> Tensor:
    """Index (B,Nt,1) of the closest ``source`` row per ``target`` row (gcn3d.py:31-35).

    Note the association differs from knn_index: ``s_norm[j] + t_norm[i] - 2*inner``.
    """
    inner = torch.bmm(target, source.transpose(1, 2))
    s2 = (source ** 2).sum(dim=2)
    t2 = (target ** 2).sum(dim=2)
    d = s2.unsqueeze(1) + t2.unsqueeze(2) - 2 * inner
    return torch.topk(d, 1, dim=-1, largest=False)[1]


def gather_rows(t: Tensor, index: Tensor) -> Tensor:
    """rows of ``t`` (B,Np,D) picked per batch by ``index`` (B,No,n) -> (B,No,n,D)  (gcn3d.py:39-47)."""
    B, Np, D = t.shape
    flat = (index + torch.arange(B, device=t.device).view(B, 1, 1) * Np).reshape(-1)
    return t.reshape(B * Np, D)[flat].view(B, index.shape[1], index.shape[2], D)


def neighbor_dirs(xyz: Tensor, index: Tensor) -> Tensor:
    """unit vectors from each point to its listed neighbours, (B,N,k,3) fp32  (gcn3d.py:49-59)."""
    rel = gather_rows(xyz, index) - xyz.unsqueeze(2)
    return F.normalize(rel, dim=-1).float()


# ---------------------------------------------------------------------------------------------
# HS layers                                                           gcn3d.py:61-218
# ---------------------------------------------------------------------------------------------

def _conv1x1(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    """nn.Conv1d(kernel_size=1) applied to a (B,N,Cin) tensor the way the reference does it:
    transpose -> conv1d -> transpose (gcn3d.py:85,112,149,186)."""
    return F.conv1d(x.transpose(-1, -2), w, b).transpose(-1, -2).contiguous()


def orl_global(feature: Tensor, xyz: Tensor, k: int) -> Tensor:
    """outlier-robust global feature, (B,N,C) constant along N   (gcn3d.py:211-218)."""
    idx = knn_index(xyz, k)
    g = gather_rows(feature, idx).max(dim=2)[0]
    return g.mean(dim=1, keepdim=True).repeat(1, feature.shape[1], 1)


def points_max(feature: Tensor) -> Tensor:
    """(B,N,C) -> (B,C): the heads' max over the points, torch.max(x, 2, keepdim=True)[0] on the reference's (B,C,N)
    layout (PoseR.py:30, :61; PoseTs.py:35; FaceRecon.py:98)."""
    return torch.max(feature.transpose(1, 2), 2, keepdim=True)[0].squeeze(-1)


def orl_forward(feature: Tensor, xyz: Tensor, k: int, conv2_w: Tensor) -> Tensor:
    """conv2(cat[feature, f_global]) + feature     (gcn3d.py:109-113, :183-187)."""
    fg = orl_global(feature, xyz, k)
    return _conv1x1(torch.cat([feature, fg], dim=-1), conv2_w) + feature


def surface_graph_conv(rf: Tensor, directions: Tensor, S: int, K: int) -> Tensor:
    """gcn3d.py:92-107: relu(rf @ normalize(D, dim=0)) -> max over neighbours -> mean over supports."""
    B, N, k, _ = rf.shape
    theta = torch.relu(rf @ F.normalize(directions, dim=0))
    theta = theta.reshape(B, N, k, S, K).max(dim=2)[0]
    return theta.mean(dim=2)


def surface_layer(p: Dict[str, Tensor], prefix: str, xyz: Tensor, k: int, S: int) -> Tensor:
    """HSlayer_surface.forward (gcn3d.py:79-90).  Keys: directions, STE_layer.weight, conv2.weight."""
    D = p[prefix + "directions"]
    K = D.shape[1] // S
    ste = _conv1x1(xyz, p[prefix + "STE_layer.weight"])
    idx = knn_index(xyz, k)                                   # RF-P
    feat = surface_graph_conv(neighbor_dirs(xyz, idx), D, S, K)
    feat = orl_forward(feat, xyz, k, p[prefix + "conv2.weight"])
    return feat + ste


def hs_graph_conv(rf: Tensor, idx: Tensor, x: Tensor, weights: Tensor, bias: Tensor,
                  directions: Tensor, S: int) -> Tensor:
    """HS_layer.graph_conv (gcn3d.py:158-181)."""
    B, N, k, _ = rf.shape
    Cout = directions.shape[1] // S
    theta = torch.relu(rf @ F.normalize(directions, dim=0))            # (B,N,k,S*Cout)
    fm = x @ weights + bias                                           # (B,N,(S+1)*Cout)
    center, support = fm[:, :, :Cout], fm[:, :, Cout:]
    act = theta * gather_rows(support, idx)
    act = act.view(B, N, k, S, Cout).max(dim=2)[0].mean(dim=2)
    return center + act


def hs_layer(p: Dict[str, Tensor], prefix: str, xyz: Tensor, x: Tensor, k: int, S: int,
             return_idx: bool = False):
    """HS_layer.forward (gcn3d.py:143-156): feature-space KNN (RF-F), directions in xyz space."""
    ste = _conv1x1(x, p[prefix + "STE_layer.weight"])
    idx = knn_index(x, k)                                            # RF-F: neighbours in feature space
    rf = neighbor_dirs(xyz, idx)
    feat = hs_graph_conv(rf, idx, x, p[prefix + "weights"], p[prefix + "bias"], p[prefix + "directions"], S)
    feat = orl_forward(feat, xyz, k, p[prefix + "conv2.weight"])
    out = feat + ste
    return (out, idx) if return_idx else out


def pool_layer(xyz: Tensor, x: Tensor, sample_idx: Tensor, k: int = 4) -> Tuple[Tensor, Tensor]:
    """Pool_layer.forward (gcn3d.py:226-246) with the randperm draw passed in explicitly
    (``sample_idx = torch.randperm(N)[:int(N/rate)]``, one draw shared by the whole batch)."""
    idx = knn_index(xyz, k)
    pooled = gather_rows(x, idx).max(dim=2)[0]
    return xyz[:, sample_idx, :], pooled[:, sample_idx, :]


def draw_pool_indices(n_points: int, rate: int = 4, levels: int = 2) -> List[Tensor]:
    """Consume the CPU default generator exactly like FaceRecon's two Pool_layers do
    (gcn3d.py:242-243 called from FaceRecon.py:91 then :96)."""
    out, n = [], n_points
    for _ in range(levels):
        m = int(n / rate)
        out.append(torch.randperm(n)[:m])
        n = m
    return out


# ---------------------------------------------------------------------------------------------
# HS stack wiring                                                     FaceRecon.py:70-128
# ---------------------------------------------------------------------------------------------

def _bn(p, prefix, x, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm1d over the channel dim of a (B,N,C) tensor (FaceRecon.py:90: transpose,bn,transpose)."""
    y = F.batch_norm(x.transpose(1, 2), p.get(prefix + "running_mean"), p.get(prefix + "running_var"),
                     p[prefix + "weight"], p[prefix + "bias"], training, momentum, eps)
    return y.transpose(1, 2)


def _bn_cn(p, prefix, x, training, momentum=0.1, eps=1e-5):
    """BatchNorm1d on a (B,C,N) or (B,C) tensor."""
    return F.batch_norm(x, p.get(prefix + "running_mean"), p.get(prefix + "running_var"),
                        p[prefix + "weight"], p[prefix + "bias"], training, momentum, eps)


def face_recon(p: Dict[str, Tensor], xyz: Tensor, cat_id: Tensor, pool_idx: Sequence[Tensor], *,
               k: int = 20, S: int = 7, obj_c: int = 6, train_heads: bool = False,
               bn_training: bool = True, prefix: str = "") -> Dict[str, Tensor]:
    """FaceRecon.forward (FaceRecon.py:70-128).  ``pool_idx`` = the two randperm slices.

    Returns a dict with feat (B,N,1286) and, when ``train_heads`` (FLAGS.train), recon/face, plus the
    intermediate feature maps (handy for layer-by-layer parity)."""
    B, N, _ = xyz.shape
    one_hot = torch.zeros(B, obj_c, device=xyz.device).scatter_(1, cat_id.view(-1, 1).long(), 1)
    fm0 = torch.relu(surface_layer(p, prefix + "conv_0.", xyz, k, S))
    fm1 = torch.relu(_bn(p, prefix + "bn1.", hs_layer(p, prefix + "conv_1.", xyz, fm0, k, S), bn_training))
    v1, fp1 = pool_layer(xyz, fm1, pool_idx[0])
    k1 = min(k, v1.shape[1] // 8)
    fm2 = torch.relu(_bn(p, prefix + "bn2.", hs_layer(p, prefix + "conv_2.", v1, fp1, k1, S), bn_training))
    fm3 = torch.relu(_bn(p, prefix + "bn3.", hs_layer(p, prefix + "conv_3.", v1, fm2, k1, S), bn_training))
    v2, fp2 = pool_layer(v1, fm3, pool_idx[1])
    k2 = min(k, v2.shape[1] // 8)
    fm4 = hs_layer(p, prefix + "conv_4.", v2, fp2, k2, S)
    f_global = fm4.max(dim=1)[0]
    near1 = nearest_index(xyz, v1)
    near2 = nearest_index(xyz, v2)
    up2 = gather_rows(fm2, near1).squeeze(2)
    up3 = gather_rows(fm3, near1).squeeze(2)
    up4 = gather_rows(fm4, near2).squeeze(2)
    feat = torch.cat([fm0, fm1, up2, up3, up4, one_hot.unsqueeze(1).repeat(1, N, 1)], dim=2)
    out = {"feat": feat, "fm0": fm0, "fm1": fm1, "fm2": fm2, "fm3": fm3, "fm4": fm4,
           "v1": v1, "v2": v2, "near1": near1, "near2": near2, "recon": None, "face": None}
    if train_heads:
        x = feat.permute(0, 2, 1)
        h = _seq_conv_bn_relu(p, prefix + "conv1d_block.", x, 3, bn_training)
        recon = _seq_conv_bn_relu(p, prefix + "recon_head.", h, 1, bn_training)
        recon = F.conv1d(recon, p[prefix + "recon_head.3.weight"], p[prefix + "recon_head.3.bias"])
        face_in = torch.cat([f_global.view(B, -1, 1).repeat(1, 1, N), h, xyz.permute(0, 2, 1)], dim=1)
        face = _seq_conv_bn_relu(p, prefix + "face_head.", face_in, 3, bn_training)
        face = F.conv1d(face, p[prefix + "face_head.9.weight"], p[prefix + "face_head.9.bias"])
        out["recon"], out["face"] = recon.permute(0, 2, 1), face.permute(0, 2, 1)
    return out


def _seq_conv_bn_relu(p, prefix, x, n_blocks, bn_training):
    """n_blocks x (Conv1d, BatchNorm1d, ReLU) laid out as an nn.Sequential with indices 3i,3i+1,3i+2
    (FaceRecon.py:38-66)."""
    for i in range(n_blocks):
        x = F.conv1d(x, p[f"{prefix}{3 * i}.weight"], p[f"{prefix}{3 * i}.bias"])
        x = torch.relu(_bn_cn(p, f"{prefix}{3 * i + 1}.", x, bn_training))
    return x


# ---------------------------------------------------------------------------------------------
# pose heads + PoseNet9D                                    PoseR.py:10-70, PoseTs.py:12-45, PoseNet9D.py:23-52
# ---------------------------------------------------------------------------------------------

def pose_head(p: Dict[str, Tensor], prefix: str, x: Tensor, bn_training: bool, dropout_p: float = 0.0) -> Tensor:
    """Rot_green / Rot_red / Pose_Ts trunk: x (B,Cin,N) -> (B,k).  Dropout(0.2) sits before conv4
    (PoseR.py:33); parity runs use p=0 / eval mode since device RNG cannot match a CPU oracle."""
    x = torch.relu(_bn_cn(p, prefix + "bn1.", F.conv1d(x, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"]), bn_training))
    x = torch.relu(_bn_cn(p, prefix + "bn2.", F.conv1d(x, p[prefix + "conv2.weight"], p[prefix + "conv2.bias"]), bn_training))
    x = x.max(dim=2, keepdim=True)[0]
    x = torch.relu(_bn_cn(p, prefix + "bn3.", F.conv1d(x, p[prefix + "conv3.weight"], p[prefix + "conv3.bias"]), bn_training))
    if dropout_p > 0:
        x = F.dropout(x, dropout_p, training=True)
    x = F.conv1d(x, p[prefix + "conv4.weight"], p[prefix + "conv4.bias"])
    return x.squeeze(2).contiguous()


def posenet9d(p: Dict[str, Tensor], points: Tensor, obj_id: Tensor, pool_idx: Sequence[Tensor], *,
              train_heads: bool, bn_training: bool, k: int = 20, S: int = 7, obj_c: int = 6,
              prefix: str = "") -> Dict[str, Tensor]:
    """PoseNet9D.forward (PoseNet9D.py:23-52); returns the 10 outputs by name."""
    B, N, _ = points.shape
    mean = points.mean(dim=1, keepdim=True)
    fr = face_recon(p, points - mean, obj_id, pool_idx, k=k, S=S, obj_c=obj_c, train_heads=train_heads,
                    bn_training=bn_training, prefix=prefix + "face_recon.")
    feat = fr["feat"]
    out: Dict[str, Optional[Tensor]] = {"feat": feat, "recon": None, "face_normal": None, "face_dis": None, "face_f": None}
    if train_heads:
        face = fr["face"]
        out["recon"] = fr["recon"] + mean
        fn = face[:, :, :18].view(B, N, 6, 3)
        out["face_normal"] = fn / torch.norm(fn, dim=-1, keepdim=True)
        out["face_dis"] = face[:, :, 18:24]
        out["face_f"] = torch.sigmoid(face[:, :, 24:])
    x = feat.permute(0, 2, 1)
    green = pose_head(p, prefix + "rot_green.", x, bn_training)
    red = pose_head(p, prefix + "rot_red.", x, bn_training)
    out["p_green_R"] = green[:, 1:] / (torch.norm(green[:, 1:], dim=1, keepdim=True) + 1e-6)
    out["p_red_R"] = red[:, 1:] / (torch.norm(red[:, 1:], dim=1, keepdim=True) + 1e-6)
    out["f_green_R"] = torch.sigmoid(green[:, 0])
    out["f_red_R"] = torch.sigmoid(red[:, 0])
    ts_in = torch.cat([feat, points - mean], dim=2).permute(0, 2, 1)
    ts = pose_head(p, prefix + "ts.", ts_in, bn_training)
    out["Pred_T"] = ts[:, 0:3] + points.mean(dim=1)
    out["Pred_s"] = ts[:, 3:6]
    return out


# ---------------------------------------------------------------------------------------------
# Chamfer + FPS (float side; index rules live in hsp_oracle.c)
# ---------------------------------------------------------------------------------------------

def chamfer(x1: Tensor, x2: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """ChamferDistanceFunction.forward semantics (chamfer_distance.cpp:59-111): squared distances from
    differences, first minimum wins.  Differentiable (autograd reproduces the +-2g(p-q) backward of
    chamfer_distance.cpp:141-175 because min() routes the gradient to the arg-min pair)."""
    d = ((x1.unsqueeze(2) - x2.unsqueeze(1)) ** 2)
    d = (d[..., 0] + d[..., 1]) + d[..., 2]
    dist1, idx1 = d.min(dim=2)
    dist2, idx2 = d.min(dim=1)
    return dist1, dist2, idx1, idx2


# ---------------------------------------------------------------------------------------------
# inference front / back end                         pc_sample.py:8-77, load_data.py:308-333,
#                                                    geom_utils.py:232-244, rot_utils.py:39-100
# ---------------------------------------------------------------------------------------------

def valid_pixels(mask_hw: Tensor, depth_hw: Tensor) -> Tensor:
    """row-major ids of the pixels with mask * (depth > 0) > 0 (pc_sample.py:38-39, :52; the order of
    boolean-mask indexing)."""
    fuse = mask_hw.reshape(-1).float() * (depth_hw.reshape(-1) > 0.0).float()
    return torch.nonzero(fuse > 0).reshape(-1)


def pc_sample(obj_mask: Tensor, depth: Tensor, camK: Tensor, coor2d: Tensor, samplenum: int, rng) -> Optional[Tensor]:
    """PC_sample with an explicit numpy RandomState-like ``rng`` (the reference uses the global one):
    per image one ``rng.choice(l_all, samplenum, replace=l_all < samplenum)`` (pc_sample.py:57-66);
    None stands for the reference's (None, None) return (:59-60)."""
    if obj_mask.shape[1] == 2:
        obj_mask = torch.max(F.softmax(obj_mask, dim=1), dim=1)[1]
    bs = depth.shape[0]
    out = torch.zeros(bs, samplenum, 3)
    for i in range(bs):
        d = depth[i].reshape(-1)
        u, v = coor2d[i, 0].reshape(-1), coor2d[i, 1].reshape(-1)
        fx, fy, ux, uy = camK[i, 0, 0], camK[i, 1, 1], camK[i, 0, 2], camK[i, 1, 2]
        x = (u - ux) * d / fx
        y = (v - uy) * d / fy
        ids = valid_pixels(obj_mask[i], depth[i])
        l_all = ids.numel()
        if l_all <= 1.0:
            return None
        choose = torch.as_tensor(rng.choice(l_all, samplenum, replace=l_all < samplenum))
        sel = ids[choose]
        out[i] = torch.stack([x[sel], y[sel], d[sel]], dim=1)
    return out / 1000.0


def depth_to_pcl(depth, K, xymap, mask):
    """PoseDataset._depth_to_pcl (load_data.py:322-333) in numpy float64, all valid pixels, fp32 result
    (NOT yet divided by 1000)."""
    import numpy as np
    K = np.asarray(K, dtype=np.float64).reshape(-1)
    cx, cy, fx, fy = K[2], K[5], K[0], K[4]
    d = np.asarray(depth).reshape(-1).astype(np.float64)
    valid = ((d > 0) * np.asarray(mask).reshape(-1)) > 0
    d = d[valid]
    xm = np.asarray(xymap[0]).reshape(-1)[valid].astype(np.float64)
    ym = np.asarray(xymap[1]).reshape(-1)[valid].astype(np.float64)
    return np.stack(((xm - cx) * d / fx, (ym - cy) * d / fy, d), axis=-1).astype(np.float32)


def sample_points(pcl, n_pts: int, rng):
    """PoseDataset._sample_points (load_data.py:308-320) with an explicit ``rng``."""
    import numpy as np
    total = pcl.shape[0]
    if total < n_pts:
        return np.concatenate([np.tile(pcl, (n_pts // total, 1)), pcl[:n_pts % total]], axis=0)
    if total > n_pts:
        return pcl[rng.permutation(total)[:n_pts]]
    return pcl


def _rodrigues(rx: Tensor, s: Tensor, c: Tensor) -> Tensor:
    """to_rot_matrix_in_batch (rot_utils.py:67-75): rotation about unit axis rx, (B,3,3)."""
    x, y, z = rx[:, 0:1], rx[:, 1:2], rx[:, 2:3]
    t = 1 - c
    r1 = torch.cat([x * x * t + c, x * y * t - z * s, x * z * t + y * s], dim=-1)
    r2 = torch.cat([y * x * t + z * s, y * y * t + c, y * z * t - x * s], dim=-1)
    r3 = torch.cat([x * z * t - y * s, z * y * t + x * s, z * z * t + c], dim=-1)
    return torch.stack([r1, r2, r3], dim=-2)


def generate_rt(p_green: Tensor, p_red: Tensor, f_green: Tensor, f_red: Tensor, T: Tensor, sym: Tensor) -> Tensor:
    """generate_RT(mode='vec') (geom_utils.py:232-244): the red confidence is zeroed for symmetric objects,
    both axes are rotated about their common normal by confidence-weighted shares of (angle - pi/2)
    (rot_utils.py:39-65), then orthonormalised y-first (rot_utils.py:77-86)."""
    c1 = f_green.reshape(-1, 1)
    c2 = torch.where(sym[:, 0] == 1, torch.zeros_like(f_red.reshape(-1)), f_red.reshape(-1)).reshape(-1, 1)
    y, z = p_green, p_red
    rx = torch.cross(y, z, dim=-1)
    rx = rx / (torch.norm(rx, dim=-1, keepdim=True) + 1e-8)
    cos = torch.clamp(torch.sum(y * z, dim=-1, keepdim=True), -1 + 1e-6, 1 - 1e-6)
    theta = torch.acos(cos)
    th2 = c1 / (c1 + c2) * (theta - math.pi / 2)
    th1 = c2 / (c1 + c2) * (theta - math.pi / 2)
    ny = torch.matmul(_rodrigues(rx, torch.sin(th1), torch.cos(th1)), y.unsqueeze(-1)).squeeze(-1)
    nz = torch.matmul(_rodrigues(rx, torch.sin(-th2), torch.cos(-th2)), z.unsqueeze(-1)).squeeze(-1)
    yy = F.normalize(ny, p=2, dim=-1)
    zz = F.normalize(torch.cross(nz, yy, dim=-1), p=2, dim=-1)
    xx = torch.cross(yy, zz, dim=-1)
    res = torch.eye(4, dtype=T.dtype).unsqueeze(0).repeat(T.shape[0], 1, 1)
    res[:, :3, :3] = torch.stack((xx, yy, zz), dim=-1)
    res[:, :3, 3] = T
    return res


# ---------------------------------------------------------------------------------------------
# training driver: gradient clipping, Ranger step, learning-rate schedule
#     engine/train.py:96-110, tools/torch_utils/solver/ranger2020.py:135-246, lr_scheduler.py:177-263
# ---------------------------------------------------------------------------------------------

def clip_grads_(grads: Sequence[Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ (L2): total = ||(||g_i||)_i||, g_i *= min(1, max_norm / (total + 1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def ranger_step_(params: Sequence[Tensor], grads: Sequence[Tensor], state: List[Dict[str, Tensor]], step: int, *, lr: float,
                 alpha: float = 0.5, k: int = 6, n_sma_threshold: float = 5, betas=(0.95, 0.999), eps: float = 1e-5,
                 weight_decay: float = 0.0, use_gc: bool = True, gc_conv_only: bool = False, gc_loc: bool = True) -> None:
    """one Ranger.step() for `step` (1-based) on every tensor: gradient centralisation over all dims but the
    first (ranger2020.py:31-41), RAdam moments and rectified step size (:186-212), adaptive or plain update
    (:215-229), Lookahead interpolation every k steps (:232-238).  state[i]: exp_avg, exp_avg_sq, slow_buffer."""
    beta1, beta2 = betas
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma > n_sma_threshold:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) / (1 - beta1 ** step)
    else:
        step_size = 1.0 / (1 - beta1 ** step)

    def central(x):
        if use_gc and x.dim() > (3 if gc_conv_only else 1):
            return x - x.mean(dim=tuple(range(1, x.dim())), keepdim=True)
        return x

    for p, g, st in zip(params, grads, state):
        g = central(g) if gc_loc else g
        st["exp_avg_sq"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        st["exp_avg"].mul_(beta1).add_(g, alpha=1 - beta1)
        if n_sma > n_sma_threshold:
            G = st["exp_avg"] / (st["exp_avg_sq"].sqrt() + eps)
        else:
            G = st["exp_avg"]              # an ALIAS in the reference (:219): the in-place ops below then also
                                           # rewrite exp_avg during the first, non-adaptive steps -- kept as is
        if weight_decay != 0:
            G.add_(p, alpha=weight_decay)
        if not gc_loc:
            G.copy_(central(G))
        p.add_(G, alpha=-step_size * lr)
        if step % k == 0:
            st["slow_buffer"].add_(p - st["slow_buffer"], alpha=alpha)
            p.copy_(st["slow_buffer"])


def flat_and_anneal_factor(x: int, total_iters: int, warmup_iters: int = 1000, warmup_factor: float = 0.001,
                           anneal_point: float = 0.72, target_lr_factor: float = 0.0) -> float:
    """the reference's default schedule (linear warm-up, flat, cosine anneal): lr_scheduler.py:219-261."""
    anneal_start = anneal_point * total_iters
    if x < warmup_iters:
        a = float(x) / warmup_iters
        return warmup_factor * (1 - a) + a
    if x >= anneal_start:
        return target_lr_factor + 0.5 * (1 - target_lr_factor) * (
            1 + math.cos(math.pi * ((float(x) - anneal_start) / (total_iters - anneal_start))))
    return 1


OPT_SHAPES = [(4, 6), (5, 3, 1), (7,), (3, 2, 2, 2), (9, 70), (4200,)]


def opt_case_tensors(step: int):
    """closed-form gradients of optimizer-fixture step `step` (1-based); step 0: the initial parameters."""
    return [hash_tensor(sh, 3000 + 37 * step + i, 1.0 if step else 0.5) for i, sh in enumerate(OPT_SHAPES)]


# ---------------------------------------------------------------------------------------------
# deterministic closed-form parameter fill shared by the golden generator and the tests
# ---------------------------------------------------------------------------------------------

def hash_unit(n: int, seed: int):
    """n reproducible pseudo-random numbers in [0,1) with 24 significant bits (exact in fp32):
    splitmix64 of (index, seed) in wrapping uint64 arithmetic -- no RNG state, no libm, identical on
    every machine.  Returned as a float64 numpy array."""
    import numpy as np
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64((seed * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) % (1 << 64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def hash_fill_(t: Tensor, seed: int, scale: float) -> Tensor:
    """t.flat[i] = (hash_unit(i, seed) - 0.5) * 2 * scale, computed in fp64 and rounded once to fp32."""
    v = (hash_unit(t.numel(), seed) - 0.5) * (2.0 * scale)
    with torch.no_grad():
        t.copy_(torch.from_numpy(v).to(torch.float32).view_as(t))
    return t


def hash_tensor(shape, seed: int, scale: float = 1.0, offset: float = 0.0) -> Tensor:
    """fresh fp32 tensor filled by hash_fill_ (+ offset)."""
    t = torch.empty(*shape, dtype=torch.float32)
    hash_fill_(t, seed, scale)
    if offset:
        t += offset
    return t


def fill_state_closed_form(state: Dict[str, Tensor]) -> None:
    """Fill every tensor of a (reference-shaped) state dict in place: weights by hash_fill_ with a
    fan-in scale, BN affine near (1, 0), running stats (0, 1).  Key order = sorted(), so the fill is
    independent of module construction order."""
    for n, key in enumerate(sorted(state.keys())):
        t = state[key]
        if key.endswith("num_batches_tracked"):
            t.zero_()
        elif key.endswith("running_mean"):
            t.zero_()
        elif key.endswith("running_var"):
            t.fill_(1.0)
        elif ".bn" in key or _is_seq_bn(key, state):
            if key.endswith("weight"):
                hash_fill_(t, n, 0.25); t.add_(1.0)
            else:
                hash_fill_(t, n, 0.1)
        else:
            fan = t.shape[1] if t.dim() >= 2 else t.shape[0]
            if key.endswith("directions"):
                hash_fill_(t, n, 1.0)
            elif key.endswith(".weights"):
                hash_fill_(t, n, 1.0 / math.sqrt(t.shape[0]))
            else:
                hash_fill_(t, n, 1.0 / math.sqrt(max(fan, 1)))


def _is_seq_bn(key: str, state: Dict[str, Tensor]) -> bool:
    """BatchNorm layers inside nn.Sequential blocks have numeric names; spot them by their sibling
    running_mean entry."""
    stem = key.rsplit(".", 1)[0]
    return (stem + ".running_mean") in state


def frontend_inputs(B, H, W, seed, radii):
    """closed-form depth crops: a disc-shaped object mask of the given radius per image, depth in mm with
    ~8 % invalid (zero) pixels, a crop-window pixel grid and per-image intrinsics."""
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    depth = torch.empty(B, 1, H, W)
    mask = torch.empty(B, 1, H, W)
    coor = torch.empty(B, 2, H, W)
    camK = torch.zeros(B, 3, 3)
    for b in range(B):
        u = hash_tensor((H, W), seed + 10 * b, 0.5, 0.5)            # [0,1)
        d = 600.0 + 400.0 * hash_tensor((H, W), seed + 10 * b + 1, 0.5, 0.5)
        depth[b, 0] = torch.where(u < 0.08, torch.zeros_like(d), d)
        r2 = (yy - H / 2.0 + b) ** 2 + (xx - W / 2.0 - b) ** 2
        mask[b, 0] = (r2 < radii[b] ** 2).float()
        coor[b, 0] = xx * 1.75 + 100.0 + 3 * b                          # crop-resized pixel grid
        coor[b, 1] = yy * 1.75 + 60.0 + 2 * b
        camK[b] = torch.tensor([[577.5 + b, 0.0, 319.5], [0.0, 577.5 - b, 239.5], [0.0, 0.0, 1.0]])
    return mask, depth, camK, coor


def generate_rt_inputs():
    """closed-form (p_green, p_red, f_green, f_red, T, sym) for the generate_RT fixture: unit axes incl. a nearly
    parallel and an exactly perpendicular pair, confidences in (0,1), every third object symmetric."""
    Bn = 16
    pg = F.normalize(hash_tensor((Bn, 3), 1000, 1.0), dim=1)
    pr = F.normalize(hash_tensor((Bn, 3), 1001, 1.0), dim=1)
    pr[3] = F.normalize(pg[3] + 0.05 * pr[3], dim=0)      # nearly parallel axes
    pr[4] = F.normalize(torch.cross(pg[4], pr[4], dim=0), dim=0)   # exactly perpendicular
    fg = torch.sigmoid(hash_tensor((Bn,), 1002, 3.0))
    fr = torch.sigmoid(hash_tensor((Bn,), 1003, 3.0))
    T = hash_tensor((Bn, 3), 1004, 0.3) + torch.tensor([0.0, 0.0, 0.8])
    sym = torch.zeros(Bn, 4)
    sym[::3, 0] = 1.0                                                         # bottle / bowl / can style symmetry
    return pg, pr, fg, fr, T, sym


# ---------------------------------------------------------------------------------------------
# closed-form inputs of the loss / augmentation fixtures (oracle/gen_golden_losses.py, tests)
# ---------------------------------------------------------------------------------------------

LOSS_SYM = [[1, 1, 0, 1], [1, 1, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 0], [0, 1, 0, 0], [1, 0, 0, 0]]
LOSS_OBJ = [0, 1, 2, 3, 4, 5, 5]          # bottle, bowl, camera, can, laptop, mug with handle, mug without


def loss_case(n_points: int = 96, seed: int = 4000):
    """a batch of 7 objects (one per symmetry class of the dataset) with a plausible ground truth and network outputs
    that are the ground truth plus noise.  Returns (gt, pred): dicts of tensors; pred tensors are fresh leaves."""
    B, N = len(LOSS_OBJ), n_points
    sym = torch.tensor(LOSS_SYM, dtype=torch.float32)
    obj = torch.tensor(LOSS_OBJ, dtype=torch.float32)
    q, r = torch.linalg.qr(hash_tensor((B, 3, 3), seed, 1.0))
    q = q * torch.sign(torch.diagonal(r, dim1=-2, dim2=-1)).unsqueeze(-2)
    gt_R = q * torch.sign(torch.linalg.det(q)).view(B, 1, 1)                      # proper rotations
    gt_t = hash_tensor((B, 3), seed + 1, 0.1) + torch.tensor([0.0, 0.0, 0.8])
    mean_shape = 0.15 + hash_tensor((B, 3), seed + 2, 0.05)
    gt_s = hash_tensor((B, 3), seed + 3, 0.02)
    size = gt_s + mean_shape
    canon = hash_tensor((B, N, 3), seed + 4, 0.5) * size.unsqueeze(1)             # inside the box
    PC = torch.matmul(canon, gt_R.transpose(1, 2)) + gt_t.unsqueeze(1)
    gt = dict(PC=PC, gt_R=gt_R, gt_t=gt_t, gt_s=gt_s, mean_shape=mean_shape, sym=sym, obj_id=obj)

    def leaf(t):
        return t.clone().requires_grad_(True)
    p_green = F.normalize(gt_R[:, :, 1] + hash_tensor((B, 3), seed + 5, 0.08), dim=1)
    p_red = F.normalize(gt_R[:, :, 0] + hash_tensor((B, 3), seed + 6, 0.08), dim=1)
    axes = gt_R.transpose(1, 2)                                                    # row a = axis a
    order = [(1, 1.0), (0, 1.0), (2, 1.0), (0, -1.0), (2, -1.0), (1, -1.0)]       # network order y+ x+ z+ x- z- y-
    fn = torch.stack([sg * axes[:, a] for a, sg in order], dim=1).unsqueeze(1).expand(B, N, 6, 3)
    fn = F.normalize(fn + hash_tensor((B, N, 6, 3), seed + 7, 0.1), dim=-1)
    fd = torch.stack([size[:, a].unsqueeze(1) / 2 - sg * canon[:, :, a] for a, sg in order], dim=2)
    fd = fd + hash_tensor((B, N, 6), seed + 8, 0.01)
    pred = dict(p_green_R=leaf(p_green), p_red_R=leaf(p_red),
                f_green_R=leaf(torch.sigmoid(hash_tensor((B,), seed + 9, 2.0))),
                f_red_R=leaf(torch.sigmoid(hash_tensor((B,), seed + 10, 2.0))),
                Pred_T=leaf(gt_t + hash_tensor((B, 3), seed + 11, 0.01)),
                Pred_s=leaf(gt_s + hash_tensor((B, 3), seed + 12, 0.01)),
                recon=leaf(PC + hash_tensor((B, N, 3), seed + 13, 0.01)),
                face_normal=leaf(fn), face_dis=leaf(fd),
                face_f=leaf(torch.sigmoid(hash_tensor((B, N, 6), seed + 14, 2.0))))
    return gt, pred


def augment_case(n_points: int = 64, seed: int = 4100):
    """inputs of the augmentation fixture: the loss batch plus per-sample augmentation parameters and model points."""
    gt, _ = loss_case(n_points, seed)
    B = gt["PC"].shape[0]
    ang = hash_tensor((B, 3), seed + 20, 0.2)
    cx, sx, cy, sy, cz, sz = torch.cos(ang[:, 0]), torch.sin(ang[:, 0]), torch.cos(ang[:, 1]), torch.sin(ang[:, 1]), \
        torch.cos(ang[:, 2]), torch.sin(ang[:, 2])
    zero, one = torch.zeros(B), torch.ones(B)
    Rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).view(B, 3, 3)
    Ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(B, 3, 3)
    Rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).view(B, 3, 3)
    gt.update(aug_bb=1.0 + hash_tensor((B, 3), seed + 21, 0.2), aug_rt_t=hash_tensor((B, 3), seed + 22, 0.02),
              aug_rt_r=Rz @ Ry @ Rx, model_point=hash_tensor((B, 48, 3), seed + 23, 0.5),
              nocs_scale=0.3 + hash_tensor((B,), seed + 24, 0.05))
    return gt


def hspose_train_case(B: int, N: int, seed: int):
    """closed-form ground truth for the full-step fixture: the cloud and category ids of the stack fixtures
    (hash cloud of 5 cm spread at 0.8 m) with a pose / size ground truth placed on it."""
    pts = hash_tensor((B, N, 3), seed, 0.05)
    pts[:, :, 2] += 0.8
    import numpy as np
    obj = torch.from_numpy((hash_unit(B, seed + 1) * 6).astype(np.int64)).float()
    q, r = torch.linalg.qr(hash_tensor((B, 3, 3), seed + 30, 1.0))
    q = q * torch.sign(torch.diagonal(r, dim1=-2, dim2=-1)).unsqueeze(-2)
    gt_R = q * torch.sign(torch.linalg.det(q)).view(B, 1, 1)
    table = torch.tensor(LOSS_SYM[:6], dtype=torch.float32)
    return dict(PC=pts, obj_id=obj, gt_R=gt_R, gt_t=pts.mean(dim=1) + hash_tensor((B, 3), seed + 31, 0.01),
                gt_s=hash_tensor((B, 3), seed + 32, 0.02), mean_shape=0.12 + hash_tensor((B, 3), seed + 33, 0.03),
                sym=table[obj.long()], aug_bb=torch.ones(B, 3), aug_rt_t=torch.zeros(B, 3),
                aug_rt_r=torch.eye(3).repeat(B, 1, 1), model_point=hash_tensor((B, 32, 3), seed + 34, 0.5),
                nocs_scale=torch.full((B,), 0.3))


# ------------------------------------------------------------------------------------------------
# closed-form inputs of the Chamfer / FPS fixtures (oracle/gen_golden_chamfer_fps.py writes the reference's
# outputs for exactly these; the tests re-create them)
# ------------------------------------------------------------------------------------------------
CHAMFER_CASES = ["chamfer_100_50", "chamfer_257_1028", "chamfer_ties", "chamfer_1_7"]
FPS_CASES = ["fps_512_64", "fps_1028_256", "fps_lattice_512_128", "fps_dups_300_40"]


def chamfer_case(name: str):
    """(xyz1 (B,n,3), xyz2 (B,m,3), grad_dist1 (B,n), grad_dist2 (B,m)) fp32"""
    if name == "chamfer_100_50":                                   # SURVEY 8c's case
        x1, x2 = hash_tensor((2, 100, 3), 91, 0.5), hash_tensor((2, 50, 3), 92, 0.5)
    elif name == "chamfer_257_1028":                               # the hot path's level sizes, object scale
        x1, x2 = hash_tensor((2, 257, 3), 95, 0.05), hash_tensor((2, 1028, 3), 96, 0.05)
    elif name == "chamfer_ties":                                   # exact ties: duplicated points on a coarse lattice
        x1 = torch.round(hash_tensor((1, 96, 3), 97, 2.0) * 2) / 2
        x2 = torch.round(hash_tensor((1, 160, 3), 98, 2.0) * 2) / 2
        x2[:, 100:140] = x2[:, 20:60]
    elif name == "chamfer_1_7":                                    # n == 1, ragged tiny
        x1, x2 = hash_tensor((3, 1, 3), 99, 1.0), hash_tensor((3, 7, 3), 100, 1.0)
    else:
        raise KeyError(name)
    B, n, m = x1.shape[0], x1.shape[1], x2.shape[1]
    return x1.contiguous(), x2.contiguous(), hash_tensor((B, n), 93, 1.0), hash_tensor((B, m), 94, 1.0)


def fps_case(name: str):
    """(points (N,3) float64 numpy -- cast to float32 for the fp32 rule --, n_samples)"""
    if name == "fps_512_64":
        return hash_tensor((512, 3), 81, 1.0).double().numpy(), 64
    if name == "fps_1028_256":                                     # object-scale cloud at the hot path's N
        return (hash_tensor((1028, 3), 82, 0.05) + torch.tensor([0.0, 0.0, 0.8])).double().numpy(), 256
    if name == "fps_lattice_512_128":                              # 8x8x8 lattice, every point moved by a few ulps
        g = torch.stack(torch.meshgrid(torch.arange(8.), torch.arange(8.), torch.arange(8.), indexing="ij"), -1).reshape(-1, 3)
        return (g + 3.0 + hash_tensor((512, 3), 83, 2.0 ** -20)).double().numpy(), 128
    if name == "fps_dups_300_40":                                  # duplicated points: zero distances, index ties
        p = hash_tensor((300, 3), 84, 1.0)
        p[150:300] = p[0:150]
        return p.double().numpy(), 40
    raise KeyError(name)


# ------------------------------------------------------------------------------------------------------------------------------
# closed-form inputs shared by the fixture generators (oracle/gen_golden_tiled.py, gen_golden_eval_loop.py) and the tests
# ------------------------------------------------------------------------------------------------------------------------------
def tiled_batch(bases, seed, n_pts=1028):
    """(B, n_pts, 3) float32: cloud b = bases[b] closed-form points at ~0.8 m, brought to n_pts points the way the reference's
    loader pads a short crop (datasets/load_data.py:314-316: whole repetitions, then the leading remainder)"""
    clouds = []
    for b, L in enumerate(bases):
        pcl = hash_tensor((L, 3), seed + 17 * b, 0.05).numpy()
        pcl[:, 2] += np.float32(0.8)
        if L < n_pts:
            pcl = np.concatenate([np.tile(pcl, (n_pts // L, 1)), pcl[:n_pts % L]], axis=0)
        clouds.append(pcl)
    return torch.from_numpy(np.stack(clouds, 0).astype(np.float32))


EVAL_LOOP_IMAGES = {1: (1028,), 4: (1028, 350, 1028, 1028), 6: (1028, 1028, 900, 1028, 1028, 514)}   # base points per instance
EVAL_LOOP_SEED = 300


def eval_loop_move_bn_stats(state):
    """running statistics off their defaults, closed form (a trained checkpoint's are not 0 / 1)"""
    for i, (k, v) in enumerate(sorted(state.items())):
        if k.endswith("running_mean"):
            hash_fill_(v, 9000 + i, 0.05)
        elif k.endswith("running_var"):
            hash_fill_(v, 9000 + i, 0.4)
            v.abs_().add_(0.6)


def eval_loop_inputs(n_inst):
    """(PC (n,1028,3), obj_id (n,1) int64, mean_shape (n,3), sym (n,4)) of one surrogate image (evaluate.py:91-96)"""
    pts = tiled_batch(EVAL_LOOP_IMAGES[n_inst], EVAL_LOOP_SEED + 10 * n_inst)
    obj = torch.from_numpy((hash_unit(n_inst, EVAL_LOOP_SEED + n_inst) * 6).astype(np.int64)).view(n_inst, 1)
    mean_shape = hash_tensor((n_inst, 3), EVAL_LOOP_SEED + 100 + n_inst, 0.05) + 0.15
    sym = torch.zeros(n_inst, 4)
    for i in range(n_inst):
        if i % 3 == 0:
            sym[i] = torch.tensor([1.0, 1.0, 0.0, 1.0])           # bottle-like: the red axis is dropped (geom_utils.py:238)
    return pts, obj, mean_shape, sym
