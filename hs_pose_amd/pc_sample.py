"""Depth map -> sampled object point cloud; mirror of network/point_sample/pc_sample.py:8-77.

``PC_sample(obj_mask, Depth, camK, coor2d)`` keeps the reference's signature, host RNG consumption
(one ``np.random.choice(l_all, samplenum, replace=l_all < samplenum)`` per image, in image order, on
numpy's global generator) and return convention (``(None, None)`` as soon as an image has <= 1 valid
pixel -- pc_sample.py:59-60).  The device work is two launches for the whole batch
(``hsp_pc_compact`` / ``hsp_pc_gather``, csrc/frontend.hip) with ONE device->host copy (the per-image
valid-pixel counts the host needs before it can draw) instead of the reference's per-image chain of
H x W maps, boolean-index compactions and implicit syncs.
"""
import numpy as np
import torch

from . import ops
from .config import FLAGS


def PC_sample(obj_mask, Depth, camK, coor2d):
    """obj_mask (bs,1,H,W) (or (bs,2,H,W) mask logits), Depth (bs,1,H,W) in mm, camK (bs,3,3),
    coor2d (bs,2,H,W) pixel coordinates -> PC (bs, FLAGS.random_points, 3) in metres."""
    if obj_mask.shape[1] == 2:                                 # predicted mask (pc_sample.py:16-18)
        # argmax(softmax(m)) == argmax(m); first index on ties like torch.max
        obj_mask = (obj_mask[:, 1] > obj_mask[:, 0])
    if getattr(FLAGS, "sample_method", "basic") != "basic":    # pc_sample.py:68-70
        raise NotImplementedError
    samplenum = int(FLAGS.random_points)
    bs, H, W = Depth.shape[0], Depth.shape[2], Depth.shape[3]
    mask = obj_mask.reshape(bs, H * W).float()
    pix, count = ops.pc_compact(mask, Depth.reshape(bs, H * W))
    counts = count.cpu().numpy()                               # the one sync of the front end
    choose = np.empty((bs, samplenum), dtype=np.int32)
    for i in range(bs):
        l_all = int(counts[i])
        if l_all <= 1.0:
            return None, None
        choose[i] = np.random.choice(l_all, samplenum, replace=l_all < samplenum)
    choose_d = torch.from_numpy(choose).to(Depth.device, non_blocking=True)
    return ops.pc_gather(Depth.reshape(bs, H * W), coor2d.reshape(bs, 2, H * W), camK, pix, choose_d)


def sample_point_ids(total_pts_num, n_pts):
    """row ids PoseDataset._sample_points keeps (load_data.py:308-320): tile when short, a
    ``np.random.permutation`` prefix when long (same global-RNG consumption), identity otherwise."""
    if total_pts_num < n_pts:
        base = np.arange(total_pts_num)
        return np.concatenate([np.tile(base, n_pts // total_pts_num), base[:n_pts % total_pts_num]])
    if total_pts_num > n_pts:
        return np.random.permutation(total_pts_num)[:n_pts]
    return np.arange(total_pts_num)


def depth_to_pcl(depth, K, xymap, mask, n_pts=None, min_pts=50):
    """Batched mirror of the loader's cloud extraction: ``_depth_to_pcl(depth, K, xymap, mask) / 1000.0``
    (load_data.py:275, :322-333), the ``len(pcl_in) < 50`` rejection (:276) and ``_sample_points`` (:278).

    depth (B,1,H,W) or (B,H,W) fp32 mm, K (3,3) or (B,3,3) (float64 like the loader's intrinsics),
    xymap (B,2,H,W), mask (B,1,H,W) -> (B,n_pts,3) fp32 metres, or None if any image has < min_pts
    valid pixels (the loader skips such an item).  n_pts defaults to FLAGS.random_points."""
    n_pts = int(FLAGS.random_points if n_pts is None else n_pts)
    B = depth.shape[0]
    HW = depth[0].numel()
    d = depth.reshape(B, HW).float()
    K64 = torch.as_tensor(K, dtype=torch.float64, device=depth.device).reshape(-1, 9)
    if K64.shape[0] == 1 and B > 1:
        K64 = K64.expand(B, 9)
    pix, count = ops.pc_compact(mask.reshape(B, HW).float(), d)
    counts = count.cpu().numpy()
    choose = np.empty((B, n_pts), dtype=np.int32)
    for i in range(B):
        if int(counts[i]) < min_pts:
            return None
        choose[i] = sample_point_ids(int(counts[i]), n_pts)
    choose_d = torch.from_numpy(choose).to(depth.device, non_blocking=True)
    return ops.depth_to_pcl(d, xymap.reshape(B, 2, HW), K64.contiguous(), pix, choose_d)
