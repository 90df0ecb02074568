"""Drop-in mirror of the reference's ``network/fs_net_repo/gcn3d.py`` operator surface, running on
libhsp.so (gfx950 HIP kernels).

Same public names, constructor / forward signatures, parameter names, shapes and init distributions
as the reference (cited per symbol), so ``FaceRecon``-shaped callers and published checkpoints work
unchanged.  What differs is the execution plan:

  * neighbour indices come from the LDS / f32-MFMA KNN kernels -- no (B,N,N) matrix;
  * graph_conv is one fused kernel per layer -- no (B,N,k,S*C) tensors;
  * the xyz-space KNN of one resolution is computed ONCE per forward and shared by the RF-P branch,
    every ORL branch and the pool of that resolution (the reference recomputes it 4x at N0, 3x at
    N1).  One search yields BOTH lists a resolution needs -- the layers' k = 20 and Pool_layer's k = 4
    (reference :236) -- each in torch.topk's own order among equal distances: on a tiled cloud
    (datasets/load_data.py:314-316) the short list is not the prefix of the long one (ops.knn_xyz);
  * conv2(cat[feature, f_global]) is evaluated as feature @ Wa^T + (f_global @ Wb^T) broadcast --
    f_global is constant over the points of a cloud, so half of that GEMM is a per-cloud bias.
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, ops_bf16

# ------------------------------------------------------------------------------------------------
# per-forward xyz-KNN memo
# ------------------------------------------------------------------------------------------------
_knn_memo = None


POOL_K = 4                                 # Pool_layer's list length in FaceRecon (reference FaceRecon.py:21,24)


@contextlib.contextmanager
def knn_scope():
    """Within the scope, xyz-space KNN results are memoised per (vertices tensor -- by identity --, list length); the first
    search of a resolution also produces the POOL_K-list of the Pool_layer that follows (ops.knn_xyz: one search, two lists)."""
    global _knn_memo, _levels
    prev, _knn_memo = _knn_memo, {}
    prev_levels, _levels = _levels, None
    try:
        yield
    finally:
        _knn_memo, _levels = prev, prev_levels


# ------------------------------------------------------------------------------------------------
# the coarse levels' geometry, prefetched (one launch for what used to be four spread over the forward)
# ------------------------------------------------------------------------------------------------
_levels = None          # id(level-0 vertices) / id(v1) -> dict(vertices, sel, v_pool); "up": [up1, up2]


def prefetch_levels(vertices, k, pool_k=None, rates=(4, 4)):
    """FaceRecon's two Pool_layers keep rows that are drawn on the HOST (torch.randperm, gcn3d.py:243) and depend on nothing the
    network computes, so everything the forward will ask of the two coarse clouds -- their vertices, their neighbour lists, the
    nearest-point maps of the up-sampling -- can be computed from the input cloud at once (ops.geometry_levels).  Draws (or takes
    from the pool_index_feed) both index sets in the reference's order, fills the knn_scope memo for the two levels and remembers
    the kept rows for the Pool_layers.  Returns (up1, up2) or None when the shapes are not the fused kernel's (nothing is consumed
    then, and the layers do their own searches as before).  Call inside knn_scope()."""
    global _levels
    _levels = None
    if _knn_memo is None or vertices.dtype != torch.float32 or vertices.requires_grad or not vertices.is_cuda:
        return None
    pool_k = POOL_K if pool_k is None else pool_k
    n0 = vertices.shape[1]
    n1 = int(n0 / 4)
    n2 = int(n1 / 4)
    k1, k2 = min(k, n1 // 8), min(k, n2 // 8)
    # every condition of ops.geometry_levels / geometry_all is checked HERE, before an index draw is consumed (a refusal after the
    # draws could not fall back without drawing again, out of the reference's order); ``rates``: the two Pool_layers' pooling rates
    # (another rate than 4 and the layers would ignore the prefetch and draw a second time)
    if (tuple(rates) != (4, 4) or vertices.shape[2] != 3 or not (64 <= n2 <= n1 <= 576) or k1 <= pool_k or k1 < 1 or k2 < 1
            or k1 + 2 > 33 or k2 + 2 > 33 or k1 + 1 > n1 or k2 + 1 > n2):
        return None
    if _pool_feed is not None:
        sel1, sel2 = next(_pool_feed), next(_pool_feed)
        assert sel1.numel() == n1 and sel2.numel() == n2 and sel1.dtype == torch.int32
    else:                                                    # the reference's own draws, in its order (pool_1, then pool_2)
        sel1 = torch.randperm(n0)[:n1].to(device=vertices.device, dtype=torch.int32)
        sel2 = torch.randperm(n1)[:n2].to(device=vertices.device, dtype=torch.int32)
    # with the input cloud's own search in the same pair of launches where the shapes allow (its tie pass rides in the levels' launch)
    geo = ops.geometry_all(vertices, k, pool_k, sel1, sel2, k1, pool_k, k2) if k > pool_k else None
    if geo is not None:
        _knn_memo[id(vertices)] = (vertices, {k: geo["idx0"], pool_k: geo["idx0_pool"]})
    else:
        geo = ops.geometry_levels(vertices, sel1, sel2, k1, pool_k, k2)
    if geo is None:
        raise RuntimeError("prefetch_levels: shape check and kernel disagree")      # (indices already consumed: cannot fall back)
    v1, v2 = geo["v1"], geo["v2"]
    _knn_memo[id(v1)] = (v1, {k1: geo["idx1"], pool_k: geo["idx1_pool"]})
    _knn_memo[id(v2)] = (v2, {k2: geo["idx2"]})
    _levels = {id(vertices): (vertices, sel1, v1), id(v1): (v1, sel2, v2), "up": (geo["up1"], geo["up2"])}
    return _levels["up"]


def _xyz_knn(vertices, k):
    """int32 (B,N,k): get_neighbor_index(vertices, k) of the reference, ties as torch.topk leaves them."""
    if _knn_memo is None or vertices.dtype != torch.float32:
        return ops.knn(vertices, k)
    key = id(vertices)
    hit = _knn_memo.get(key)
    if hit is None or hit[0] is not vertices:
        hit = (vertices, {})               # holding the tensor keeps id() unique for the scope
        _knn_memo[key] = hit
    lists = hit[1]
    if k not in lists:
        if k > POOL_K and POOL_K not in lists:
            lists[k], lists[POOL_K] = ops.knn_xyz(vertices, k, POOL_K)
        else:
            lists[k] = ops.knn(vertices, k)
    return lists[k]


# ------------------------------------------------------------------------------------------------
# Pool_layer sample indices supplied from outside (hipGraph replay: no host work inside the graph)
# ------------------------------------------------------------------------------------------------
_pool_feed = None


@contextlib.contextmanager
def pool_index_feed(device_indices):
    """Within the scope, successive Pool_layer.forward calls take their kept-row indices (int32 device
    tensors) from ``device_indices`` in call order instead of drawing torch.randperm themselves.  The
    caller draws them exactly as the reference would (hs_pose_amd.graph.draw_pool_indices)."""
    global _pool_feed
    prev, _pool_feed = _pool_feed, iter(device_indices)
    try:
        yield
    finally:
        _pool_feed = prev


# ------------------------------------------------------------------------------------------------
# functional API (reference gcn3d.py:15-59, :189-218)
# ------------------------------------------------------------------------------------------------

def get_neighbor_index(vertices: "(bs, vertice_num, C)", neighbor_num: int):
    """(bs, vertice_num, neighbor_num) int64 -- reference gcn3d.py:15-24 (works for xyz and features)."""
    return ops.knn(vertices, neighbor_num).long()


def get_nearest_index(target: "(bs, v1, 3)", source: "(bs, v2, 3)"):
    """(bs, v1, 1) int64 -- reference gcn3d.py:27-36."""
    return ops.nn1(target, source).long().unsqueeze(-1)


def indexing_neighbor_new(tensor: "(bs, vertice_num, dim)", index: "(bs, out_num, neighbor_num)"):
    """(bs, out_num, neighbor_num, dim) batched row gather -- reference gcn3d.py:39-47."""
    bs, out_num, n = index.shape
    flat = index.reshape(bs, out_num * n).to(torch.int32)
    return ops.gather_rows(tensor, flat).view(bs, out_num, n, tensor.shape[-1])


def get_neighbor_direction_norm(vertices, neighbor_index, return_unnormed=False):
    """(bs, vertice_num, neighbor_num, 3) unit directions -- reference gcn3d.py:49-59.  (The fused
    layers recompute these in-kernel; this materialising form exists for API parity.)"""
    neighbors = indexing_neighbor_new(vertices, neighbor_index)
    direction = neighbors - vertices.unsqueeze(2)
    normed = F.normalize(direction, dim=-1).float()
    return (normed, direction) if return_unnormed else normed


def get_receptive_fields(neighbor_num, vertices, feature_map=None, mode='RF-F'):
    """reference gcn3d.py:189-209: KNN in feature space ('RF-F') or xyz space ('RF-P'); directions are
    always measured in xyz space."""
    assert mode in ['RF-F', 'RF-P']
    if mode == 'RF-F':
        assert feature_map is not None, "The feature_map should be provided if 'RF-F' is used"
        feat = feature_map
    else:
        feat = vertices
    neighbor_index = get_neighbor_index(feat, neighbor_num)
    return get_neighbor_direction_norm(vertices, neighbor_index), neighbor_index


def get_ORL_global(feature, vertices, neighbor_num):
    """(bs, vertice_num, C), constant along points -- reference gcn3d.py:211-218."""
    fg = ops.orl_global(feature, _xyz_knn(vertices, neighbor_num), neighbor_num)
    return fg.unsqueeze(1).repeat(1, feature.size(1), 1)


def _orl_fused(feature, vertices, neighbor_num, conv2_weight):
    """conv2(cat[feature, f_global]) + feature (reference gcn3d.py:109-113 / :183-187) without the
    concat: W = [Wa | Wb];  feature @ Wa^T + (fg @ Wb^T)[:, None, :] + feature."""
    C = feature.shape[-1]
    w = conv2_weight.squeeze(-1)                       # (C, 2C)
    fg = ops.orl_global(feature, _xyz_knn(vertices, neighbor_num), neighbor_num)   # (B,C)
    b, n, _ = feature.shape
    lin = ops.linear_rows(feature.reshape(b * n, C), w[:, :C]).view(b, n, C)
    return feature + lin + ops.linear_rows(fg, w[:, C:]).unsqueeze(1)


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------

class HSlayer_surface(nn.Module):
    """reference gcn3d.py:61-113.  Parameters: directions (3, S*K), STE_layer.weight (K,3,1),
    conv2.weight (K,2K,1).  ``out_dtype`` (fp32 default): torch.bfloat16 makes the layer emit bf16 feature rows
    (FaceRecon.set_feature_dtype; the coordinates and directions it consumes stay fp32)."""
    out_dtype = torch.float32

    def __init__(self, kernel_num, support_num):
        super().__init__()
        self.feat_k = 8
        self.kernel_num = kernel_num
        self.support_num = support_num
        self.relu = nn.ReLU(inplace=True)
        self.directions = nn.Parameter(torch.empty(3, support_num * kernel_num))
        self.STE_layer = nn.Conv1d(3, kernel_num, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(2 * kernel_num, kernel_num, kernel_size=1, bias=False)
        self.initialize()

    def initialize(self):
        stdv = 1. / math.sqrt(self.support_num * self.kernel_num)
        self.directions.data.uniform_(-stdv, stdv)

    def forward(self, vertices: "(bs, vertice_num, 3)", neighbor_num: int, relu_fork: bool = False):
        """(bs, vertice_num, kernel_num) -- STE + RF-P graph conv + ORL as one fused autograd node.  ``relu_fork`` (fp32 rows):
        relu inside the node, the result returned twice (ops.surface_layer)."""
        idx = _xyz_knn(vertices, neighbor_num)                       # RF-P
        if idx.shape[2] != neighbor_num:
            idx = idx[:, :, :neighbor_num].contiguous()
        if relu_fork and self.out_dtype != torch.bfloat16:
            return ops.surface_layer(vertices, idx, neighbor_num, self.support_num, self.directions, self.STE_layer.weight,
                                     self.conv2.weight, relu=True)
        layer = ops_bf16.surface_layer if self.out_dtype == torch.bfloat16 else ops.surface_layer
        return layer(vertices, idx, neighbor_num, self.support_num, self.directions, self.STE_layer.weight, self.conv2.weight)

    def graph_conv(self, neighbor_index, vertices, neighbor_num):
        """fused relu(R @ D^) -> max over neighbours -> mean over supports (reference :92-107).  Takes the
        int32 neighbour index (directions are recomputed in-kernel) instead of the reference's
        materialised (bs,N,k,3) receptive field."""
        idx = neighbor_index[:, :, :neighbor_num] if neighbor_index.shape[2] != neighbor_num else neighbor_index
        # F.normalize(self.directions, dim=0) and its Jacobian are applied inside the kernels
        return ops.rf_surface(vertices, idx.to(torch.int32), self.directions, self.support_num)

    def ORL_forward(self, feature, vertices, neighbor_num):
        return _orl_fused(feature, vertices, neighbor_num, self.conv2.weight)


class HS_layer(nn.Module):
    """reference gcn3d.py:116-187.  Parameters: weights (Cin,(S+1)*Cout), bias ((S+1)*Cout),
    directions (3,S*Cout), STE_layer.weight (Cout,Cin,1), conv2.weight (Cout,2*Cout,1).  With bf16 feature rows
    (FaceRecon.set_feature_dtype) ``out_fp32`` makes the layer write its output in fp32 -- set for the layers a BatchNorm
    follows."""
    out_fp32 = False

    def __init__(self, in_channel, out_channel, support_num):
        super().__init__()
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.support_num = support_num
        self.relu = nn.ReLU(inplace=True)
        self.weights = nn.Parameter(torch.empty(in_channel, (support_num + 1) * out_channel))
        self.bias = nn.Parameter(torch.empty((support_num + 1) * out_channel))
        self.directions = nn.Parameter(torch.empty(3, support_num * out_channel))
        self.feat_k = 8
        self.STE_layer = nn.Conv1d(self.in_channel, self.out_channel, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(2 * out_channel, out_channel, kernel_size=1, bias=False)
        self.initialize()

    def initialize(self):
        stdv = 1. / math.sqrt(self.out_channel * (self.support_num + 1))
        self.weights.data.uniform_(-stdv, stdv)
        self.bias.data.uniform_(-stdv, stdv)
        self.directions.data.uniform_(-stdv, stdv)

    def forward(self, vertices: "(bs, vertice_num, 3)", feature_map: "(bs, vertice_num, in_channel)",
                neighbor_num: int, bn_shift=None, transposed_view=False):
        """(bs, vertice_num, out_channel) -- STE + fm GEMM + RF-F graph conv + ORL as one fused autograd node.  ``bn_shift``
        (fp32 rows): see ops.hs_layer -- returns (out, BatchNorm partial sums)."""
        # RF-F: neighbours in feature space (``transposed_view``: see ops.knn -- FaceRecon sets it for conv_3)
        neighbor_index = ops.knn(feature_map, neighbor_num, **({"transposed_view": True} if transposed_view else {}))
        if feature_map.dtype == torch.bfloat16:                      # bf16 feature rows in -> out (fp32 out ahead of a BatchNorm)
            return ops_bf16.hs_layer(vertices, feature_map, neighbor_index, _xyz_knn(vertices, neighbor_num), neighbor_num,
                                     self.support_num, self.weights, self.bias, self.directions, self.STE_layer.weight,
                                     self.conv2.weight, out_f32=self.out_fp32)
        return ops.hs_layer(vertices, feature_map, neighbor_index, _xyz_knn(vertices, neighbor_num), neighbor_num,
                            self.support_num, self.weights, self.bias, self.directions, self.STE_layer.weight, self.conv2.weight,
                            bn_shift=bn_shift)

    def graph_conv(self, neighbor_index, feature_map, vertices, neighbor_num):
        """reference :158-181 with the gather, theta product, max and mean fused into one kernel."""
        bs, n, cin = feature_map.shape
        fm = ops.linear_rows(feature_map.reshape(bs * n, cin), self.weights.t().contiguous(), self.bias).view(bs, n, -1)
        return ops.rf_conv(vertices, neighbor_index.to(torch.int32), self.directions, fm, self.support_num)

    def ORL_forward(self, feature_fuse, vertices, neighbor_num):
        return _orl_fused(feature_fuse, vertices, neighbor_num, self.conv2.weight)


class Pool_layer(nn.Module):
    """reference gcn3d.py:220-246: max over the 4 nearest (rank 0 dropped), then ONE torch.randperm
    draw on the CPU default generator shared by the whole batch (same RNG consumption as the
    reference, so fixed-seed runs pick the same points)."""

    def __init__(self, pooling_rate: int = 4, neighbor_num: int = 4):
        super().__init__()
        self.pooling_rate = pooling_rate
        self.neighbor_num = neighbor_num

    def forward(self, vertices: "(bs, vertice_num, 3)", feature_map: "(bs, vertice_num, channel_num)"):
        """-> vertices_pool (bs, pool_num, 3), feature_map_pool (bs, pool_num, channel_num)"""
        bs, vertice_num, _ = vertices.size()
        neighbor_index = _xyz_knn(vertices, self.neighbor_num)
        pool_num = int(vertice_num / self.pooling_rate)
        pre = _levels.get(id(vertices)) if _levels is not None else None
        if pre is not None and pre[0] is vertices and pre[1].numel() == pool_num and self.pooling_rate == 4:
            # the kept rows were drawn, and the pooled vertices gathered, by prefetch_levels: only the features are pooled here
            # (the SAME vertices tensor is handed on, so the next level finds its prefetched neighbour lists)
            _, sel, v_pool = pre
            if feature_map.dtype == torch.float32 and feature_map.shape[2] >= 12:
                feature_map_pool, _ = ops.pool_layer(feature_map, vertices, neighbor_index, sel, self.neighbor_num)
            else:
                feature_map_pool = ops.gather_max(feature_map, neighbor_index, self.neighbor_num, qsel=sel)
            return v_pool, feature_map_pool
        if _pool_feed is not None:
            sel = next(_pool_feed)
            assert sel.numel() == pool_num and sel.dtype == torch.int32
        else:
            sample_idx = torch.randperm(vertice_num)[:pool_num]
            sel = sample_idx.to(device=vertices.device, dtype=torch.int32)
        # only the kept rows are pooled (the reference pools all N rows, then selects)
        if feature_map.dtype == torch.float32 and feature_map.shape[2] >= 12 and not vertices.requires_grad:
            feature_map_pool, vertices_pool = ops.pool_layer(feature_map, vertices, neighbor_index, sel, self.neighbor_num)
            return vertices_pool, feature_map_pool
        feature_map_pool = ops.gather_max(feature_map, neighbor_index, self.neighbor_num, qsel=sel)
        vertices_pool = ops.gather_rows(vertices, sel)
        return vertices_pool, feature_map_pool
