"""Drop-in mirror of the reference's ``network/fs_net_repo/FaceRecon.py`` (HS-layer stack wiring).

Module / parameter names match the reference (conv_0..conv_4, pool_1/2, bn1..3, and -- when
FLAGS.train -- conv1d_block / recon_head / face_head as nn.Sequential with the same indices), so its
state_dict loads here with strict=True.  Execution differs: point-major (B,N,C) tensors end to end
(BatchNorm and the 1x1 convolutions run on the flattened (B*N, C) view: same statistics, no
transposes), one shared xyz-KNN per resolution, fused graph-conv kernels.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gcn3d, ops, ops_bf16
from .config import FLAGS


def _conv_bn_relu_rows(seq, x, n_blocks, first=None):
    """Apply n_blocks x (Conv1d k=1, BatchNorm1d, ReLU) of an nn.Sequential to (R, C) rows.  first: the first Conv1d's output when
    the caller already has it, as (rows, BatchNorm first-pass buffer) (``ops.fan_linear_rows``)."""
    for i in range(n_blocks):
        conv, bn = seq[3 * i], seq[3 * i + 1]
        y, part = first if (i == 0 and first is not None) else ops.linear_rows(x, conv.weight.squeeze(-1), conv.bias, bn_partials=True)
        x = ops.bn_relu(y, bn, partial=part)                       # fused BatchNorm + ReLU (norm.hip), first pass from the product
    return x


class FaceRecon(nn.Module):
    """reference FaceRecon.py:12-128."""

    def __init__(self):
        super(FaceRecon, self).__init__()
        self.neighbor_num = FLAGS.gcn_n_num
        self.support_num = FLAGS.gcn_sup_num

        self.conv_0 = gcn3d.HSlayer_surface(kernel_num=128, support_num=self.support_num)
        self.conv_1 = gcn3d.HS_layer(128, 128, support_num=self.support_num)
        self.pool_1 = gcn3d.Pool_layer(pooling_rate=4, neighbor_num=4)
        self.conv_2 = gcn3d.HS_layer(128, 256, support_num=self.support_num)
        self.conv_3 = gcn3d.HS_layer(256, 256, support_num=self.support_num)
        self.pool_2 = gcn3d.Pool_layer(pooling_rate=4, neighbor_num=4)
        self.conv_4 = gcn3d.HS_layer(256, 512, support_num=self.support_num)

        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(256)
        self.bn3 = nn.BatchNorm1d(256)

        self.recon_num = 3
        self.face_recon_num = FLAGS.face_recon_c
        dim_fuse = sum([128, 128, 256, 256, 512, FLAGS.obj_c])

        if FLAGS.train:
            def cbr(i, o):
                return [nn.Conv1d(i, o, 1), nn.BatchNorm1d(o), nn.ReLU(inplace=True)]

            self.conv1d_block = nn.Sequential(*cbr(dim_fuse, 512), *cbr(512, 512), *cbr(512, 256))
            self.recon_head = nn.Sequential(*cbr(256, 128), nn.Conv1d(128, self.recon_num, 1))
            self.face_head = nn.Sequential(*cbr(FLAGS.feat_face + 3, 512), *cbr(512, 256), *cbr(256, 128),
                                           nn.Conv1d(128, self.face_recon_num, 1))

    _x3 = None                       # ops.X3Planes of this network (created on first use; PoseNet9D shares it with the heads)
    feat_consumers = None            # PoseNet9D: callable (feat rows, xyz) -> conv1d_block[0]'s output (ops.fan_linear_rows)
    keep_backward_cut = False
    backward_cut = None
    exact_train = os.environ.get("HSP_EXACT_TRAIN") == "1"
    feature_dtype = torch.float32
    _bf16 = None

    def set_feature_dtype(self, dtype):
        """torch.bfloat16: the HS stack stores its feature rows, ``fm`` and activation gradients in bf16 and runs its dense
        products on the bf16 matrix cores (BASELINE configs[3]; hs_pose_amd/ops_bf16.py says what stays fp32); ``feat``
        comes out bf16.  The parameters stay fp32 masters -- call again after anything that re-seats them (moving the
        module, building the fused optimizer).  torch.float32 restores the default path."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("feature dtype: torch.float32 or torch.bfloat16")
        if dtype == torch.bfloat16 and FLAGS.train:             # (checked before anything is changed: the module stays fp32)
            raise NotImplementedError("bf16 feature rows: the HS stack (feat); the train-only heads take fp32 rows")
        self.feature_dtype = dtype
        self.conv_0.out_dtype = dtype
        for layer in (self.conv_1, self.conv_2, self.conv_3):          # BatchNorm follows: fp32 rows out of the bf16 layer
            layer.out_fp32 = dtype == torch.bfloat16
        self._bf16 = None
        if dtype == torch.bfloat16:
            # DETACHED views: a view with a grad_fn would create (and keep alive) the parameter's gradient accumulator bound
            # to whatever stream is current here, and a later backward inside a hipGraph capture would then hop to that
            # stream (an event on the null stream inside a capture crashes hipStreamEndCapture)
            specs = [(self.conv_0.conv2.weight.detach().squeeze(-1), True, True)]
            for layer in (self.conv_1, self.conv_2, self.conv_3, self.conv_4):
                specs += [(layer.weights.detach(), True, True), (layer.STE_layer.weight.detach().squeeze(-1), True, True),
                          (layer.conv2.weight.detach().squeeze(-1), True, True)]
            self._bf16 = ops_bf16.Bf16Params(specs)
        return self

    def forward(self, vertices: "tensor (bs, vetice_num, 3)", cat_id: "tensor (bs, 1)"):
        """-> (recon (bs,N,3) | None, face (bs,N,face_recon_c) | None, feat (bs,N,1286))"""
        bs, vertice_num, _ = vertices.size()
        k = self.neighbor_num
        if self._x3 is None:
            self._x3 = ops.X3Planes()
        # eval mode: the forward in the reference's own operation order, so that the feature-space neighbour search sees the
        # reference's bits (ops.exact_scope); train mode: the faster products
        # ``exact_train`` (HSP_EXACT_TRAIN=1, or set the attribute): the reference-order arithmetic under train-mode BatchNorm too.
        # Measured in round 6 (tools/exact_train_probe.py, fixture stack_refinit_trainbn_1028, free-running): every ordered neighbour
        # list of every HS layer equals the reference's and the six pose / size outputs agree to 9e-6 (fast products: 0.91 / 0.76 /
        # 0.63 / 0.86 of the rows, 4e-2) -- for 2.02 instead of 1.71 ms per training step of the HS stack (fp32 matrix cores instead
        # of the bf16 x3 products, BatchNorm's first pass not fused into the layer's out product).  Off by default: the step that is
        # timed is the fast one; tests/test_gpu_stack.py holds the switched-on form to 1e-4.
        exact = (not self.training or self.exact_train) and self.feature_dtype == torch.float32
        with ops.x3_scope(self._x3), ops.exact_scope(exact):
            return self._forward(vertices, cat_id)

    def _forward(self, vertices, cat_id):
        bs, vertice_num, _ = vertices.size()
        k = self.neighbor_num
        if self._bf16 is not None:
            self._bf16.refresh()                      # fp32 master weights -> this step's bf16 working copies (one launch)
        else:
            ops.x3_refresh()                          # fp32 weights -> the three bf16 slices of the x3 products (one launch)
        with gcn3d.knn_scope():
            # the two coarse levels' vertices, neighbour lists and up-sampling maps in one launch, up front (they depend on the
            # coordinates and the host-drawn pool rows only)
            up = (gcn3d.prefetch_levels(vertices, k, self.pool_1.neighbor_num, rates=(self.pool_1.pooling_rate, self.pool_2.pooling_rate))
                  if self.pool_1.neighbor_num == self.pool_2.neighbor_num else None)
            od = self.feature_dtype if self.feature_dtype == torch.bfloat16 else None
            fork0 = od is None and not self.keep_backward_cut and torch.is_grad_enabled()
            if fork0:
                fm_0, a_0 = self.conv_0(vertices, k, relu_fork=True)          # relu in the node, one tensor per consumer
            else:
                fm_0 = F.relu(self.conv_0(vertices, k), inplace=True)
            # fm_1 .. fm_3 have two consumers each (the next level and the concat): the BatchNorm node hands out one tensor per
            # consumer (``fork``), so their gradients meet inside its backward kernels instead of in an element-wise add.  The
            # two-graph split (keep_backward_cut) cuts at single aliases and keeps the plain form.
            fork = od is None and not self.keep_backward_cut

            def layer_bn(conv, bn, *a):
                """bn_relu(conv(...)); fp32 rows in train mode: the layer's out product also leaves the first pass of the
                BatchNorm statistics (ops.hs_layer bn_shift), whatever the fork mode -- graph and eager twins stay bit-equal"""
                if od is None and bn.training and bn.track_running_stats and torch.is_grad_enabled() and not ops.exact_forward():
                    out, part = conv(*a, bn_shift=True)
                    return ops.bn_relu(out, bn, fork=fork, partial=part)
                if conv is self.conv_3 and ops.exact_forward():
                    # (the reference hands conv_3 relu(bn2(...)).transpose(1, 2) WITHOUT making it contiguous, FaceRecon.py:92-95:
                    # the |x|^2 of its feature-space neighbour search then rounds as a strided sum)
                    return ops.bn_relu(conv(*a, transposed_view=True), bn, out_dtype=od, fork=fork)
                return ops.bn_relu(conv(*a), bn, out_dtype=od, fork=fork)
            if fork:
                fm_1, a_1 = layer_bn(self.conv_1, self.bn1, vertices, fm_0, k)
                v_pool_1, fm_pool_1 = self.pool_1(vertices, fm_1)
                k1 = min(k, v_pool_1.shape[1] // 8)
                fm_2, a_2 = layer_bn(self.conv_2, self.bn2, v_pool_1, fm_pool_1, k1)
                if not fork0:
                    a_0 = fm_0.view_as(fm_0)
                self.backward_cut = None
                fm_3, a_3 = layer_bn(self.conv_3, self.bn3, v_pool_1, fm_2, k1)
            else:
                fm_1 = layer_bn(self.conv_1, self.bn1, vertices, fm_0, k)
                v_pool_1, fm_pool_1 = self.pool_1(vertices, fm_1)
                k1 = min(k, v_pool_1.shape[1] // 8)
                fm_2 = layer_bn(self.conv_2, self.bn2, v_pool_1, fm_pool_1, k1)
                # The coarse levels and the concat read the fine levels through aliases (no kernels): every path from feat
                # down to an alias stays above the others, so a backward pass can stop at them and be resumed
                # (graph.py::GraphedStep(split=True) reduces the coarse levels' gradients while the fine levels still run).
                a_0, a_1, a_2 = fm_0.view_as(fm_0), fm_1.view_as(fm_1), fm_2.view_as(fm_2)
                self.backward_cut = (a_0, a_1, a_2) if self.keep_backward_cut else None   # holds the autograd graph: opt-in
                fm_3 = a_3 = layer_bn(self.conv_3, self.bn3, v_pool_1, a_2, k1)
            v_pool_2, fm_pool_2 = self.pool_2(v_pool_1, fm_3)
            k2 = min(k, v_pool_2.shape[1] // 8)
            fm_4 = self.conv_4(v_pool_2, fm_pool_2, k2)

        nearest_pool_1, nearest_pool_2 = up if up is not None else (ops.nn1(vertices, v_pool_1), ops.nn1(vertices, v_pool_2))
        # nearest up-sampling of the coarse levels, the one-hot category columns and the concat in one kernel
        # (the reference's one_hot = zeros(bs, obj_c).scatter_(1, cat_id.long(), 1), FaceRecon.py:80-85, is built inside the kernel)
        ops.ONE_HOT_WIDTH = FLAGS.obj_c
        feat = ops.assemble_feat([(a_0, None, 0), (a_1, None, 0), (a_2, nearest_pool_1, 1), (a_3, nearest_pool_1, 1),
                                  (fm_4, nearest_pool_2, 1), (cat_id.detach().reshape(-1).float(), None, 3)])

        if FLAGS.train:
            f_global = ops.points_max(fm_4)          # (FaceRecon.py:98 computes it unconditionally; only this branch reads it)
            rows = feat.reshape(bs * vertice_num, -1)
            first = None
            if self.feat_consumers is not None:
                # the layers that read feat's rows -- this block's first Conv1d and, handed over by PoseNet9D, the first layers of
                # the three pose heads -- as ONE node: their input gradients are summed in the products' epilogues
                first = self.feat_consumers(rows, vertices)
            h = _conv_bn_relu_rows(self.conv1d_block, rows, 3, first=first)          # (B*N, 256)
            r = _conv_bn_relu_rows(self.recon_head, h, 1)
            last = self.recon_head[3]
            recon = ops.linear_rows(r, last.weight.squeeze(-1), last.bias).view(bs, vertice_num, -1)
            w0 = self.face_head[0].weight.squeeze(-1)
            if ops.cloud_cat_linear_ok(f_global, h, vertices, w0):
                # cat[f_global over the cloud, h, xyz] (FaceRecon.py:113-116) is never formed: f_global's columns of the first
                # Conv1d act as a per-cloud bias of a K = 259 product
                y0 = ops.cloud_cat_linear(f_global, h, vertices, w0, self.face_head[0].bias)
                f = _conv_bn_relu_rows(self.face_head, None, 3, first=(y0, None))
            else:
                face_in = ops.cat_rows_pitched([f_global.unsqueeze(1).expand(-1, vertice_num, -1).reshape(bs * vertice_num, -1), h,
                                                vertices.reshape(bs * vertice_num, 3)])
                f = _conv_bn_relu_rows(self.face_head, face_in, 3)
            last = self.face_head[9]
            face = ops.linear_rows(f, last.weight.squeeze(-1), last.bias).view(bs, vertice_num, -1)
            return recon, face, feat
        return None, None, feat
