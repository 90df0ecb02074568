"""Drop-in mirror of the reference's ``network/HSPose.py`` operator surface (HSPose.py:23-276).

Same constructor, ``forward`` keyword set, ``output_dict`` keys and ``build_params`` as the reference.
The inference path (``FLAGS.train = 0``; evaluation/evaluate.py:91-106), the depth -> cloud entry (``PC is None``,
HSPose.py:39-48), the on-device augmentation (``FLAGS.train``, HSPose.py:53-61, hs_pose_amd/augment.py) and the
training losses (``do_loss=True``, HSPose.py:84-181, hs_pose_amd/losses.py): ``forward`` returns ``output_dict``
or ``(output_dict, loss_dict)`` with the four sub-dictionaries ``engine/train.py:84-90`` sums.
"""
import os

import torch
import torch.nn as nn

from .augment import data_augment
from .config import FLAGS
from .losses import control_loss, fs_net_loss, geo_transform_loss, prop_rot_loss, recon_6face_loss
from .pc_sample import PC_sample
from .PoseNet9D import PoseNet9D


_NET_OUTPUTS = ('recon', 'face_normal', 'face_dis', 'face_f', 'p_green_R', 'p_red_R', 'f_green_R', 'f_red_R', 'Pred_T',
                'Pred_s')


def get_gt_v(Rs, axis=2):
    """green (y) and red (x) axes of the ground-truth rotations, R[:, :, 1] and R[:, :, 0] (tools/training_utils.py:59-73
    multiplies R by a 0/1 corner matrix and picks rows of the transposed product: the same numbers, exactly)."""
    assert axis in (2, 3)
    return Rs[:, :, 1].clone(), Rs[:, :, 0].clone()


class HSPose(nn.Module):
    def __init__(self, train_stage):
        super(HSPose, self).__init__()
        self.posenet = PoseNet9D()
        self.train_stage = train_stage
        self.loss_recon = recon_6face_loss()
        self.loss_fs_net = fs_net_loss()
        self.loss_geo = geo_transform_loss()
        self.loss_prop = prop_rot_loss()
        self.name_fs_list, self.name_recon_list, self.name_geo_list, self.name_prop_list = control_loss(self.train_stage)

    def forward(self, PC=None, depth=None, obj_id=None, camK=None,
                gt_R=None, gt_t=None, gt_s=None, mean_shape=None, gt_2D=None, sym=None, aug_bb=None,
                aug_rt_t=None, aug_rt_r=None, def_mask=None, model_point=None, nocs_scale=None, do_loss=False):
        output_dict = {}

        if PC is None:
            if self.train_stage == 'PoseNet_only':             # HSPose.py:40-48
                FLAGS.sample_method = 'basic'
                PC = PC_sample(def_mask, depth, camK, gt_2D)
                if PC is None or isinstance(PC, tuple):        # PC_sample signals "no points" with (None, None)
                    return output_dict, None
            else:
                raise NotImplementedError

        obj_mask = None
        sketch = None
        PC = PC.detach()
        if FLAGS.train:
            with torch.no_grad():
                PC, gt_R, gt_t, gt_s = self.data_augment(PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t,
                                                         aug_rt_r, model_point, nocs_scale, obj_id)

        runner = self.graphed_posenet.get(tuple(PC.shape)) if self.graphed_posenet else None
        if runner is not None and self.training and torch.is_grad_enabled():
            net_out = runner(PC, obj_id)                      # two hipGraph replays behind one autograd node
        else:
            net_out = self.posenet(PC, obj_id)
        out = dict(zip(_NET_OUTPUTS, net_out))
        # the reference's 16 keys in its order (HSPose.py:67-82): mask / sketch are always None in this training stage
        output_dict.update(mask=obj_mask, sketch=sketch, recon=out['recon'], PC=PC)
        output_dict.update({k: out[k] for k in _NET_OUTPUTS[1:]})
        output_dict.update(gt_R=gt_R, gt_t=gt_t, gt_s=gt_s)
        if not do_loss:
            return output_dict

        if self.fused_losses and PC.is_cuda and self.train_stage == 'PoseNet_only' and FLAGS.prop_sym_w > 0:
            # the same 19 terms from libhsp's five loss kernels (fused_losses.py; csrc/losses.hip)
            from .fused_losses import pose_losses
            return output_dict, pose_losses(out, PC, gt_R, gt_t, gt_s, mean_shape, sym, obj_id)

        # the four loss modules read their inputs from dictionaries keyed as in the reference (HSPose.py:84-160); the axis
        # confidences enter every loss except fs_net's own confidence terms as constants (detached)
        axes = {'Rot1': out['p_green_R'], 'Rot2': out['p_red_R']}
        conf = {'Rot1_f': out['f_green_R'], 'Rot2_f': out['f_red_R']}
        conf_const = {k: v.detach() for k, v in conf.items()}
        pose = {'Tran': out['Pred_T'], 'Size': out['Pred_s']}
        gt_pose = {'Points': PC, 'R': gt_R, 'T': gt_t, 'Mean_shape': mean_shape}
        green_gt, red_gt = (None, None) if self.train_stage == 'Backbone_only' else get_gt_v(gt_R)

        fsnet_loss = self.loss_fs_net(self.name_fs_list, {**axes, **conf, **pose, 'Recon': out['recon']},
                                      {'Rot1': green_gt, 'Rot2': red_gt, 'Recon': PC, 'Tran': gt_t, 'Size': gt_s}, sym)
        prop_loss = self.loss_prop(self.name_prop_list,
                                   {**axes, **conf_const, 'Recon': out['recon'], 'Tran': out['Pred_T'], 'Scale': out['Pred_s']},
                                   gt_pose, sym)
        recon_loss = self.loss_recon(self.name_recon_list,
                                     {**axes, **conf_const, **pose, 'F_n': out['face_normal'], 'F_d': out['face_dis'],
                                      'F_c': out['face_f']},
                                     {**gt_pose, 'Size': gt_s}, sym, obj_id)
        geo_loss = self.loss_geo(self.name_geo_list, {**axes, **conf_const, **pose}, gt_pose, sym)

        loss_dict = {'fsnet_loss': fsnet_loss, 'recon_loss': recon_loss, 'geo_loss': geo_loss, 'prop_loss': prop_loss}
        return output_dict, loss_dict

    @staticmethod
    def total_loss(loss_dict):
        """the scalar engine/train.py:84-90 backpropagates: the sum of every term of ``loss_dict`` (from the fused loss kernels'
        own reduction when they produced the dict, hs_pose_amd/fused_losses.py::total_loss)"""
        from .fused_losses import total_loss
        return total_loss(loss_dict)

    graphed_posenet = None
    # device batches take the fused loss kernels; False keeps the torch-op composition of losses.py (the readable statement: tests)
    fused_losses = True

    def enable_graphed_posenet(self, PC, obj_id):
        """capture ``posenet`` forward / backward for training batches of this shape (hs_pose_amd.graph.GraphedNetwork);
        ``forward`` then replays the graphs whenever a training batch has a captured shape and runs eagerly otherwise
        (call it once per shape, e.g. also for the last, smaller batch of an epoch).
        Call it before the first eager backward of the network, with FLAGS.train set as in training."""
        from .graph import GraphedNetwork
        runners = dict(self.graphed_posenet or {})
        runners[tuple(PC.shape)] = GraphedNetwork(self.posenet, PC, obj_id)
        object.__setattr__(self, "graphed_posenet", runners)
        return runners[tuple(PC.shape)]

    def data_augment(self, PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t, aug_rt_r, model_point, nocs_scale,
                     obj_ids, check_points=False):
        """HSPose.py:185-256 (check_points: the reference's interactive visualisation, not reproduced)."""
        return data_augment(PC, gt_R, gt_t, gt_s, mean_shape, sym, aug_bb, aug_rt_t, aug_rt_r, model_point, nocs_scale,
                            obj_ids)

    def build_params(self, training_stage_freeze=None):
        """HSPose.py:258-275: one param group, lr = FLAGS.lr * FLAGS.lr_pose.  (The reference's 'pose'
        freeze loop sets an attribute on a tuple and therefore freezes nothing; kept as a no-op.)"""
        return [{
            "params": filter(lambda p: p.requires_grad, self.posenet.parameters()),
            "lr": float(FLAGS.lr) * FLAGS.lr_pose,
        }]
