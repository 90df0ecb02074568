"""Drop-in mirror of the reference's ``network/HSPose.py`` operator surface (HSPose.py:23-276).

Same constructor, ``forward`` keyword set, ``output_dict`` keys and ``build_params`` as the reference.
The inference path (``FLAGS.train = 0``; evaluation/evaluate.py:91-106), the depth -> cloud entry (``PC is None``,
HSPose.py:39-48), the on-device augmentation (``FLAGS.train``, HSPose.py:53-61, hs_pose_amd/augment.py) and the
training losses (``do_loss=True``, HSPose.py:84-181, hs_pose_amd/losses.py): ``forward`` returns ``output_dict``
or ``(output_dict, loss_dict)`` with the four sub-dictionaries ``engine/train.py:84-90`` sums.
"""
import torch
import torch.nn as nn

from .augment import data_augment
from .config import FLAGS
from .losses import control_loss, fs_net_loss, geo_transform_loss, prop_rot_loss, recon_6face_loss
from .pc_sample import PC_sample
from .PoseNet9D import PoseNet9D


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
        recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R, Pred_T, Pred_s = net_out

        output_dict['mask'] = obj_mask
        output_dict['sketch'] = sketch
        output_dict['recon'] = recon
        output_dict['PC'] = PC
        output_dict['face_normal'] = face_normal
        output_dict['face_dis'] = face_dis
        output_dict['face_f'] = face_f
        output_dict['p_green_R'] = p_green_R
        output_dict['p_red_R'] = p_red_R
        output_dict['f_green_R'] = f_green_R
        output_dict['f_red_R'] = f_red_R
        output_dict['Pred_T'] = Pred_T
        output_dict['Pred_s'] = Pred_s
        output_dict['gt_R'] = gt_R
        output_dict['gt_t'] = gt_t
        output_dict['gt_s'] = gt_s

        if not do_loss:
            return output_dict

        pred_fsnet_list = {'Rot1': p_green_R, 'Rot1_f': f_green_R, 'Rot2': p_red_R, 'Rot2_f': f_red_R, 'Recon': recon,
                           'Tran': Pred_T, 'Size': Pred_s}
        gt_green_v, gt_red_v = (None, None) if self.train_stage == 'Backbone_only' else get_gt_v(gt_R)
        gt_fsnet_list = {'Rot1': gt_green_v, 'Rot2': gt_red_v, 'Recon': PC, 'Tran': gt_t, 'Size': gt_s}
        fsnet_loss = self.loss_fs_net(self.name_fs_list, pred_fsnet_list, gt_fsnet_list, sym)

        pred_prop_list = {'Recon': recon, 'Rot1': p_green_R, 'Rot2': p_red_R, 'Tran': Pred_T, 'Scale': Pred_s,
                          'Rot1_f': f_green_R.detach(), 'Rot2_f': f_red_R.detach()}
        gt_prop_list = {'Points': PC, 'R': gt_R, 'T': gt_t, 'Mean_shape': mean_shape}
        prop_loss = self.loss_prop(self.name_prop_list, pred_prop_list, gt_prop_list, sym)

        pred_recon_list = {'F_n': face_normal, 'F_d': face_dis, 'F_c': face_f, 'Rot1': p_green_R, 'Rot1_f': f_green_R.detach(),
                           'Rot2': p_red_R, 'Rot2_f': f_red_R.detach(), 'Tran': Pred_T, 'Size': Pred_s}
        gt_recon_list = {'R': gt_R, 'T': gt_t, 'Size': gt_s, 'Mean_shape': mean_shape, 'Points': PC}
        recon_loss = self.loss_recon(self.name_recon_list, pred_recon_list, gt_recon_list, sym, obj_id)

        pred_geo_list = {'Rot1': p_green_R, 'Rot2': p_red_R, 'Tran': Pred_T, 'Size': Pred_s, 'Rot1_f': f_green_R.detach(),
                         'Rot2_f': f_red_R.detach()}
        gt_geo_list = {'Points': PC, 'R': gt_R, 'T': gt_t, 'Mean_shape': mean_shape}
        geo_loss = self.loss_geo(self.name_geo_list, pred_geo_list, gt_geo_list, sym)

        loss_dict = {'fsnet_loss': fsnet_loss, 'recon_loss': recon_loss, 'geo_loss': geo_loss, 'prop_loss': prop_loss}
        return output_dict, loss_dict

    graphed_posenet = None

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
