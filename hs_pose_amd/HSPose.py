"""Drop-in mirror of the reference's ``network/HSPose.py`` operator surface (HSPose.py:23-276).

Same constructor, ``forward`` keyword set, ``output_dict`` keys and ``build_params`` as the reference.
Built so far: the inference path (``FLAGS.train = 0``; evaluation/evaluate.py:91-106) and the
depth -> cloud entry (``PC is None``, HSPose.py:39-48).  The training-only branches -- on-device
augmentation (HSPose.py:185-256) and the four loss modules (``do_loss=True``, HSPose.py:84-181) -- are
SURVEY 8f-1 and raise NotImplementedError here rather than silently doing something else; training the
network itself (forward + backward of ``self.posenet``) is fully supported through ``PoseNet9D``.
"""
import torch
import torch.nn as nn

from .config import FLAGS
from .pc_sample import PC_sample
from .PoseNet9D import PoseNet9D


class HSPose(nn.Module):
    def __init__(self, train_stage):
        super(HSPose, self).__init__()
        self.posenet = PoseNet9D()
        self.train_stage = train_stage

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
            raise NotImplementedError(
                "HSPose.forward with FLAGS.train: on-device augmentation (HSPose.py:185-256) is SURVEY 8f-1 "
                "(next); call self.posenet(PC, obj_id) for the network forward/backward")

        recon, face_normal, face_dis, face_f, p_green_R, p_red_R, f_green_R, f_red_R, \
            Pred_T, Pred_s = self.posenet(PC, obj_id)

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

        if do_loss:
            raise NotImplementedError("HSPose.forward(do_loss=True): the loss modules (HSPose.py:84-181) are "
                                      "SURVEY 8f-1 (next)")
        return output_dict

    def build_params(self, training_stage_freeze=None):
        """HSPose.py:258-275: one param group, lr = FLAGS.lr * FLAGS.lr_pose.  (The reference's 'pose'
        freeze loop sets an attribute on a tuple and therefore freezes nothing; kept as a no-op.)"""
        return [{
            "params": filter(lambda p: p.requires_grad, self.posenet.parameters()),
            "lr": float(FLAGS.lr) * FLAGS.lr_pose,
        }]
