"""Train-step driver semantics of the reference's ``engine/train.py:72-130``, around the fused
optimizer of ``hs_pose_amd.solver``:

  total_loss NaN -> the iteration is skipped (counters advance, nothing else happens)       train.py:91-95
  backward; clip_grad_norm_(network.parameters(), 5) after EVERY backward -- on iterations that only accumulate the
  coefficient is applied to the accumulated gradients in place, like the reference's in-place clip; on stepping
  iterations it is folded into the fused optimizer step                                      train.py:98-99,103-104
  optimizer.step(); scheduler.step(); optimizer.zero_grad() when global_step % accumulate == 0   train.py:97-102
  checkpoint = {'seed','epoch','posenet_state_dict','scheduler','optimizer'}                 train.py:115-123

``TrainDriver.step(total_loss)`` is the body of that loop for one batch.  With ``check_nan=False`` the NaN test
(the loop's only host synchronisation besides logging) is skipped.
"""
import math

import torch
import torch.distributed as dist

from .config import FLAGS
from .parallel import mean_flat_gradients
from .solver import build_lr_rate, build_optimizer


class TrainDriver:
    def __init__(self, network, optimizer=None, scheduler=None, total_iters=None, accumulate=None, max_norm=5,
                 check_nan=True, global_step=0, data_parallel=None):
        """data_parallel (default: whenever torch.distributed is initialised with more than one rank): after every
        backward the gradients -- already in the optimizer's flat buffers -- are averaged over the ranks with one RCCL
        all-reduce per parameter group, before clipping (the reference trains on a single device, train.py:23)."""
        self.network = network
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.data_parallel = bool(data_parallel)
        self.optimizer = optimizer if optimizer is not None else build_optimizer(network.build_params(training_stage_freeze=[]))
        if scheduler is None:
            if total_iters is None:
                acc = int(accumulate if accumulate is not None else getattr(FLAGS, "accumulate", 1))
                total_iters = int(getattr(FLAGS, "train_steps", 1500)) * int(getattr(FLAGS, "total_epoch", 150)) // acc   # train.py:51
            scheduler = build_lr_rate(self.optimizer, total_iters=total_iters)
        self.scheduler = scheduler
        self.accumulate = int(accumulate if accumulate is not None else getattr(FLAGS, "accumulate", 1))
        self.max_norm = max_norm
        self.check_nan = check_nan
        self.global_step = int(global_step)
        self.skipped = 0

    def step(self, total_loss):
        """one batch of engine/train.py's loop; returns False when the batch was skipped for a NaN loss."""
        if self.check_nan and math.isnan(float(total_loss.detach())):
            print('Found nan in total loss')
            self.global_step += 1
            self.skipped += 1
            return False
        total_loss.backward()
        if self.data_parallel:                                 # one process per GPU: gradient mean over the ranks
            self.optimizer.sync_grads()                        # (a backward may have re-created .grad outside the flat buffer)
            mean_flat_gradients([fg.flat_g for fg in self.optimizer._flat])
        self.optimizer.clip_grad_norm_(self.max_norm)
        if self.global_step % self.accumulate == 0:
            self.optimizer.step()
            self.scheduler.step()
            self.optimizer.zero_grad()
        else:
            self.optimizer.scale_grads_by_clip_()              # accumulate-only iteration: clip in place (train.py:103-104)
        self.global_step += 1
        return True

    def checkpoint(self, seed, epoch):
        """the dict engine/train.py:115-123 passes to torch.save (same keys, same sub-layouts)."""
        return {
            'seed': seed,
            'epoch': epoch,
            'posenet_state_dict': self.network.state_dict(),
            'scheduler': self.scheduler.state_dict(),
            'optimizer': self.optimizer.state_dict(),
        }

    def load_checkpoint(self, ckpt):
        """resume (engine/train.py:53-59: model weights, optimizer and scheduler state when present); returns the epoch
        to START from, i.e. the checkpoint's epoch + 1 like the reference's ``s_epoch``."""
        self.network.load_state_dict(ckpt['posenet_state_dict'])
        if 'optimizer' in ckpt:
            self.optimizer.load_state_dict(ckpt['optimizer'])
        if 'scheduler' in ckpt:
            self.scheduler.load_state_dict(ckpt['scheduler'])
        return ckpt.get('epoch', -1) + 1
