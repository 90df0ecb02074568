"""List the ATen operators (shapes, calling line) one EAGER full training step (unit U3) issues besides the libhsp calls --
the glue left to fuse.  Run on the GPU box:  python tools/list_aten_ops_u3.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
from tools.list_aten_ops import Lister


def main():
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.train import TrainDriver
    dev = torch.device("cuda:0")
    FLAGS.reset(); FLAGS.train = 1
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev).train()
    drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
    case = bench.u3_case(16, 1028, dev)

    def fwd():
        _, ld = net(do_loss=True, **case)
        return sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) \
            + sum(ld['prop_loss'].values())
    for _ in range(2):
        drv.step(fwd())
    with Lister() as fw:
        total = fwd()
    with Lister() as bw:
        total.backward()
    with Lister() as op:
        drv.optimizer.clip_grad_norm_(5)
        drv.optimizer.step(); drv.scheduler.step(); drv.optimizer.zero_grad()
    for title, l in (("forward", fw), ("backward", bw), ("clip + optimizer", op)):
        print(f"== {title}: {sum(l.rows.values())} non-view ATen calls")
        for (name, shapes, where), n in sorted(l.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
            print(f"  {n:3d} x {name:22s} {where:22s} {shapes[:110]}")


if __name__ == "__main__":
    main()
