"""Run by test_gpu_train_graph.py in a fresh process (the HIP runtime reads DEBUG_CLR_GRAPH_PACKET_CAPTURE when it
starts): one GraphedTrainStep replay against the same step issued eagerly on a twin network --
all loss terms, every parameter gradient and the parameters after the Ranger step.
usage: python tests/_train_graph_check.py B N
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch

from hs_pose_amd import augment, gcn3d
from hs_pose_amd.config import FLAGS
from hs_pose_amd.graph import GraphedTrainStep
from hs_pose_amd.HSPose import HSPose
from hs_pose_amd.train import TrainDriver
import ref_cpu as oc

KEYS = ("PC", "obj_id", "gt_R", "gt_t", "gt_s", "mean_shape", "sym", "aug_bb", "aug_rt_t", "aug_rt_r", "model_point",
        "nocs_scale")


def make(dev):
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev).train()
    for m in net.modules():                      # dropout draws come from the device generator: not comparable
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net, TrainDriver(net, total_iters=1000, check_nan=False)


def main():
    B, N = int(sys.argv[1]), int(sys.argv[2])
    dev = torch.device("cuda:0")
    FLAGS.reset()
    FLAGS.train = 1
    FLAGS.aug_bb_pro = FLAGS.aug_rt_pro = FLAGS.aug_bc_pro = FLAGS.aug_pc_pro = -1.0   # device-generator draws off
    case = {k: v.to(dev) for k, v in oc.hspose_train_case(B, N, 7).items()}
    batch = {k: case[k] for k in KEYS}

    net_g, drv_g = make(dev)
    torch.manual_seed(3)                         # host draws (pool permutations) of the capture + first replay ...
    graphed = GraphedTrainStep(net_g, drv_g.optimizer, batch, scheduler=drv_g.scheduler, warmup=2)
    graphed.run()
    torch.cuda.synchronize()
    pool = [p.clone() for p in graphed.pool_idx]
    noise = graphed.noise.clone()
    grads_g = {k: p.grad.detach().clone() for k, p in net_g.named_parameters()}
    loss_g = {f"{g}.{k}": float(v) for g, d in graphed.loss_dict.items() for k, v in d.items()}

    net_e, drv_e = make(dev)                     # ... replayed into the eager twin
    with gcn3d.pool_index_feed(pool), augment.jitter_noise_feed(noise):
        _, ld = net_e(do_loss=True, **batch)
    total = sum(sum(d.values()) for d in ld.values())
    drv_e.optimizer.zero_grad()
    total.backward()
    grads_e = {k: p.grad.detach().clone() for k, p in net_e.named_parameters() if p.grad is not None}
    drv_e.optimizer.clip_grad_norm_(5)
    drv_e.optimizer.step()
    torch.cuda.synchronize()

    bad = []
    for g, d in ld.items():
        for k, v in d.items():
            a, b = float(v), loss_g[f"{g}.{k}"]
            if abs(a - b) > 1e-4 * max(1.0, abs(a)):
                bad.append(f"loss {g}.{k}: eager {a} graph {b}")
    gmax = max(v.abs().max().item() for v in grads_e.values())
    for k, v in grads_e.items():
        err = (v - grads_g[k]).abs().max().item()
        if err > 1e-4 * gmax:
            bad.append(f"grad {k}: |diff| {err:.3e} vs max|grad| {gmax:.3e}")
    pe = dict(net_e.named_parameters())
    for k, p in net_g.named_parameters():
        err = (p - pe[k]).abs().max().item()
        if err > 1e-5 * max(1.0, p.abs().max().item()):
            bad.append(f"param after step {k}: |diff| {err:.3e}")
    print(f"B={B} N={N} total loss eager {float(total):.6f} graph {float(graphed.total):.6f}; {len(bad)} mismatches")
    for line in bad[:20]:
        print("  " + line)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
