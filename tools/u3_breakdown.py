"""host-synchronised breakdown of the full training step (U3): where the 19 ms go.  python tools/u3_breakdown.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench
from hs_pose_amd.config import FLAGS
from hs_pose_amd.HSPose import HSPose
from hs_pose_amd.train import TrainDriver
from tools import gemm_tuning

gemm_tuning.enable()
dev = torch.device("cuda:0")
B, N = 16, 1028
FLAGS.reset(); FLAGS.train = 1
torch.manual_seed(0)
net = HSPose("PoseNet_only").to(dev).train()
drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
case = bench.u3_case(B, N, dev)
net.enable_graphed_posenet(case["PC"], case["obj_id"])
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(rec):
    torch.cuda.synchronize(); t = time.perf_counter()
    _, ld = net(do_loss=True, **case)
    if rec: t = tick("forward: augment + network + losses", t)
    total = sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) + sum(ld['prop_loss'].values())
    if rec: t = tick("sum of the terms", t)
    total.backward()
    if rec: t = tick("backward", t)
    drv.optimizer.clip_grad_norm_(5)
    drv.optimizer.step(); drv.scheduler.step(); drv.optimizer.zero_grad()
    if rec: t = tick("clip + optimizer", t)


for i in range(4):
    step(False)
n = 10
for i in range(n):
    step(True)
for k, v in acc.items():
    print(f"{k:45s} {1e3 * v / n:8.3f} ms")
print("total", 1e3 * sum(acc.values()) / n)

# augmentation alone (eager torch composition), host-synchronised
from hs_pose_amd.augment import data_augment
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    with torch.no_grad():
        data_augment(case["PC"], case["gt_R"], case["gt_t"], case["gt_s"], case["mean_shape"], case["sym"], case["aug_bb"],
                     case["aug_rt_t"], case["aug_rt_r"], case["model_point"], case["nocs_scale"], case["obj_id"])
torch.cuda.synchronize()
print(f"data_augment alone                            {1e3 * (time.perf_counter() - t0) / 20:8.3f} ms")
