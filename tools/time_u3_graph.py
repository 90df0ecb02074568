"""unit U3 two ways on one box: the bench's form (network graphed, losses / augmentation / optimizer eager) and the whole step as ONE
hipGraph (graph.GraphedTrainStep) -- ms per step each (run on the GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from hs_pose_amd.config import FLAGS
from hs_pose_amd.HSPose import HSPose
from hs_pose_amd.train import TrainDriver
from hs_pose_amd.graph import GraphedTrainStep
dev = torch.device("cuda:0")
B, N = 16, 1028
print("bench form:", bench.u3_full_step(B, N, dev, steps=20, warmup=5))
FLAGS.reset(); FLAGS.train = 1
torch.manual_seed(0)
net = HSPose("PoseNet_only").to(dev).train()
drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
case = bench.u3_case(B, N, dev)
t0 = time.perf_counter()
gs = GraphedTrainStep(net, drv.optimizer, case, scheduler=drv.scheduler, warmup=3)
torch.cuda.synchronize()
print(f"capture {time.perf_counter() - t0:.1f} s")
for _ in range(5):
    gs.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    gs.run()
torch.cuda.synchronize()
print(f"one graph: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per step, total loss {float(gs.total):.4f}")
