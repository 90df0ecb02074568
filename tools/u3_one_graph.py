"""the whole training step as ONE hipGraph, replayed 10 times (trace it: tools/trace_one.sh / rocprofv3 --kernel-trace --memory-copy-trace)"""
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
if os.environ.get("ONE_THREAD") == "1":  # (debug)
    torch.set_num_threads(1)
B, N = 16, 1028
FLAGS.reset(); FLAGS.train = 1
torch.manual_seed(0)
net = HSPose("PoseNet_only").to(dev).train()
drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
case = bench.u3_case(B, N, dev)
gs = GraphedTrainStep(net, drv.optimizer, case, scheduler=drv.scheduler, warmup=3)
torch.cuda.synchronize()
for phase in ("replay only", "full run()"):
    t0 = time.perf_counter()
    for _ in range(10):
        if phase == "replay only":
            gs.graph.replay()
        else:
            gs.run()
    torch.cuda.synchronize()
    print(f"{phase}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms", flush=True)
def wall(name, body, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    print(f"wall {name}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms", flush=True)
wall("replay only", gs.graph.replay)
wall("host draws + replay", lambda: (gs._host_draws(), gs.graph.replay()))
wall("run() = draws + replay + fused Ranger launch + schedule", gs.run)
