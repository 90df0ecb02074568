"""which Python line issues the device-to-device copies of one eager HS-stack forward+backward (torch profiler, with stacks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = FaceRecon().to(dev).train()
c, o, d = make_inputs(16, 1028, dev)
from hs_pose_amd import gcn3d
from hs_pose_amd.graph import alloc_pool_indices, upload_pool_indices
pidx = alloc_pool_indices(1028, dev)
upload_pool_indices(pidx, 1028)
for _ in range(3):
    with gcn3d.pool_index_feed(pidx):
        _, _, f = net(c, o)
    f.backward(d)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with gcn3d.pool_index_feed(pidx):
        _, _, f = net(c, o)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60)[:200])
for e in prof.events():
    n = e.name
    if n in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::sum", "aten::mul", "aten::threshold_backward", "aten::relu", "aten::relu_") and e.device_time_total > 0 or "emcpy" in n:
        st = [s for s in (e.stack or []) if "hs_pose_amd" in s or "bench" in s][:3]
        print(f"{n:28s} cuda {e.device_time_total:7.1f} us  shapes {e.input_shapes if hasattr(e, 'input_shapes') else ''}  {st}")
