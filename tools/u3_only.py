"""the full training step (unit U3, bench.py::u3_full_step) alone -- the command tools/prof_train_step.sh profiles.
python tools/u3_only.py [steps] [warmup]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print(json.dumps(bench.u3_full_step(16, 1028, torch.device("cuda:0"), steps=steps, warmup=warmup)))
