"""DOT dump of the captured step graph (node kinds and order): python tools/graph_dump.py out.dot"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hs_pose_amd.config import FLAGS
from hs_pose_amd.FaceRecon import FaceRecon
from hs_pose_amd import graph as G
from bench import make_inputs
dev = torch.device("cuda:0")
FLAGS.reset(); FLAGS.train = 0
torch.manual_seed(0)
net = FaceRecon().to(dev).train()
c, o, d = make_inputs(16, 1028, dev)
orig = torch.cuda.CUDAGraph
class Dbg(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls, *a, **k)
        return g
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.enable_debug_mode()
torch.cuda.CUDAGraph = Dbg
gs = G.GraphedStep(net, c, o, d)
gs.graph.debug_dump(sys.argv[1])
txt = open(sys.argv[1]).read()
import re
kinds = re.findall(r'label="?\{?\s*([A-Za-z_]+)', txt)
from collections import Counter
print(Counter(kinds).most_common(12))
for m in re.finditer(r'[^\n]*(?i:memcpy|memset)[^\n]*', txt):
    print(m.group(0)[:300])
