"""List the ATen operators (with shapes and the calling line) that one eager HS-stack step issues besides the libhsp
calls -- candidates for fusion.  Run on the GPU box."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch
from torch.utils._python_dispatch import TorchDispatchMode

VIEW = ("view", "reshape", "_unsafe_view", "unsqueeze", "squeeze", "transpose", "expand", "slice", "select", "permute",
        "t", "detach", "alias", "as_strided", "unbind", "split", "view_as", "expand_as", "lift_fresh", "empty", "empty_like",
        "empty_strided", "new_empty", "_local_scalar_dense", "unsqueeze_", "squeeze_", "is_same_size", "stride", "size")


class Lister(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func.overloadpacket).replace("aten.", "")
        if name not in VIEW:
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            where = ""
            for fr in reversed(traceback.extract_stack()):
                if "hs_pose_amd" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            self.rows[(name, str(shapes), where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    sys.path.insert(0, ROOT)
    from bench import make_inputs
    dev = torch.device("cuda:0")
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(0)
    net = FaceRecon().to(dev).train()
    centred, obj, dfeat = make_inputs(16, 1028, dev)
    for _ in range(2):
        _, _, feat = net(centred, obj); feat.backward(dfeat)
    for p in net.parameters():
        p.grad = None
    with Lister() as fw:
        _, _, feat = net(centred, obj)
    with Lister() as bw:
        feat.backward(dfeat)
    for title, l in (("forward", fw), ("backward", bw)):
        print(f"== {title}: {sum(l.rows.values())} non-view ATen calls")
        for (name, shapes, where), n in sorted(l.rows.items(), key=lambda kv: (kv[0][2], kv[0][0])):
            print(f"  {n:3d} x {name:22s} {where:22s} {shapes[:110]}")


if __name__ == "__main__":
    main()
