"""Inference unit (not the headline metric): the timed body of the reference's evaluation loop
(evaluation/evaluate.py:90-106) -- HSPose.forward in eval mode on the n instances of one image + generate_RT -- on
synthetic clouds of N=1028 points, eagerly and as a hipGraph replay.  "images/s" is the reference's FPS definition
(one call per image, all instances batched), here with the device synchronised after every image.
Run on the GPU box:  python tools/bench_infer.py [--instances 1 4 6] [--images 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, nargs="+", default=[1, 4, 6])
    ap.add_argument("--images", type=int, default=200)
    ap.add_argument("--points", type=int, default=1028)
    args = ap.parse_args()
    from tools import gemm_tuning
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.geom_utils import generate_RT
    from hs_pose_amd.graph import GraphedInference
    from hs_pose_amd.HSPose import HSPose
    gemm_tuning.enable()
    dev = torch.device("cuda:0")
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev).eval()
    for n in args.instances:
        g = torch.Generator().manual_seed(n)
        PC = (torch.randn(n, args.points, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.8])).to(dev)
        obj = torch.randint(0, 6, (n,), generator=g).to(dev)
        mean_shape = (torch.rand(n, 3, generator=g) * 0.2 + 0.1).to(dev)
        sym = torch.zeros(n, 4, dtype=torch.int32)
        sym[::2, 0] = 1
        sym = sym.to(dev)

        @torch.no_grad()
        def eager():
            out = net(PC=PC, obj_id=obj, mean_shape=mean_shape, sym=sym)
            RT = generate_RT([out['p_green_R'], out['p_red_R']], [out['f_green_R'], out['f_red_R']], out['Pred_T'],
                             mode='vec', sym=sym)
            return RT, out['Pred_s'] + mean_shape

        graphed = GraphedInference(net, PC, obj, mean_shape, sym)
        res = {"instances_per_image": n, "points": args.points}
        for name, fn in (("eager", eager), ("hipgraph", graphed.run)):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.images):
                fn()
                torch.cuda.synchronize()           # one image at a time, result on the device
            dt = time.perf_counter() - t0
            res[name] = {"ms_per_image": round(1e3 * dt / args.images, 3), "images_per_s": round(args.images / dt, 1),
                         "clouds_per_s": round(n * args.images / dt, 1)}
        print(json.dumps(res))

    gemm_tuning.save()                           # only with HSP_TUNABLEOP_OUT set


if __name__ == "__main__":
    main()
