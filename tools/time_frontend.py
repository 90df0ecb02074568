"""Front / back end (rows a-14, f-4): PC_sample on B depth crops and generate_RT, wall time per call with the device
synchronised (the front end contains one D2H of B ints by design), next to the CPU oracle restatement on one image batch.
Run on the GPU box:  python tools/time_frontend.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
import torch


def wall_us(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


def main():
    import ref_cpu as oc
    from hs_pose_amd import ops
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.geom_utils import generate_RT
    from hs_pose_amd.pc_sample import PC_sample
    dev = torch.device("cuda:0")
    FLAGS.reset()
    out = []
    for B, H, W in ((16, 256, 256), (16, 480, 640)):
        mask, depth, camK, coor = oc.frontend_inputs(B, H, W, 5, [60 + 3 * b for b in range(B)])
        md, dd, kd, cd = (t.to(dev) for t in (mask, depth, camK, coor))
        np.random.seed(0)
        t_all = wall_us(lambda: PC_sample(md, dd, kd, cd))
        m2, d2 = md.reshape(B, H * W), dd.reshape(B, H * W)
        t_compact = wall_us(lambda: ops.pc_compact(m2, d2))
        rng = np.random.RandomState(0)
        t0 = time.perf_counter()
        for b in range(B):
            oc.pc_sample(mask[b:b + 1], depth[b:b + 1], camK[b:b + 1], coor[b:b + 1], int(FLAGS.random_points), rng)
        t_cpu = 1e6 * (time.perf_counter() - t0)
        out.append({"op": "PC_sample", "B": B, "H": H, "W": W, "points": int(FLAGS.random_points), "us_per_call": round(t_all, 1),
                    "compact_kernel_us": round(t_compact, 1), "compact_GBps": round(B * H * W * 12 / t_compact / 1e3, 1),
                    "images_per_s": round(B / t_all * 1e6, 0), "cpu_oracle_images_per_s": round(B / t_cpu * 1e6, 1)})
    for B in (6, 64):
        g = torch.Generator().manual_seed(B)
        pg = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(dev)
        pr = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(dev)
        fg, fr = torch.rand(B, generator=g).to(dev), torch.rand(B, generator=g).to(dev)
        T = torch.randn(B, 3, generator=g).to(dev)
        sym = torch.zeros(B, 4); sym[::2, 0] = 1
        symd = sym.to(dev)
        t = wall_us(lambda: generate_RT([pg, pr], [fg, fr], T, mode='vec', sym=symd))
        t0 = time.perf_counter()
        for _ in range(10):
            oc.generate_rt(pg.cpu(), pr.cpu(), fg.cpu(), fr.cpu(), T.cpu(), sym)
        t_cpu = 1e6 * (time.perf_counter() - t0) / 10
        out.append({"op": "generate_RT", "B": B, "us_per_call": round(t, 1), "cpu_oracle_us_per_call": round(t_cpu, 1)})
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
