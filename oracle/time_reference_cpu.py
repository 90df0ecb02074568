"""
oracle/time_reference_cpu.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

The CPU timing SURVEY.md section 8d(1) / BASELINE.md section 3 asks for: the IMPORTED reference (/root/reference, stubs under
oracle/stubs/) on this container's host cores, torch.set_num_threads(8), the bench's synthetic inputs (B = 16, N = 1028):
  U1  HS stack (FaceRecon with FLAGS.train = 0 -> feat) forward + backward from a given dfeat
  U3  HSPose.forward(do_loss=True) + backward + clip_grad_norm_(5) + Ranger step (engine/train.py:72-110)
>= 3 timed repetitions each after one warm-up; prints one JSON line and (with --write) refreshes the table in BASELINE.md.

usage:  python oracle/time_reference_cpu.py [--reps 3] [--threads 8]
"""
import argparse
import json
import os
import platform
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import torch

import config.config  # noqa: F401  (reference flag definitions)
from absl import flags

FLAGS = flags.FLAGS


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def time_u1(B, N, reps):
    FLAGS.train = 0
    from network.fs_net_repo.FaceRecon import FaceRecon
    torch.manual_seed(0)
    net = FaceRecon().train()
    g = torch.Generator().manual_seed(0)
    pc = torch.randn(B, N, 3, generator=g) * 0.05
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    dfeat = torch.randn(B, N, 1286, generator=g)
    ts = []
    for r in range(reps + 1):
        net.zero_grad(set_to_none=True)
        t0 = time.perf_counter()
        out = net(pc, obj)
        feat = out[-1] if isinstance(out, (tuple, list)) else out
        feat.backward(dfeat)
        ts.append(time.perf_counter() - t0)
    return ts[1:]


def time_u3(B, N, reps):
    FLAGS.train = 1
    from network.HSPose import HSPose
    from tools.training_utils import build_lr_rate, build_optimizer
    import ref_cpu as oc
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").train()
    opt = build_optimizer(net.build_params(training_stage_freeze=[]))      # engine/train.py:45-48
    case = oc.hspose_train_case(B, N, 5000)
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        output_dict, loss_dict = net(**case, do_loss=True)
        total = sum(v for d in loss_dict.values() for v in d.values())
        opt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
        opt.step()
        ts.append(time.perf_counter() - t0)
    return ts[1:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--skip-u3", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    res = {"host": cpu_model(), "nproc": os.cpu_count(), "threads": a.threads, "torch": torch.__version__, "B": a.batch, "N": a.points,
           "reps": a.reps}
    u1 = time_u1(a.batch, a.points, a.reps)
    res["u1_s_per_step"] = [round(t, 3) for t in u1]
    res["u1_clouds_per_s"] = round(a.batch / (sum(u1) / len(u1)), 3)
    if not a.skip_u3:
        try:
            u3 = time_u3(a.batch, a.points, a.reps)
            res["u3_s_per_step"] = [round(t, 3) for t in u3]
            res["u3_clouds_per_s"] = round(a.batch / (sum(u3) / len(u3)), 3)
        except Exception as e:                                   # (kept visible: the U1 figure is still worth printing)
            res["u3_error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(res))


if __name__ == "__main__":
    main()
