"""
oracle/gen_golden_solver.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Golden vectors for the training-driver rows: the reference's Ranger optimizer
(tools/torch_utils/solver/ranger2020.py) preceded by torch.nn.utils.clip_grad_norm_ exactly as in
engine/train.py:96-103, run for 13 steps on closed-form parameters / gradients (ref_cpu.opt_case_tensors),
and the flat-and-anneal learning-rate factors (tools/torch_utils/solver/lr_scheduler.py:177-263 through
tools/training_utils.build_lr_rate).  oracle/ref_cpu.py is checked against them, outputs go to
tests/golden/solver_*.npz.

usage:  python oracle/gen_golden_solver.py
"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch

import config.config  # noqa: F401
from absl import flags

FLAGS = flags.FLAGS
from tools.torch_utils.solver.ranger2020 import Ranger as RefRanger
from tools.torch_utils.solver.lr_scheduler import flat_and_anneal_lr_scheduler as ref_sched

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    mpath = os.path.join(GOLD, "manifest.json")
    man = json.load(open(mpath))
    man["files"][name] = {k: [list(v.shape), str(v.dtype)] for k, v in arrs.items()}
    json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def run_case(name, lr, max_norm, nsteps, snaps, **kw):
    params = [torch.nn.Parameter(t.clone()) for t in oc.opt_case_tensors(0)]
    with redirect_stdout(io.StringIO()):
        opt = RefRanger(params, lr=lr, **kw)
    # oracle twin
    op = [t.clone() for t in oc.opt_case_tensors(0)]
    ost = [dict(exp_avg=torch.zeros_like(t), exp_avg_sq=torch.zeros_like(t), slow_buffer=t.clone()) for t in op]
    out = {}
    worst = 0.0
    for step in range(1, nsteps + 1):
        grads = oc.opt_case_tensors(step)
        for p, g in zip(params, grads):
            p.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm)
        opt.step()
        og = [g.clone() for g in grads]
        onorm = oc.clip_grads_(og, max_norm)
        assert abs(float(norm) - float(onorm)) <= 1e-6 * float(norm)
        oc.ranger_step_(op, og, ost, step, lr=lr, **{dict(N_sma_threshhold="n_sma_threshold").get(k, k): v for k, v in kw.items()})
        for i, (p, q) in enumerate(zip(params, op)):
            worst = max(worst, (p.detach() - q).abs().max().item())
        if step in snaps:
            for i, p in enumerate(params):
                out[f"s{step}.p{i}"] = p.detach().numpy().copy()
            out[f"s{step}.norm"] = np.array([float(norm)], np.float32)
        if step == nsteps:
            for i, p in enumerate(params):
                st = opt.state[p]
                out[f"final.m{i}"] = st["exp_avg"].numpy().copy()
                out[f"final.v{i}"] = st["exp_avg_sq"].numpy().copy()
                out[f"final.slow{i}"] = st["slow_buffer"].numpy().copy()
    print(f"  {name}: oracle vs reference after {nsteps} steps: max |dp| = {worst:.3e}")
    assert worst < 1e-6
    save(name, **out)


print("Ranger")
run_case("solver_ranger_default", lr=1e-2, max_norm=5.0, nsteps=13, snaps=(5, 6, 13))
run_case("solver_ranger_wd_gcafter", lr=1e-2, max_norm=1e9, nsteps=7, snaps=(7,), weight_decay=0.01, gc_loc=False)
run_case("solver_ranger_convonly", lr=1e-2, max_norm=50.0, nsteps=6, snaps=(6,), gc_conv_only=True, alpha=0.8, k=3)

print("lr schedule")
total = 150 * 1500
p = [torch.nn.Parameter(torch.zeros(1))]
opt = torch.optim.SGD(p, lr=1.0)
sch = ref_sched(opt, total_iters=total, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear",
                anneal_method="cosine", anneal_point=0.72, steps=(0.5, 0.75), target_lr_factor=0, poly_power=1.0, step_gamma=0.1)
f = sch.lr_lambdas[0]
xs = np.array([0, 1, 10, 500, 999, 1000, 1001, 50000, 161999, 162000, 162001, 200000, 224999, 225000], dtype=np.int64)
fac = np.array([f(int(x)) for x in xs], dtype=np.float64)
for x, v in zip(xs, fac):
    assert abs(oc.flat_and_anneal_factor(int(x), total) - v) < 1e-15
save("solver_lr_schedule", x=xs, factor=fac, total=np.array([total], np.int64))
print("done")
