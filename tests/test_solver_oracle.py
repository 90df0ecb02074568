"""CPU: the oracle's training-driver restatements (gradient clipping, Ranger step, lr schedule) against the
fixtures written by the reference (oracle/gen_golden_solver.py)."""
import numpy as np
import pytest
import torch

from conftest import golden

CASES = {
    "solver_ranger_default": dict(lr=1e-2, max_norm=5.0, nsteps=13, kw={}),
    "solver_ranger_wd_gcafter": dict(lr=1e-2, max_norm=1e9, nsteps=7, kw=dict(weight_decay=0.01, gc_loc=False)),
    "solver_ranger_convonly": dict(lr=1e-2, max_norm=50.0, nsteps=6, kw=dict(gc_conv_only=True, alpha=0.8, k=3)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_ranger_oracle_matches_reference(ref, name):
    g, c = golden(name), CASES[name]
    p = [t.clone() for t in ref.opt_case_tensors(0)]
    st = [dict(exp_avg=torch.zeros_like(t), exp_avg_sq=torch.zeros_like(t), slow_buffer=t.clone()) for t in p]
    for step in range(1, c["nsteps"] + 1):
        grads = ref.opt_case_tensors(step)
        norm = ref.clip_grads_(grads, c["max_norm"])
        ref.ranger_step_(p, grads, st, step, lr=c["lr"], **c["kw"])
        if f"s{step}.norm" in g.files:
            assert abs(float(norm) - float(g[f"s{step}.norm"][0])) <= 1e-5 * float(norm)
            for i, t in enumerate(p):
                assert np.abs(t.numpy() - g[f"s{step}.p{i}"]).max() <= 1e-6, (step, i)
    for i, s_ in enumerate(st):
        assert np.abs(s_["exp_avg"].numpy() - g[f"final.m{i}"]).max() <= 1e-6
        assert np.abs(s_["exp_avg_sq"].numpy() - g[f"final.v{i}"]).max() <= 1e-6
        assert np.abs(s_["slow_buffer"].numpy() - g[f"final.slow{i}"]).max() <= 1e-6


def test_lr_schedule_matches_reference(ref):
    from hs_pose_amd.solver import flat_and_anneal_factor
    g = golden("solver_lr_schedule")
    total = int(g["total"][0])
    for x, want in zip(g["x"], g["factor"]):
        assert abs(ref.flat_and_anneal_factor(int(x), total) - want) < 1e-15
        got = flat_and_anneal_factor(int(x), total, warmup_iters=1000, warmup_factor=0.001, warmup_method="linear",
                                     anneal_point=0.72, anneal_method="cosine", steps=(0.5, 0.75))
        assert abs(got - want) < 1e-15
    assert g["factor"][0] == 0.001 and g["factor"][-1] < 1e-12 and g["factor"][7] == 1.0


def test_radam_rectification_switches_at_step_six():
    from hs_pose_amd.solver import _radam_step_size
    flags = [_radam_step_size(s, 0.95, 0.999, 5)[0] for s in range(1, 9)]
    assert flags == [False] * 5 + [True] * 3
    assert abs(_radam_step_size(1, 0.95, 0.999, 5)[1] - 1 / (1 - 0.95)) < 1e-12
