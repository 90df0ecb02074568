"""GPU parity (through the C-ABI): the fused Ranger step + device-side gradient clipping against the
fixtures written by the reference optimizer, and its checkpoint compatibility."""
import numpy as np
import pytest
import torch

from conftest import golden
from test_solver_oracle import CASES

pytestmark = pytest.mark.gpu


def _params(ref, dev):
    return [torch.nn.Parameter(t.clone().to(dev)) for t in ref.opt_case_tensors(0)]


@pytest.mark.parametrize("name", list(CASES))
def test_fused_ranger_matches_reference(dev, ref, name):
    from hs_pose_amd.solver import Ranger, clip_grad_norm_
    g, c = golden(name), CASES[name]
    params = _params(ref, dev)
    opt = Ranger(params, lr=c["lr"], **c["kw"])
    for step in range(1, c["nsteps"] + 1):
        for p, gr in zip(params, ref.opt_case_tensors(step)):
            p.grad.copy_(gr.to(dev))                       # .grad is a view of the flat gradient buffer
        norm = clip_grad_norm_(opt, c["max_norm"])
        opt.step()
        if f"s{step}.norm" in g.files:
            assert abs(float(norm) - float(g[f"s{step}.norm"][0])) <= 1e-5 * float(norm)
            for i, p in enumerate(params):
                err = np.abs(p.detach().cpu().numpy() - g[f"s{step}.p{i}"]).max()
                assert err <= 2e-6, (name, step, i, err)      # fp32, fused multiply-adds and a different sum order
    for i, p in enumerate(params):
        st = opt.state[p]
        assert st["step"] == c["nsteps"]
        assert np.abs(st["exp_avg"].cpu().numpy() - g[f"final.m{i}"]).max() <= 2e-6
        assert np.abs(st["exp_avg_sq"].cpu().numpy() - g[f"final.v{i}"]).max() <= 2e-6
        assert np.abs(st["slow_buffer"].cpu().numpy() - g[f"final.slow{i}"]).max() <= 2e-6


def test_slow_buffer_is_taken_at_the_first_step(dev, ref):
    """weights loaded AFTER the optimizer was built (pretrained checkpoint without optimizer state) are what Lookahead
    interpolates towards at step k=6 -- the reference creates slow_buffer lazily at its first step (ranger2020.py:168-170)."""
    from hs_pose_amd.solver import Ranger, clip_grad_norm_
    g, c = golden("solver_ranger_default"), CASES["solver_ranger_default"]
    params = [torch.nn.Parameter(torch.randn_like(t).to(dev)) for t in ref.opt_case_tensors(0)]     # "random init"
    opt = Ranger(params, lr=c["lr"], **c["kw"])
    with torch.no_grad():
        for p, t in zip(params, ref.opt_case_tensors(0)):
            p.copy_(t.to(dev))                                 # "load_state_dict" after construction
    for step in range(1, c["nsteps"] + 1):
        for p, gr in zip(params, ref.opt_case_tensors(step)):
            p.grad.copy_(gr.to(dev))
        clip_grad_norm_(opt, c["max_norm"])
        opt.step()
    for i, p in enumerate(params):
        assert np.abs(p.detach().cpu().numpy() - g[f"s{c['nsteps']}.p{i}"]).max() <= 2e-6
        assert np.abs(opt.state[p]["slow_buffer"].cpu().numpy() - g[f"final.slow{i}"]).max() <= 2e-6


def test_gradients_replaced_by_autograd_are_picked_up(dev, ref):
    """a first backward after ``p.grad = None`` makes a fresh .grad tensor: step() copies it back into the flat buffer."""
    from hs_pose_amd.solver import Ranger
    a = _params(ref, dev)
    b = _params(ref, dev)
    oa, ob = Ranger(a, lr=1e-2), Ranger(b, lr=1e-2)
    grads = [t.to(dev) for t in ref.opt_case_tensors(1)]
    for p, gr in zip(a, grads):
        p.grad.copy_(gr)
    for p, gr in zip(b, grads):
        p.grad = None
        (p * gr).sum().backward()
    oa.step(); ob.step()
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    ob.zero_grad()
    assert all(float(p.grad.abs().max()) == 0.0 for p in b)


def test_state_dict_round_trip_with_reference_layout(dev, ref):
    """per-parameter step / exp_avg / exp_avg_sq / slow_buffer entries, as the reference Ranger saves them."""
    from hs_pose_amd.solver import Ranger
    c = CASES["solver_ranger_default"]
    g = golden("solver_ranger_default")
    pa = _params(ref, dev)
    oa = Ranger(pa, lr=c["lr"])
    for step in range(1, 7):
        for p, gr in zip(pa, ref.opt_case_tensors(step)):
            p.grad.copy_(gr.to(dev))
        oa.clip_grad_norm_(c["max_norm"]); oa.step()
    sd = oa.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq", "slow_buffer"}
    assert sd["param_groups"][0]["betas"] == (0.95, 0.999) and sd["param_groups"][0]["k"] == 6
    # a fresh optimizer on the same weights resumes from the checkpoint and lands on the reference's step 13
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ob = Ranger(pb, lr=c["lr"])
    ob.load_state_dict({"state": {k: {n: (v.cpu() if torch.is_tensor(v) else v) for n, v in s.items()} for k, s in sd["state"].items()},
                        "param_groups": sd["param_groups"]})
    for step in range(7, 14):
        for p, gr in zip(pb, ref.opt_case_tensors(step)):
            p.grad.copy_(gr.to(dev))
        ob.clip_grad_norm_(c["max_norm"]); ob.step()
    for i, p in enumerate(pb):
        assert np.abs(p.detach().cpu().numpy() - g[f"s13.p{i}"]).max() <= 2e-6


def test_sumsq_and_clip_coefficient(dev, ref):
    from hs_pose_amd import ops
    from hs_pose_amd._lib import lib
    for n in (1, 3, 4, 1000, 262147):
        x = ref.hash_tensor((n + 4,), 77 + n, 2.0).to(dev)[:n]          # 16-byte aligned start
        out = torch.empty(1, device=dev)
        wsb = lib().hsp_sumsq_workspace_bytes(n)
        ws = ops._ws(wsb, dev)
        ops._run("hsp_sumsq_f32", (ops._p(x), n, ops._p(out), ops._p(ws), wsb, ops._stream()))
        want = float((x.double() ** 2).sum())
        assert abs(float(out) - want) <= 1e-5 * want


def test_optimizer_on_the_network(dev, flags):
    """build_params -> build_optimizer -> build_lr_rate as engine/train.py:44-58 does; one clipped step moves every weight."""
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.solver import build_lr_rate, build_optimizer, clip_grad_norm_
    flags.train = 1
    net = HSPose("PoseNet_only").to(dev)
    opt = build_optimizer(net.build_params(training_stage_freeze=[]))
    sch = build_lr_rate(opt, total_iters=150 * 1500)
    assert abs(opt.param_groups[0]["lr"] - 1e-4 * 0.001) < 1e-12          # warm-up start
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    for p in net.parameters():
        p.grad.copy_(1e3 * torch.randn_like(p))            # (warm-up lr is 1e-7: keep the update above fp32 resolution)
    norm = clip_grad_norm_(opt, 1e9)
    opt.step(); sch.step()
    assert 1e6 < float(norm) < 1e7                          # ~ 1e3 * sqrt(9.7e6)
    moved = [k for k, v in net.named_parameters() if not torch.equal(v, before[k])]
    assert len(moved) == len(before)
    assert opt.param_groups[0]["lr"] > 1e-4 * 0.001


def test_train_driver_loop_semantics(dev, flags, tmp_path):
    """engine/train.py:72-123: NaN skip, clip after every backward, step / schedule / zero_grad every `accumulate`
    batches, checkpoint keys."""
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.train import TrainDriver
    flags.train = 0                                           # the small eval-mode module set is enough here
    net = HSPose("PoseNet_only").to(dev)
    drv = TrainDriver(net, total_iters=100, accumulate=2, check_nan=True)
    w = net.posenet.ts.conv1.weight
    w0 = w.detach().clone()

    def loss_of(scale):
        return scale * sum((p * p).sum() for p in net.parameters())

    assert drv.step(loss_of(float("nan"))) is False            # skipped: counters advance, nothing else
    assert drv.global_step == 1 and drv.skipped == 1 and torch.equal(w, w0) and float(w.grad.abs().max()) == 0.0
    assert drv.step(loss_of(1.0)) is True                      # global_step 1: accumulate only
    assert drv.global_step == 2 and torch.equal(w, w0) and float(w.grad.abs().max()) > 0.0
    # ... and was clipped IN PLACE like the reference's clip_grad_norm_ on accumulate-only iterations (train.py:103-104)
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters()))
    assert abs(float(total) - 5.0) < 1e-3
    lr_before = drv.optimizer.param_groups[0]["lr"]
    assert drv.step(loss_of(1.0)) is True                      # global_step 2: optimizer + scheduler + zero_grad
    assert not torch.equal(w, w0) and float(w.grad.abs().max()) == 0.0
    assert drv.optimizer.param_groups[0]["lr"] != lr_before
    ck = drv.checkpoint(seed=3, epoch=7)
    assert list(ck) == ['seed', 'epoch', 'posenet_state_dict', 'scheduler', 'optimizer']
    path = tmp_path / "model_07.pth"
    torch.save(ck, path)
    net2 = HSPose("PoseNet_only").to(dev)
    drv2 = TrainDriver(net2, total_iters=100, accumulate=2)
    assert drv2.load_checkpoint(torch.load(path, weights_only=False)) == 8     # the epoch to resume FROM (train.py:56)
    assert torch.equal(net2.posenet.ts.conv1.weight, w)
    assert drv2.optimizer.state[net2.posenet.ts.conv1.weight]["step"] == 1
    assert drv2.scheduler.last_epoch == drv.scheduler.last_epoch
