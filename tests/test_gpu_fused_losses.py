"""GPU: the fused loss kernels (hsp_pose_losses_fwd / _bwd, hs_pose_amd/fused_losses.py) against the fixtures written by
the reference's loss modules (oracle/gen_golden_losses.py -> tests/golden/losses_*.npz: the 19 weighted terms and the
gradient of their sum w.r.t. every network output) and against the torch-op statement of the same formulas
(hs_pose_amd/losses.py, itself pinned by the same fixtures on the CPU) on larger random batches and with unequal term
weights.  Tolerances as in test_losses.py's GPU branch: the plane-fit terms go through an ill-conditioned 3x3 solve."""
import numpy as np
import pytest
import torch

from conftest import golden
from test_losses import LOSS_KEYS, run_losses

pytestmark = pytest.mark.gpu
NET = ("recon", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s")


def fused(gt, pred):
    from hs_pose_amd.fused_losses import pose_losses
    return pose_losses(pred, gt["PC"], gt["gt_R"], gt["gt_t"], gt["gt_s"], gt["mean_shape"], gt["sym"], gt["obj_id"])


def case(ref, dev, n_points=96, seed=4000, repeat=1):
    gt, pred = ref.loss_case(n_points, seed)
    if repeat > 1:                                  # more clouds: the 7 symmetry classes, repeated with other noise
        gts, preds = [gt], [pred]
        for i in range(1, repeat):
            g2, p2 = ref.loss_case(n_points, seed + 100 * i)
            gts.append(g2); preds.append(p2)
        gt = {k: torch.cat([g[k] for g in gts]) for k in gt}
        pred = {k: torch.cat([p[k].detach() for p in preds]) for k in pred}
    gt = {k: v.to(dev) for k, v in gt.items()}
    pred = {k: v.detach().to(dev).requires_grad_(True) for k, v in pred.items()}
    return gt, pred


@pytest.mark.parametrize("name,loss_type", [("losses_l1", "l1"), ("losses_smoothl1", "smoothl1")])
def test_fused_losses_match_reference_fixture(dev, ref, flags, name, loss_type):
    flags.fsnet_loss_type = loss_type
    g = golden(name)
    gt, pred = case(ref, dev)
    ld = fused(gt, pred)
    assert {k: list(v) for k, v in ld.items()} == LOSS_KEYS
    for grp, d in ld.items():
        for k, v in d.items():
            want = g[f"{grp}.{k}"]
            got = v.detach().cpu().numpy().reshape(-1)
            assert got.shape == want.shape, (grp, k, got.shape, want.shape)
            assert np.abs(got - want).max() <= 1e-3 * max(np.abs(want).max(), 1e-2), (grp, k, got, want)
    total = sum(sum(d.values()) for d in ld.values())
    assert abs(float(total.detach()) - float(g["total"][0])) <= 1e-3
    total.backward()
    for k, v in pred.items():
        want = g["grad." + k]
        err = np.abs(v.grad.cpu().numpy() - want).max()
        tol = 5e-3 if k in ("face_normal", "face_dis") else 2e-4
        assert err <= tol * max(1.0, np.abs(want).max()), (k, err)


@pytest.mark.parametrize("loss_type,n_points,repeat,weighted", [("l1", 1028, 2, False), ("smoothl1", 257, 3, True),
                                                                 ("l1", 64, 1, True)])
def test_fused_losses_match_torch_composition(dev, ref, flags, loss_type, n_points, repeat, weighted):
    """the same batch through hs_pose_amd/losses.py (autograd) and through the fused kernels; with `weighted` the objective
    is a random positive combination of the 19 terms (the backward kernels take d(objective)/d(term))"""
    from hs_pose_amd import HSPose as H
    flags.fsnet_loss_type = loss_type
    gt, pred = case(ref, dev, n_points, 4300, repeat)
    pred2 = {k: v.detach().clone().requires_grad_(True) for k, v in pred.items()}
    ld_f = fused(gt, pred)
    # losses.py wired as HSPose.forward wires it
    names = H.control_loss('PoseNet_only')
    g_green, g_red = H.get_gt_v(gt["gt_R"])
    p, sym, PC = pred2, gt["sym"], gt["PC"]
    axes = {'Rot1': p['p_green_R'], 'Rot2': p['p_red_R']}
    conf = {'Rot1_f': p['f_green_R'], 'Rot2_f': p['f_red_R']}
    conf_c = {k: v.detach() for k, v in conf.items()}
    pose = {'Tran': p['Pred_T'], 'Size': p['Pred_s']}
    gt_pose = {'Points': PC, 'R': gt["gt_R"], 'T': gt["gt_t"], 'Mean_shape': gt["mean_shape"]}
    ld_t = {
        'fsnet_loss': H.fs_net_loss()(names[0], {**axes, **conf, **pose, 'Recon': p['recon']},
                                      {'Rot1': g_green, 'Rot2': g_red, 'Recon': PC, 'Tran': gt["gt_t"], 'Size': gt["gt_s"]}, sym),
        'recon_loss': H.recon_6face_loss()(names[1], {**axes, **conf_c, **pose, 'F_n': p['face_normal'], 'F_d': p['face_dis'],
                                                      'F_c': p['face_f']}, {**gt_pose, 'Size': gt["gt_s"]}, sym, gt["obj_id"]),
        'geo_loss': H.geo_transform_loss()(names[2], {**axes, **conf_c, **pose}, gt_pose, sym),
        'prop_loss': H.prop_rot_loss()(names[3], {**axes, **conf_c, 'Recon': p['recon'], 'Tran': p['Pred_T'], 'Scale': p['Pred_s']},
                                       gt_pose, sym),
    }
    gen = torch.Generator().manual_seed(5)
    tot_f = tot_t = 0.0
    for grp, keys in LOSS_KEYS.items():
        for k in keys:
            a, b = ld_f[grp][k], ld_t[grp][k]
            assert a.shape == b.shape, (grp, k)
            scale = max(abs(float(b.detach().sum())), 1e-2)
            assert abs(float(a.detach().sum()) - float(b.detach().sum())) <= 1e-3 * scale, (grp, k, a, b)
            w = float(0.25 + 2 * torch.rand((), generator=gen)) if weighted else 1.0
            tot_f = tot_f + w * a.sum()
            tot_t = tot_t + w * b.sum()
    tot_f.backward()
    tot_t.backward()
    for k in NET:
        ga, gb = pred[k].grad, pred2[k].grad
        tol = 5e-3 if k in ("face_normal", "face_dis") else 2e-4
        err = float((ga - gb).abs().max())
        assert err <= tol * max(1.0, float(gb.abs().max())), (k, err, float(gb.abs().max()))


def test_fused_losses_reproducible(dev, ref, flags):
    gt, pred = case(ref, dev, 1028, 4400, 2)
    outs = []
    for _ in range(2):
        for v in pred.values():
            v.grad = None
        ld = fused(gt, pred)
        sum(sum(d.values()) for d in ld.values()).backward()
        outs.append([v.grad.clone() for v in pred.values()] + [torch.stack([x.reshape(()) for d in ld.values() for x in d.values()])])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_fused_total_equals_the_term_by_term_sum(dev, ref, flags):
    """HSPose.total_loss(loss_dict) -- the kernels' own reduction over the 19 terms -- against the sum engine/train.py:84-90
    spells out: value, the gradient w.r.t. every network output (also when terms AND the total are both used), and the
    plain-dict fallback"""
    from hs_pose_amd.fused_losses import total_loss, LossDict
    res = []
    for mode in ("terms", "total", "both"):
        gt, pred = case(ref, dev, n_points=128, seed=4100, repeat=2)
        ld = fused(gt, pred)
        assert isinstance(ld, LossDict) and {k: list(v) for k, v in ld.items()} == LOSS_KEYS
        by_terms = sum(sum(d.values()) for d in ld.values())
        total = {"terms": by_terms, "total": total_loss(ld), "both": 0.5 * by_terms + 0.5 * total_loss(ld)}[mode]
        total = total.reshape(())
        grads = torch.autograd.grad(total, [pred[k] for k in NET])
        res.append((total.detach(), grads))
    for tot, grads in res[1:]:
        assert abs(float(tot) - float(res[0][0])) <= 2e-6 * abs(float(res[0][0]))
        for a, b in zip(grads, res[0][1]):
            assert (a - b).abs().max().item() <= 2e-6 * max(b.abs().max().item(), 1e-6)
    plain = {k: dict(v) for k, v in ld.items()}
    assert abs(float(total_loss(plain)) - float(res[0][0])) <= 2e-6 * abs(float(res[0][0]))
    # an EDITED dict (an ablation drops, re-weights or adds a term) must be summed as it stands, not answered from the cached total
    full = float(res[0][0])
    gt, pred = case(ref, dev, n_points=128, seed=4100, repeat=2)
    ld = fused(gt, pred)
    dropped = float(ld['geo_loss'].pop(next(iter(ld['geo_loss']))))
    assert abs(float(total_loss(ld)) - (full - dropped)) <= 2e-6 * abs(full)
    ld = fused(gt, pred)
    key = next(iter(ld['fsnet_loss']))
    old = float(ld['fsnet_loss'][key])
    ld['fsnet_loss'][key] = 3.0 * ld['fsnet_loss'][key]
    assert abs(float(total_loss(ld)) - (full + 2.0 * old)) <= 2e-6 * abs(full)
    ld = fused(gt, pred)
    ld['extra'] = {'reg': torch.tensor(1.5, device=dev)}
    assert abs(float(total_loss(ld)) - full) <= 2e-6 * abs(full)        # (the reference's sum names its four groups: 'extra' is not in it)
    ld = fused(gt, pred)
    assert total_loss(ld) is ld.total                                      # untouched: the kernels' own reduction


@pytest.mark.parametrize("B,N", [(16, 1028), (3, 77)])
def test_face_split_matches_torch_composition(dev, ref, monkeypatch, B, N):
    """PoseNet9D.py:31-35 (three slices, per-face normalisation, sigmoid) as one launch each way against the torch composition:
    the three outputs and the gradient of the (B, N, 30) face-head output, also with one output unused"""
    from hs_pose_amd import PoseNet9D as P9
    face0 = ref.hash_tensor((B, N, 30), 901, 1.0).to(dev)
    ups = [ref.hash_tensor(sh, 902 + i, 1.0).to(dev) for i, sh in enumerate(((B, N, 6, 3), (B, N, 6), (B, N, 6)))]
    res = []
    for fused in ("1", "0"):
        monkeypatch.setattr(P9, "FUSED_FACE_SPLIT", fused == "1")
        face = face0.clone().requires_grad_(True)
        outs = P9._split_face_head(face)
        assert [tuple(o.shape) for o in outs] == [(B, N, 6, 3), (B, N, 6), (B, N, 6)]
        (g,) = torch.autograd.grad(outs, [face], ups)
        outs2 = P9._split_face_head(face)
        (g2,) = torch.autograd.grad([outs2[0], outs2[2]], [face], [ups[0], ups[2]])      # the distances unused
        res.append([o.detach() for o in outs] + [g, g2])
    for a, b in zip(*res):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())


def test_axis_conf_matches_torch_composition(dev, ref, monkeypatch):
    """PoseNet9D.py:40-46 (axis = h[:, 1:] / (norm + 1e-6), confidence = sigmoid(h[:, 0])) as one launch each way against the torch
    composition: outputs and the gradient of the head output, incl. a zero axis row and one output unused"""
    from hs_pose_amd import PoseNet9D as P9
    h0 = ref.hash_tensor((16, 4), 911, 1.0).to(dev)
    h0[3, 1:] = 0.0
    ups = [ref.hash_tensor((16, 3), 912, 1.0).to(dev), ref.hash_tensor((16,), 913, 1.0).to(dev)]
    res = []
    for fused in ("1", "0"):
        monkeypatch.setattr(P9, "FUSED_FACE_SPLIT", fused == "1")
        h = h0.clone().requires_grad_(True)
        outs = P9._axis_and_confidence(h)
        (g,) = torch.autograd.grad(outs, [h], ups)
        (g2,) = torch.autograd.grad(P9._axis_and_confidence(h)[0], [h], ups[0])
        res.append([o.detach() for o in outs] + [g, g2])
    for a, b in zip(*res):
        assert a.shape == b.shape and (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())


def test_hspose_forward_uses_fused_losses(dev, flags):
    """HSPose.forward(do_loss=True) on a device batch returns the fused terms (same keys; finite; backward reaches the
    network) and instances with ``fused_losses = False`` agree with it"""
    from hs_pose_amd.HSPose import HSPose
    import bench
    flags.train = 1
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev).train()
    case_ = bench.u3_case(4, 256, dev) if hasattr(bench, "u3_case") else None
    if case_ is None:
        pytest.skip("bench.u3_case not available")
    torch.manual_seed(1)
    _, ld = net(do_loss=True, **case_)
    assert {k: list(v) for k, v in ld.items()} == LOSS_KEYS
    total = sum(sum(d.values()) for d in ld.values())
    assert torch.isfinite(total).all()
    total.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.posenet.parameters() if p.requires_grad)


@pytest.mark.parametrize("pro,n_points,repeat", [(0.6, 64, 1), (1.1, 1028, 3), (0.35, 257, 4), (-1.0, 64, 1)])
def test_fused_augmentation_matches_torch_composition(dev, ref, flags, pro, n_points, repeat):
    """hsp_pose_augment (one launch) against the torch-op composition of hs_pose_amd/augment.py -- itself pinned on the CPU
    by the reference-written fixture losses_augment.npz -- under the same seed: both consume the device generator (six
    draws) and the CPU generator (jitter noise) in the same order, so every cloud takes the same branches and factors."""
    from hs_pose_amd import augment
    flags.aug_bb_pro = flags.aug_rt_pro = flags.aug_bc_pro = flags.aug_pc_pro = pro
    gts = [ref.augment_case(n_points, 4100 + 50 * i) for i in range(repeat)]
    gt = {k: torch.cat([g[k] for g in gts]).to(dev) for k in gts[0]}
    args = (gt["PC"], gt["gt_R"], gt["gt_t"], gt["gt_s"], gt["mean_shape"], gt["sym"], gt["aug_bb"], gt["aug_rt_t"],
            gt["aug_rt_r"], gt["model_point"], gt["nocs_scale"], gt["obj_id"])
    outs = []
    for fused in (True, False):
        torch.manual_seed(11)
        old, augment.FUSED = augment.FUSED, fused
        try:
            outs.append(augment.data_augment(*[a.clone() for a in args]))
        finally:
            augment.FUSED = old
    for name, a, b in zip(("PC", "R", "t", "s"), *outs):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 5e-6, (name, float((a - b).abs().max()))
    if pro > 1.0:                                   # every augmentation fired: the clouds did move
        assert float((outs[0][0] - gt["PC"]).abs().max()) > 1e-3


def _subset(gt, pred, rows):
    sel = lambda d: {k: v.detach()[rows].clone() for k, v in d.items()}
    g, p = sel(gt), sel(pred)
    return g, {k: v.requires_grad_(True) for k, v in p.items()}


@pytest.mark.parametrize("rows,n_points", [([2], 96), ([0, 5], 5), ([1, 1, 1, 6, 6], 33)])
def test_fused_losses_small_and_odd_batches(dev, ref, flags, rows, n_points):
    """one cloud, five points, repeated symmetry classes (all clouds rotationally symmetric / none): the batch-level
    rescaling B / #kept and the per-class masks, against the torch composition"""
    from hs_pose_amd import HSPose as H
    gt, pred = case(ref, dev, n_points, 4500)
    gt, pred = _subset(gt, pred, rows)
    pred2 = {k: v.detach().clone().requires_grad_(True) for k, v in pred.items()}
    ld_f = fused(gt, pred)
    net = H.HSPose.__new__(H.HSPose)                     # only the loss wiring of HSPose.forward is needed
    torch.nn.Module.__init__(net)
    names = H.control_loss('PoseNet_only')
    g_green, g_red = H.get_gt_v(gt["gt_R"])
    p, sym, PC = pred2, gt["sym"], gt["PC"]
    axes = {'Rot1': p['p_green_R'], 'Rot2': p['p_red_R']}
    conf = {'Rot1_f': p['f_green_R'], 'Rot2_f': p['f_red_R']}
    conf_c = {k: v.detach() for k, v in conf.items()}
    pose = {'Tran': p['Pred_T'], 'Size': p['Pred_s']}
    gt_pose = {'Points': PC, 'R': gt["gt_R"], 'T': gt["gt_t"], 'Mean_shape': gt["mean_shape"]}
    ld_t = {
        'fsnet_loss': H.fs_net_loss()(names[0], {**axes, **conf, **pose, 'Recon': p['recon']},
                                      {'Rot1': g_green, 'Rot2': g_red, 'Recon': PC, 'Tran': gt["gt_t"], 'Size': gt["gt_s"]}, sym),
        'recon_loss': H.recon_6face_loss()(names[1], {**axes, **conf_c, **pose, 'F_n': p['face_normal'], 'F_d': p['face_dis'],
                                                      'F_c': p['face_f']}, {**gt_pose, 'Size': gt["gt_s"]}, sym, gt["obj_id"]),
        'geo_loss': H.geo_transform_loss()(names[2], {**axes, **conf_c, **pose}, gt_pose, sym),
        'prop_loss': H.prop_rot_loss()(names[3], {**axes, **conf_c, 'Recon': p['recon'], 'Tran': p['Pred_T'], 'Scale': p['Pred_s']},
                                       gt_pose, sym),
    }
    tf = tt = 0.0
    for grp, keys in LOSS_KEYS.items():
        for k in keys:
            a, b = float(ld_f[grp][k].detach().sum()), float(ld_t[grp][k].detach().sum())
            if np.isnan(b):
                assert np.isnan(a), (grp, k)
                continue
            assert abs(a - b) <= 2e-3 * max(abs(b), 1e-2), (grp, k, a, b)
            tf = tf + ld_f[grp][k].sum(); tt = tt + ld_t[grp][k].sum()
    tf.backward(); tt.backward()
    for k in NET:
        ga, gb = pred[k].grad, pred2[k].grad
        ok = torch.isfinite(gb)
        tol = 1e-2 if k in ("face_normal", "face_dis") else 5e-4
        assert float((ga - gb)[ok].abs().max()) <= tol * max(1.0, float(gb[ok].abs().max())), k


def test_fused_losses_nan_plane_poisons_box_terms(dev, ref, flags):
    """a NaN in any fitted plane turns the five box terms into NaN (recon_loss.py:632-639); the other terms stay finite"""
    gt, pred = case(ref, dev)
    with torch.no_grad():
        pred["face_dis"][0, :, 1] = float("nan")
    ld = fused(gt, pred)
    for k in ("recon_point_vote", "recon_point_r", "recon_point_t", "recon_point_s", "recon_point_self"):
        assert torch.isnan(ld["recon_loss"][k]).all(), k
    for grp, k in (("fsnet_loss", "Rot1"), ("fsnet_loss", "R_con"), ("geo_loss", "geo_point"), ("prop_loss", "Prop_pm")):
        assert torch.isfinite(ld[grp][k]).all(), (grp, k)
