"""GPU parity (through the C-ABI): depth -> cloud front end, pose-matrix back end and the HSPose facade
against the fixtures written by the reference (tests/golden/frontend_*.npz) and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _inputs(ref):
    g = golden("frontend_pc_sample")
    return g, ref.frontend_inputs(3, 64, 80, 900, [float(r) for r in g["radii"]])


def test_pc_compact_is_row_major_boolean_indexing(dev, ref):
    from hs_pose_amd import ops
    g, (mask, depth, camK, coor) = _inputs(ref)
    pix, count = ops.pc_compact(mask.reshape(3, -1).to(dev), depth.reshape(3, -1).to(dev))
    assert list(count.cpu().numpy()) == list(g["counts"])
    for b in range(3):
        want = ref.valid_pixels(mask[b], depth[b]).numpy()
        assert np.array_equal(pix[b, :len(want)].cpu().numpy(), want)          # integer work: bit-exact


@pytest.mark.parametrize("HW", [1, 63, 1024, 1025, 4095, 4096, 4097, 65536, 256 * 256 + 7, 480 * 640])
def test_pc_compact_sizes(dev, ref, HW):
    from hs_pose_amd import ops
    m = (ref.hash_tensor((2, HW), 31, 0.5, 0.5) > 0.4).float()
    d = ref.hash_tensor((2, HW), 32, 1.0) * 900.0                               # about half non-positive
    pix, count = ops.pc_compact(m.to(dev), d.to(dev))
    for b in range(2):
        want = ref.valid_pixels(m[b], d[b]).numpy()
        assert int(count[b]) == len(want)
        assert np.array_equal(pix[b, :len(want)].cpu().numpy(), want)


def test_pc_compact_all_and_none(dev, ref):
    """every pixel valid (the ids fill the whole row) and no pixel valid (count 0), several chunks per image"""
    from hs_pose_amd import ops
    HW = 3 * 4096 + 100
    ones = torch.ones(2, HW)
    depth = torch.full((2, HW), 700.0)
    depth[1] = 0.0
    pix, count = ops.pc_compact(ones.to(dev), depth.to(dev))
    assert count.cpu().tolist() == [HW, 0]
    assert torch.equal(pix[0].cpu(), torch.arange(HW, dtype=torch.int32))


def test_pc_sample_matches_reference(dev, ref, flags):
    from hs_pose_amd.pc_sample import PC_sample
    g, (mask, depth, camK, coor) = _inputs(ref)
    flags.random_points = 1028
    np.random.seed(7)
    pc = PC_sample(mask.to(dev), depth.to(dev), camK.to(dev), coor.to(dev))
    assert pc.shape == (3, 1028, 3) and pc.dtype == torch.float32
    assert np.array_equal(pc.cpu().numpy(), g["pc"])                            # same fp32 op order: exact
    # the global numpy generator was consumed exactly like the reference: the next draw agrees
    rs = np.random.RandomState(7)
    for c in g["counts"]:
        rs.choice(int(c), 1028, replace=int(c) < 1028)
    assert np.random.randint(1 << 30) == rs.randint(1 << 30)
    logits = torch.cat([ref.hash_tensor((3, 1, 64, 80), 950, 1.0),
                        2.0 * mask - 1.0 + ref.hash_tensor((3, 1, 64, 80), 951, 0.5)], dim=1)
    np.random.seed(8)
    pc2 = PC_sample(logits.to(dev), depth.to(dev), camK.to(dev), coor.to(dev))
    assert np.array_equal(pc2.cpu().numpy(), g["pc_logits"])


def test_pc_sample_too_few_pixels_returns_none_pair(dev, ref, flags):
    from hs_pose_amd.pc_sample import PC_sample
    from hs_pose_amd.HSPose import HSPose
    g, (mask, depth, camK, coor) = _inputs(ref)
    mask[1] = 0
    mask[1, 0, 3, 4] = 1.0
    depth[1, 0, 3, 4] = 700.0
    np.random.seed(9)
    assert PC_sample(mask.to(dev), depth.to(dev), camK.to(dev), coor.to(dev)) == (None, None)   # pc_sample.py:59-60
    flags.train = 0
    net = HSPose("PoseNet_only").to(dev).eval()
    out = net(depth=depth.to(dev), camK=camK.to(dev), gt_2D=coor.to(dev), def_mask=mask.to(dev),
              obj_id=torch.zeros(3, 1, device=dev))
    assert out == ({}, None)                                                    # HSPose.py:47-48


def test_depth_to_pcl_matches_loader(dev, ref):
    from hs_pose_amd.pc_sample import depth_to_pcl
    g = golden("frontend_depth_to_pcl")
    _, (mask, depth, camK, coor) = _inputs(ref)
    for b, seed in ((0, 11), (1, 12)):
        np.random.seed(seed)
        pcl = depth_to_pcl(depth[b:b + 1].to(dev), g["K"], coor[b:b + 1].to(dev), mask[b:b + 1].to(dev), n_pts=1028)
        assert np.array_equal(pcl[0].cpu().numpy(), g[f"pcl{b}"])              # float64 arithmetic, one rounding
    few = mask.clone()
    few[0] = 0
    few[0, 0, :5, :5] = 1.0                                                     # < 50 valid pixels: item rejected
    assert depth_to_pcl(depth.to(dev), g["K"], coor.to(dev), few.to(dev)) is None


def test_generate_rt_matches_reference(dev, ref):
    from hs_pose_amd.geom_utils import generate_RT, to_R_matrices
    g = golden("frontend_generate_rt")
    pg, pr, fg, fr, T, sym = [t.to(dev) for t in ref.generate_rt_inputs()]
    rt = generate_RT([pg, pr], [fg, fr], T, mode="vec", sym=sym)
    assert rt.shape == (16, 4, 4)
    assert np.abs(rt.cpu().numpy() - g["rt"]).max() < 1e-5                      # sin/cos/acos: device libm
    R = to_R_matrices(fg, fr, pg, pr)
    nosym = (sym[:, 0] != 1).cpu()
    assert np.abs(R.cpu().numpy()[nosym] - g["rt"][nosym][:, :3, :3]).max() < 1e-5
    gt = generate_RT(torch.eye(3, device=dev).expand(16, 3, 3), None, T, mode="gt", sym=sym)
    assert torch.equal(gt[:, :3, :3].cpu(), torch.eye(3).expand(16, 3, 3)) and torch.equal(gt[:, :3, 3], T)


def test_hspose_eval_matches_posenet_fixture(dev, ref, flags, monkeypatch):
    """evaluation/evaluate.py:91-106: network(PC=, obj_id=, mean_shape=, sym=) -> generate_RT -> pred_s."""
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.geom_utils import generate_RT
    from test_gpu_stack import ForcedFeatKnn, _inputs as stack_inputs
    g = golden("stack_eval_256")
    train_flag, B, N, seed, bn_training = (int(v) for v in g["meta"])
    flags.train = train_flag
    net = HSPose("PoseNet_only")
    sd = net.posenet.state_dict()
    ref.fill_state_closed_form(sd)
    net = net.to(dev).eval()
    PC, obj = stack_inputs(ref, B, N, seed, dev)
    sym = torch.zeros(B, 4, device=dev)
    sym[0, 0] = 1.0
    ForcedFeatKnn(monkeypatch, g, dev)
    torch.manual_seed(1)
    out = net(PC=PC, obj_id=obj, mean_shape=torch.zeros(B, 3, device=dev), sym=sym)
    assert list(out) == ["mask", "sketch", "recon", "PC", "face_normal", "face_dis", "face_f", "p_green_R", "p_red_R",
                         "f_green_R", "f_red_R", "Pred_T", "Pred_s", "gt_R", "gt_t", "gt_s"]
    names = ("p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s")
    for k in names:
        assert np.abs(out[k].detach().cpu().numpy() - g["out." + k]).max() <= 1e-4, k
    assert out["recon"] is None and out["mask"] is None and torch.equal(out["PC"], PC)
    rt = generate_RT([out["p_green_R"], out["p_red_R"]], [out["f_green_R"], out["f_red_R"]], out["Pred_T"],
                     mode="vec", sym=sym)
    want = ref.generate_rt(*[torch.from_numpy(g["out." + k]) for k in names[:5]], sym.cpu())
    assert (rt.cpu() - want).abs().max() < 1e-3                                 # 1e-4 inputs through acos


def test_hspose_branches_the_reference_does_not_implement(dev, flags):
    """the depth entry exists only for 'PoseNet_only' (HSPose.py:49-50 raises otherwise); unknown stages have no loss set"""
    from hs_pose_amd.HSPose import HSPose
    flags.train = 0
    with pytest.raises(NotImplementedError):
        HSPose("FSNet_only").to(dev)(depth=torch.zeros(1, 1, 4, 4, device=dev))
    with pytest.raises(NotImplementedError):
        HSPose("Backbone_only")
