"""CPU: the oracle's depth->cloud / pose-matrix restatements against the fixtures the reference wrote
(oracle/gen_golden_frontend.py), plus the host-side logic of the product mirrors that needs no GPU."""
import numpy as np
import torch

from conftest import golden


def _inputs(ref, g):
    return ref.frontend_inputs(3, 64, 80, 900, [float(r) for r in g["radii"]])


def test_pc_sample_oracle_matches_reference(ref):
    g = golden("frontend_pc_sample")
    mask, depth, camK, coor = _inputs(ref, g)
    counts = [ref.valid_pixels(mask[b], depth[b]).numel() for b in range(3)]
    assert counts == list(g["counts"])
    assert counts[1] < 1028 <= counts[2]                      # both the with- and without-replacement draws
    pc = ref.pc_sample(mask, depth, camK, coor, 1028, np.random.RandomState(7))
    assert np.array_equal(pc.numpy(), g["pc"])
    logits = torch.cat([ref.hash_tensor((3, 1, 64, 80), 950, 1.0),
                        2.0 * mask - 1.0 + ref.hash_tensor((3, 1, 64, 80), 951, 0.5)], dim=1)
    pc2 = ref.pc_sample(logits, depth, camK, coor, 1028, np.random.RandomState(8))
    assert np.array_equal(pc2.numpy(), g["pc_logits"])


def test_pc_sample_oracle_too_few_pixels(ref):
    g = golden("frontend_pc_sample")
    mask, depth, camK, coor = _inputs(ref, g)
    mask[1] = 0
    mask[1, 0, 3, 4] = 1.0
    depth[1, 0, 3, 4] = 700.0
    assert ref.pc_sample(mask, depth, camK, coor, 1028, np.random.RandomState(9)) is None


def test_depth_to_pcl_oracle_matches_reference(ref):
    g = golden("frontend_depth_to_pcl")
    gp = golden("frontend_pc_sample")
    mask, depth, camK, coor = _inputs(ref, gp)
    for b, seed in ((0, 11), (1, 12)):
        pcl = ref.depth_to_pcl(depth[b].numpy(), g["K"], coor[b].numpy(), mask[b].numpy()) / 1000.0
        assert pcl.dtype == np.float32
        got = ref.sample_points(pcl, 1028, np.random.RandomState(seed))
        assert np.array_equal(got, g[f"pcl{b}"])


def test_sample_point_ids_follow_loader_rule(ref):
    from hs_pose_amd.pc_sample import sample_point_ids
    pcl = np.arange(407 * 3, dtype=np.float32).reshape(407, 3)
    ids = sample_point_ids(407, 1028)
    assert np.array_equal(pcl[ids], ref.sample_points(pcl, 1028, None))          # tiling needs no RNG
    big = np.arange(2586 * 3, dtype=np.float32).reshape(2586, 3)
    np.random.seed(11)
    ids = sample_point_ids(2586, 1028)
    assert np.array_equal(big[ids], ref.sample_points(big, 1028, np.random.RandomState(11)))
    assert np.array_equal(sample_point_ids(1028, 1028), np.arange(1028))


def test_generate_rt_oracle_matches_reference(ref):
    g = golden("frontend_generate_rt")
    pg, pr, fg, fr, T, sym = ref.generate_rt_inputs()
    rt = ref.generate_rt(pg, pr, fg, fr, T, sym)
    assert np.abs(rt.numpy() - g["rt"]).max() < 1e-6
    R = rt[:, :3, :3]
    assert (R.transpose(1, 2) @ R - torch.eye(3)).abs().max() < 1e-5            # a rotation ...
    assert torch.allclose(torch.linalg.det(R), torch.ones(16), atol=1e-5)       # ... and a proper one
    assert torch.equal(rt[:, 3], torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(16, 4))
    assert torch.equal(rt[:, :3, 3], T)
    # symmetric objects ignore the red axis entirely: y column == the green axis
    s = sym[:, 0] == 1
    assert torch.allclose(R[s][:, :, 1], pg[s], atol=1e-5)


def test_hspose_surface_matches_reference_signature():
    import inspect
    from hs_pose_amd.HSPose import HSPose
    sig = inspect.signature(HSPose.forward)
    assert list(sig.parameters) == ["self", "PC", "depth", "obj_id", "camK", "gt_R", "gt_t", "gt_s", "mean_shape",
                                    "gt_2D", "sym", "aug_bb", "aug_rt_t", "aug_rt_r", "def_mask", "model_point",
                                    "nocs_scale", "do_loss"]
    assert sig.parameters["do_loss"].default is False
    assert list(inspect.signature(HSPose.build_params).parameters) == ["self", "training_stage_freeze"]


def test_hspose_state_dict_keys(state_keys, flags):
    from hs_pose_amd.HSPose import HSPose
    for train, name in ((1, "train"), (0, "eval")):
        flags.train = train
        net = HSPose("PoseNet_only")
        assert sorted(net.state_dict().keys()) == sorted("posenet." + k for k in state_keys[name])
    flags.train = 1
    groups = HSPose("PoseNet_only").build_params(training_stage_freeze=[])
    assert len(groups) == 1 and groups[0]["lr"] == 1e-4 and len(list(groups[0]["params"])) > 0
