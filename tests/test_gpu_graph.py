"""hipGraph replay of the HS-stack step (hs_pose_amd/graph.py::GraphedStep -- what bench.py times) against the same
step issued eagerly on a twin network: feat and every parameter gradient, for the single graph, the flat-gradient
form of the data-parallel bench, and the two-graph split that overlaps the gradient exchange."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, N, dev):
    g = torch.Generator().manual_seed(5)
    pc = torch.randn(B, N, 3, generator=g) * 0.05
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    dfeat = torch.randn(B, N, 1286, generator=g)
    return (pc - pc.mean(dim=1, keepdim=True)).to(dev), obj.to(dev), dfeat.to(dev)


def _net(dev):
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    FLAGS.reset()
    FLAGS.train = 0
    torch.manual_seed(0)
    return FaceRecon().to(dev).train()


@pytest.mark.parametrize("B,N", [(2, 256), (16, 1028)])
@pytest.mark.parametrize("mode", ["single", "flat", "split"])
def test_graphed_step_matches_eager(dev, B, N, mode):
    from hs_pose_amd import gcn3d
    from hs_pose_amd.graph import GraphedStep
    centred, obj, dfeat = _inputs(B, N, dev)
    net_g, net_e = _net(dev), _net(dev)
    torch.manual_seed(11)
    graphed = GraphedStep(net_g, centred, obj, dfeat, warmup=2, flat_grads=(mode == "flat"), split=(mode == "split"))
    for _ in range(3):                                  # replays overwrite, they must not accumulate
        feat_g = graphed.run()
    torch.cuda.synchronize()
    pool = [p.clone() for p in graphed.pool_idx]
    with gcn3d.pool_index_feed(pool):
        _, _, feat_e = net_e(centred, obj)
    feat_e.backward(dfeat)
    torch.cuda.synchronize()
    scale = feat_e.abs().max().item()
    assert (feat_g - feat_e).abs().max().item() <= 1e-5 * scale
    if mode == "single":
        got = {k: p.grad for k, p in net_g.named_parameters()}
    else:
        name_of = {id(p): k for k, p in net_g.named_parameters()}
        got = {name_of[id(p)]: v for p, v in zip(graphed.params, graphed.grad_views())}
        assert sum(v.numel() for v in got.values()) == graphed.flat_grad.numel()
    gmax = max(p.grad.abs().max().item() for p in net_e.parameters() if p.grad is not None)
    for k, p in net_e.named_parameters():
        if p.grad is None:
            assert got[k] is None or got[k].abs().max().item() == 0.0, k
            continue
        err = (got[k] - p.grad).abs().max().item()
        assert err <= 2e-5 * gmax, f"{k}: |graph - eager| {err:.3e}, max|grad| {gmax:.3e}"
    if mode == "split":                                 # the cut: coarse levels first, and they carry most of the bytes
        assert graphed.flat_late.numel() > graphed.flat_early.numel() > 0


@pytest.mark.parametrize("n", [1, 5])
def test_graphed_inference_matches_eager(dev, n):
    """evaluate.py's timed body (eval-mode forward + generate_RT) replayed from a graph == issued eagerly"""
    from hs_pose_amd import gcn3d
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.geom_utils import generate_RT
    from hs_pose_amd.graph import GraphedInference
    from hs_pose_amd.HSPose import HSPose
    FLAGS.reset()
    FLAGS.train = 0
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():                       # non-trivial running statistics
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
    net.eval()
    N = 1028
    PC = (torch.randn(n, N, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.8])).to(dev)
    obj = torch.randint(0, 6, (n,), generator=g).to(dev)
    mean_shape = (torch.rand(n, 3, generator=g) * 0.2 + 0.1).to(dev)
    sym = torch.zeros(n, 4, dtype=torch.int32)
    sym[::2, 0] = 1
    sym = sym.to(dev)
    torch.manual_seed(21)
    graphed = GraphedInference(net, PC, obj, mean_shape, sym)
    for _ in range(2):
        RT_g, s_g, out_g = graphed.run()
    torch.cuda.synchronize()
    with torch.no_grad(), gcn3d.pool_index_feed([p.clone() for p in graphed.pool_idx]):
        out = net(PC=PC, obj_id=obj, mean_shape=mean_shape, sym=sym)
        RT = generate_RT([out['p_green_R'], out['p_red_R']], [out['f_green_R'], out['f_red_R']], out['Pred_T'],
                         mode='vec', sym=sym)
    assert RT_g.shape == (n, 4, 4) and s_g.shape == (n, 3)
    for k in ('p_green_R', 'p_red_R', 'f_green_R', 'f_red_R', 'Pred_T', 'Pred_s'):
        assert (out_g[k] - out[k]).abs().max().item() <= 1e-5, k
    assert (RT_g - RT).abs().max().item() <= 1e-5
    assert (s_g - (out['Pred_s'] + mean_shape)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("B,N", [(4, 256), (16, 1028)])
def test_graphed_posenet_training_step_matches_eager(dev, B, N):
    """HSPose.forward(do_loss=True) with posenet replayed from its two hipGraphs (GraphedNetwork; losses eager) against
    the all-eager step on a twin network: all loss terms, every parameter gradient, the parameters after the Ranger step.
    Default runtime settings (graph packet capture on)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import ref_cpu as oc
    from hs_pose_amd import gcn3d
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.train import TrainDriver
    FLAGS.reset()
    FLAGS.train = 1
    FLAGS.aug_bb_pro = FLAGS.aug_rt_pro = FLAGS.aug_bc_pro = FLAGS.aug_pc_pro = -1.0
    keys = ("PC", "obj_id", "gt_R", "gt_t", "gt_s", "mean_shape", "sym", "aug_bb", "aug_rt_t", "aug_rt_r", "model_point",
            "nocs_scale")
    case = {k: v.to(dev) for k, v in oc.hspose_train_case(B, N, 7).items()}
    batch = {k: case[k] for k in keys}

    def make():
        torch.manual_seed(0)
        net = HSPose("PoseNet_only").to(dev).train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        return net, TrainDriver(net, total_iters=1000, check_nan=False)

    def total_of(ld):
        return sum(sum(d.values()) for d in ld.values())

    try:
        net_g, drv_g = make()
        torch.manual_seed(3)
        runner = net_g.enable_graphed_posenet(batch["PC"], batch["obj_id"])
        for _ in range(2):                                      # replays must not accumulate state across steps
            drv_g.optimizer.zero_grad()
            _, ld_g = net_g(do_loss=True, **batch)
            total_of(ld_g).backward()
        torch.cuda.synchronize()
        pool = [p.clone() for p in runner.pool_idx]
        grads_g = {k: p.grad.detach().clone() for k, p in net_g.named_parameters()}

        net_e, drv_e = make()
        with gcn3d.pool_index_feed(pool):
            _, ld_e = net_e(do_loss=True, **batch)
        drv_e.optimizer.zero_grad()
        total_of(ld_e).backward()
        torch.cuda.synchronize()
        for g in ld_e:
            for k in ld_e[g]:
                a, b = float(ld_e[g][k]), float(ld_g[g][k])
                assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), f"loss {g}.{k}: eager {a} graph {b}"
        gmax = max(p.grad.abs().max().item() for p in net_e.parameters() if p.grad is not None)
        for k, p in net_e.named_parameters():
            if p.grad is None:
                continue
            err = (p.grad - grads_g[k]).abs().max().item()
            assert err <= 1e-4 * gmax, f"grad {k}: |diff| {err:.3e} vs max|grad| {gmax:.3e}"
        for drv in (drv_g, drv_e):
            drv.optimizer.clip_grad_norm_(5)
            drv.optimizer.step()
        pe = dict(net_e.named_parameters())
        for k, p in net_g.named_parameters():
            assert (p - pe[k]).abs().max().item() <= 1e-5 * max(1.0, p.abs().max().item()), k
        # a batch of another shape (the last one of an epoch) takes the eager path of the same network
        small = {k: (v[:B - 1] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in batch.items()}
        _, ld_s = net_g(do_loss=True, **small)
        assert torch.isfinite(total_of(ld_s)).all()
        total_of(ld_s).backward()
    finally:
        FLAGS.reset()


def test_staging_ring_uploads_in_order(dev):
    """hs_pose_amd/staging.py: values pass through unchanged, slots are reused only after their upload has run, the
    host-side generator order is the caller's"""
    from hs_pose_amd import staging
    torch.manual_seed(5)
    want = [torch.rand(33, 3) for _ in range(11)]            # more uploads than ring slots
    torch.manual_seed(5)
    got = []
    sink = torch.zeros(33, 3, device=dev)
    for i in range(11):
        x = staging.upload(lambda buf: torch.rand(33, 3, out=buf), (33, 3), torch.float32, dev,
                           out=sink if i % 2 else None, ring=3)
        got.append(x.clone())
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
