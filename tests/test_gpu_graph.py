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
