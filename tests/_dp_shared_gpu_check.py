"""Worker of tests/test_gpu_dp_shared.py: run under torchrun with 2 ranks that SHARE one GPU (HSP_DIST_DEVICE=cuda:0,
HSP_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device).  The data-parallel step of bench.py through the REAL network:
FaceRecon, B = 16 clouds per rank, N = 1028, hipGraph replay (graph.GraphedStep(flat_grads=True)) + the gradient exchange in
both forms (parallel.graphed_step_with_exchange).  Each rank writes what it computed -- its ``feat``, its local gradients and
the exchanged flat buffer -- for the parent test, which repeats every rank's step in a single process ("rank-r
forward == single-process forward on that rank's clouds with that rank's randperm; all-reduced grad == mean of per-rank grads").

``local_step`` is also what the parent calls, so both sides run literally the same function."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

B, N = 16, 1028


def build(rank, device, split):
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.graph import GraphedStep
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(0)                                   # identical replicas
    net = FaceRecon().to(device).train()
    g = torch.Generator().manual_seed(10 + rank)           # this rank's shard of the global batch
    pc = torch.randn(B, N, 3, generator=g) * 0.05
    pc = (pc - pc.mean(dim=1, keepdim=True)).to(device)
    obj = torch.randint(0, 6, (B, 1), generator=g).float().to(device)
    dfeat = torch.randn(B, N, 1286, generator=g).to(device)
    gs = GraphedStep(net, pc, obj, dfeat, flat_grads=True, split=split)
    return net, gs


def named_grads(net, gs):
    """gradient per parameter NAME out of the step's flat buffer (the split form lays it out [late | early])"""
    names = {id(p): n for n, p in net.named_parameters()}
    return {names[id(p)]: v.detach().clone().cpu() for p, v in zip(gs.params, gs.grad_views())}


def local_step(rank, device, split):
    """one replay WITHOUT the exchange, with rank ``rank``'s data and Pool_layer draws"""
    net, gs = build(rank, device, split)
    torch.manual_seed(100 + rank)                          # the Pool_layer randperm draws of this replay
    if split:
        gs.run_first(); gs.run_second()
    else:
        gs.run()
    torch.cuda.synchronize()
    return net, gs, {"feat": gs.feat.detach().clone().cpu(), "grads": named_grads(net, gs)}


def main():
    from hs_pose_amd.parallel import describe, graphed_step_with_exchange, init_distributed
    out_dir = sys.argv[1]
    rank, world, device = init_distributed()
    assert world == 2 and device == torch.device(os.environ["HSP_DIST_DEVICE"]) and dist.get_backend() == os.environ["HSP_DIST_BACKEND"]
    assert describe()["world_size"] == 2
    for split in (False, True):
        net, gs, rec = local_step(rank, device, split)
        # the same draws again, now with the exchange.  What each collective is HANDED is snapshotted (a second replay of the same
        # step differs from the first by the fp32 atomics' arrival order, so "exchanged == mean of the operands" is checked on the
        # operands of THIS replay): the flat buffer before its all-reduce, or its [late | early] halves in the two-graph form
        from hs_pose_amd import parallel
        handed, real = [], parallel.all_reduce_

        def spy(t, *a, **kw):
            torch.cuda.synchronize()
            handed.append(t.detach().clone())
            return real(t, *a, **kw)
        parallel.all_reduce_ = spy
        try:
            torch.manual_seed(100 + rank)
            graphed_step_with_exchange(gs, world)
            torch.cuda.synchronize()
        finally:
            parallel.all_reduce_ = real
        assert len(handed) == (2 if split else 1) and sum(h.numel() for h in handed) == gs.flat_grad.numel()
        rec["exchanged"] = named_grads(net, gs)
        after = gs.flat_grad.clone()
        gs.flat_grad.copy_(torch.cat(handed))              # (views by name of what went INTO the exchange)
        rec["handed"] = named_grads(net, gs)
        gs.flat_grad.copy_(after)
        rec["feat_after_exchange_step"] = gs.feat.detach().clone().cpu()
        torch.save(rec, os.path.join(out_dir, f"rank{rank}_split{int(split)}.pt"))
        del gs, net
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DP_SHARED_GPU_OK", flush=True)


if __name__ == "__main__":
    main()
