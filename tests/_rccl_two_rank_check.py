"""Run under torchrun with >= 2 ranks on one node (tests/test_gpu_stack.py::test_rccl_two_ranks launches it when the box
has >= 2 GPUs): the data-parallel step of bench.py over RCCL -- hipGraph replay + gradient exchange, single-all-reduce and
two-graph overlapped forms -- must leave in ``flat_grad`` the MEAN of the ranks' local gradients ("all-reduced
grad == mean of per-rank grads"), with per-rank inputs, BatchNorm statistics and randperm draws."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.graph import GraphedStep
    from hs_pose_amd.parallel import graphed_step_with_exchange, init_distributed
    rank, world, device = init_distributed()
    assert world >= 2 and device.type == "cuda" and dist.get_backend() == "nccl"
    B, N = 4, 256
    for split in (False, True):
        FLAGS.reset(); FLAGS.train = 0
        torch.manual_seed(0)                                   # identical replicas
        net = FaceRecon().to(device).train()
        g = torch.Generator().manual_seed(10 + rank)           # this rank's shard of the global batch
        pc = torch.randn(B, N, 3, generator=g) * 0.05
        pc = (pc - pc.mean(dim=1, keepdim=True)).to(device)
        obj = torch.randint(0, 6, (B, 1), generator=g).float().to(device)
        dfeat = torch.randn(B, N, 1286, generator=g).to(device)
        gs = GraphedStep(net, pc, obj, dfeat, flat_grads=True, split=split)
        torch.manual_seed(100 + rank)                          # the Pool_layer draws of this replay
        if split:
            gs.run_first(); gs.run_second()
        else:
            gs.run()
        local = gs.flat_grad.clone()
        assert float(local.abs().sum()) > 0
        torch.manual_seed(100 + rank)                          # same draws again, now with the exchange
        graphed_step_with_exchange(gs, world)
        torch.cuda.synchronize()
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        assert not torch.equal(gathered[0], gathered[1]), "ranks saw identical data: the test would prove nothing"
        err = (gs.flat_grad - want).abs().max().item()
        scale = want.abs().max().item()
        assert err <= 1e-6 * scale, f"rank {rank} split={split}: flat_grad vs mean of local grads {err:.3e} (scale {scale:.3e})"
        if rank == 0:
            print(f"rccl world={world} split={split}: max |flat_grad - mean| = {err:.3e} of {scale:.3e}", flush=True)
        del gs
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_TWO_RANK_OK", flush=True)


if __name__ == "__main__":
    main()
