"""Unit U3 of SURVEY 8d (not the headline metric): one FULL training step of the reference's engine/train.py loop --
HSPose.forward(do_loss=True) with on-device augmentation, the 19 losses, backward through heads + backbone, gradient
clipping and the fused Ranger step -- on B synthetic clouds of N points (B=16, N=1028 = BASELINE configs[1]).
Run on the GPU box:  python tools/bench_train_step.py [--steps 20] [--batch 16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch


def main():
    if os.environ.get("HSP_DUMP_AFTER"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["HSP_DUMP_AFTER"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly (CPU-bound) instead of replaying a hipGraph")
    ap.add_argument("--graph-net", action="store_true", help="eager step with posenet forward/backward replayed from two hipGraphs")
    args = ap.parse_args()
    from tools import gemm_tuning                # (library-GEMM tuning table: only matters under HSP_GEMM=library)
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.train import TrainDriver
    import ref_cpu as oc                         # only for the closed-form ground-truth generator of the fixtures
    gemm_tuning.enable()
    dev = torch.device("cuda:0")
    FLAGS.reset(); FLAGS.train = 1
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(dev).train()
    drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
    case = {k: v.to(dev) for k, v in oc.hspose_train_case(args.batch, args.points, 7).items()}

    def step():
        _, ld = net(PC=case["PC"], obj_id=case["obj_id"], gt_R=case["gt_R"], gt_t=case["gt_t"], gt_s=case["gt_s"],
                    mean_shape=case["mean_shape"], sym=case["sym"], aug_bb=case["aug_bb"], aug_rt_t=case["aug_rt_t"],
                    aug_rt_r=case["aug_rt_r"], model_point=case["model_point"], nocs_scale=case["nocs_scale"], do_loss=True)
        total = sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) \
            + sum(ld['prop_loss'].values())
        drv.step(total)
        return total

    graphed = None
    if args.graph_net:
        net.enable_graphed_posenet(case["PC"], case["obj_id"])
    if not args.no_graph and not args.graph_net:
        from hs_pose_amd.graph import GraphedTrainStep
        # (no eager step first: the network's first backward has to run on the capture stream -- see GraphedTrainStep)
        batch = {k: case[k] for k in ("PC", "obj_id", "gt_R", "gt_t", "gt_s", "mean_shape", "sym", "aug_bb", "aug_rt_t",
                                      "aug_rt_r", "model_point", "nocs_scale")}
        graphed = GraphedTrainStep(net, drv.optimizer, batch, scheduler=drv.scheduler)

        def step():                              # noqa: F811
            graphed.run()
            return graphed.total
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if os.environ.get("HSP_SEGMENTS"):
        def tick():
            torch.cuda.synchronize()
            return time.perf_counter()
        seg = {"network": 0.0, "losses": 0.0, "backward": 0.0, "clip+step": 0.0}
        for _ in range(args.steps):
            a = tick()
            out = net(PC=case["PC"], obj_id=case["obj_id"], gt_R=case["gt_R"], gt_t=case["gt_t"], gt_s=case["gt_s"],
                      mean_shape=case["mean_shape"], sym=case["sym"], aug_bb=case["aug_bb"], aug_rt_t=case["aug_rt_t"],
                      aug_rt_r=case["aug_rt_r"], model_point=case["model_point"], nocs_scale=case["nocs_scale"], do_loss=False)
            b = tick()
            _, ld = net(PC=case["PC"], obj_id=case["obj_id"], gt_R=case["gt_R"], gt_t=case["gt_t"], gt_s=case["gt_s"],
                        mean_shape=case["mean_shape"], sym=case["sym"], aug_bb=case["aug_bb"], aug_rt_t=case["aug_rt_t"],
                        aug_rt_r=case["aug_rt_r"], model_point=case["model_point"], nocs_scale=case["nocs_scale"], do_loss=True)
            total = sum(ld['fsnet_loss'].values()) + sum(ld['recon_loss'].values()) + sum(ld['geo_loss'].values()) \
                + sum(ld['prop_loss'].values())
            c = tick()
            total.backward()
            d = tick()
            drv.optimizer.clip_grad_norm_(5); drv.optimizer.step(); drv.optimizer.zero_grad()
            e = tick()
            seg["network"] += b - a; seg["losses"] += (c - b) - (b - a); seg["backward"] += d - c; seg["clip+step"] += e - d
        print({k: round(1e3 * v / args.steps, 2) for k, v in seg.items()}, "ms/step")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"unit": "U3 full training step (HSPose do_loss + backward + clip + Ranger), "
                              + ("hipGraph replay" if graphed is not None else "eager, posenet graphed" if args.graph_net else "eager"),
                      "batch": args.batch, "points": args.points, "ms_per_step": round(1e3 * dt / args.steps, 3),
                      "clouds_per_s": round(args.batch * args.steps / dt, 1), "last_total_loss": float(total.detach())}))

    gemm_tuning.save()                           # only with HSP_TUNABLEOP_OUT set


if __name__ == "__main__":
    main()
