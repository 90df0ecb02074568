"""bench.py -- HS-layer stack forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of unit U1 over one per-GPU batch of synthetic clouds, all inputs
already resident in HBM: FaceRecon backbone (5 HS layers, 2 pools, nearest up-sample, concat; train-mode
BatchNorm, reference init under torch.manual_seed(0)) forward to feat (B,N,1286), then backward from a
given dfeat to every HS-stack parameter; for N>1 ranks the gradient mean over RCCL is inside the step.
Workload at N=1: BASELINE.json configs[1] shape, B=16 N=1028 fp32.  Weak scaling: 16 clouds per GPU.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant libhsp kernel of the step: algorithmic bytes per launch / live HIP-event
                  average duration over the timed region vs the 8 TB/s HBM peak;
  cpu_baseline -- (rank 0, N=1 only) the CPU oracle restatement (oracle/ref_cpu.py, kind "port") timed on
                  the host cores on a bounded sample of the same workload.  Reported, never a target.
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
MFMA_F32_PEAK_TFLOPS = 157.3   # same guide: v_mfma_f32_32x32x2_f32 dense peak (= the fp32 vector rate)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # same guide: dense bf16 MFMA peak
VALU_F32_PEAK_TFLOPS = 157.3   # same guide: peak FP32 (vector)
SHADER_CLOCK_GHZ = 2.4         # same guide: max clock (the issue floor is quoted at it: a kernel above 1.0 of its floor ran slower clocks)
# issue-bound calls -> useful fp32 flops per (point, neighbour, support column)
VALU_BOUND_CALLS = {"hsp_rf_conv_fwd": 8, "hsp_rf_surface_fwd": 7}
# HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/refresh_profiles.sh)
def _latest_traffic_json():
    """profiles/rNN/traffic.json of the latest round that has one (written by tools/refresh_profiles.sh BEFORE the bench line)"""
    base = os.path.join(ROOT, "profiles")
    rounds = sorted(d for d in (os.listdir(base) if os.path.isdir(base) else []) if os.path.isfile(os.path.join(base, d, "traffic.json")))
    return os.path.join(base, rounds[-1], "traffic.json") if rounds else os.path.join(base, "traffic.json")


TRAFFIC_JSON = _latest_traffic_json()
# the same round's graph_calls.json (tools/graph_call_us.py over the committed graph-replay trace): kernels' durations INSIDE the replay
GRAPH_CALLS_JSON = os.path.join(os.path.dirname(TRAFFIC_JSON), "graph_calls.json")


def u1_algorithmic(N, k=20, S=7):
    """(bytes, GEMM-shaped flops) per cloud of unit U1 forward+backward, SURVEY.md 8(d): every (N,k,S*C) and (N,N)
    intermediate counts as zero; each layer input read once, output written once, the (N,S*Cout) support tensor and the
    graph-conv feature written + read once, int32 indices written + read once, feat written once; backward = 2x forward.
    GEMM-shaped flops = fm GEMM + feature-space distance tiles + STE + conv2 per layer (1.57 GFLOP forward at N=1028).
    The byte model is normalised to SURVEY's 31.1 MB/cloud forward at N=1028 (it gives 29.6 MB there: the survey also
    counts the pools' / up-sampling's small reads)."""
    def model(n0):
        n1 = int(n0 / 4); n2 = int(n1 / 4)
        k1, k2 = min(k, n1 // 8), min(k, n2 // 8)
        by = n0 * (12 + 4 * 128 + 8 * 128 + 8 * k)                                       # conv_0
        fl = n0 * 2 * (3 * 128 + 256 * 128)
        for n, cin, cout, kk in ((n0, 128, 128, k), (n1, 128, 256, k1), (n1, 256, 256, k1), (n2, 256, 512, k2)):
            by += n * (4 * cin + 8 * S * cout + 4 * cout + 8 * cout + 16 * kk)
            fl += 2 * n * (cin * (S + 1) * cout + n * cin + cin * cout + 2 * cout * cout)
        by += 4 * n0 * 128 + n1 * (4 * 128 + 12) + 4 * n1 * 256 + n2 * (4 * 256 + 12)   # pools
        by += 4 * n0 * 1286                                                              # feat
        return by, fl
    by, fl = model(N)
    by1028, _ = model(1028)
    return 3 * by * (31.1e6 / by1028), 3 * fl


def make_inputs(B, N, device, seed=0):
    """SURVEY 8d synthetic inputs: PC = randn*0.05 + [0,0,0.8] (centred like PoseNet9D.py:25 does),
    obj_id = randint(0,6), dfeat = randn."""
    g = torch.Generator().manual_seed(seed)
    pc = torch.randn(B, N, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.8])
    obj = torch.randint(0, 6, (B, 1), generator=g).float()
    dfeat = torch.randn(B, N, 1286, generator=g)
    centred = pc - pc.mean(dim=1, keepdim=True)
    return centred.to(device), obj.to(device), dfeat.to(device)


def u3_case(B, N, device):
    """synthetic training batch of B clouds x N points with a pose / size ground truth (HSPose.forward's keyword set)"""
    g = torch.Generator().manual_seed(7)
    pc = torch.randn(B, N, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.8])
    obj = torch.randint(0, 6, (B,), generator=g).float()
    q, r = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))
    q = q * torch.sign(torch.diagonal(r, dim1=-2, dim2=-1)).unsqueeze(-2)
    gt_R = q * torch.sign(torch.linalg.det(q)).view(B, 1, 1)
    sym_table = torch.tensor([[1, 1, 0, 1], [1, 1, 0, 1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 1, 0, 0], [0, 1, 0, 0]], dtype=torch.float32)
    case = dict(PC=pc, obj_id=obj, gt_R=gt_R, gt_t=pc.mean(dim=1) + 0.01 * torch.randn(B, 3, generator=g),
                gt_s=0.02 * torch.randn(B, 3, generator=g), mean_shape=0.12 + 0.03 * torch.rand(B, 3, generator=g),
                sym=sym_table[obj.long()], aug_bb=torch.ones(B, 3), aug_rt_t=torch.zeros(B, 3),
                aug_rt_r=torch.eye(3).repeat(B, 1, 1), model_point=0.5 * torch.randn(B, 32, 3, generator=g),
                nocs_scale=torch.full((B,), 0.3))
    return {k: v.to(device) for k, v in case.items()}


def u3_full_step(B, N, device, steps=20, warmup=5):
    """unit U3 of SURVEY 8(d) = BASELINE configs[1] as worded ("full HSPose forward+backward"): HSPose.forward(do_loss=True)
    -- on-device augmentation, backbone, the three pose heads + reconstruction / face heads, the 19 loss terms -- backward,
    clip_grad_norm_(5), fused Ranger step, exactly the body of the reference's engine/train.py:72-110, on B synthetic clouds
    with a synthetic pose / size ground truth.  The device work of a step is ONE hipGraph replay (graph.GraphedTrainStep:
    augmentation, network, the five loss kernels, backward, squared gradient norm) + the fused optimizer launch; the host
    draws of the reference (jitter factors, the two Pool_layer permutations: CPU generator, reference order) go up through a
    pinned ring before each replay.  ``u3_network_graphed_ms_per_step`` is the round-1..5 form of the same step (network
    forward / backward as two graphs behind an autograd node, augmentation / losses / optimizer issued eagerly), kept as the
    cross-check.  Reported next to the headline, never instead of it."""
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.graph import GraphedTrainStep
    from hs_pose_amd.HSPose import HSPose
    from hs_pose_amd.train import TrainDriver

    def timed(step):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    FLAGS.reset(); FLAGS.train = 1
    # the host side of a step is a handful of small serial draws; ATen would spread each > 32 k-element CPU op over an OpenMP pool
    # sized for the box's 256 hardware threads inside a 16-CPU container (measured: 8 ms for one 49 k-element multiply)
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    torch.manual_seed(0)
    net = HSPose("PoseNet_only").to(device).train()
    drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
    case = u3_case(B, N, device)
    gs = GraphedTrainStep(net, drv.optimizer, case, scheduler=drv.scheduler, warmup=3)
    ms = timed(gs.run)
    out = {"u3_host_threads": 1, "u3_ms_per_step": round(ms, 3), "u3_clouds_per_s": round(B * 1e3 / ms, 1), "u3_steps": steps,
           "u3_unit": "HSPose.forward(do_loss=True) + backward + clip + Ranger: one hipGraph replay (augmentation, network, fused "
                      "loss kernels, backward, gradient norm) + the fused optimizer launch; host draws uploaded before each replay"}
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            gs.graph.replay()
        torch.cuda.synchronize()
        out["u3_graph_replay_only_ms"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)    # the device work alone
        del gs, drv, net
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        net = HSPose("PoseNet_only").to(device).train()
        drv = TrainDriver(net, total_iters=150 * 1500, check_nan=False)
        net.enable_graphed_posenet(case["PC"], case["obj_id"])

        def eager_step():
            _, ld = net(do_loss=True, **case)
            drv.step(net.total_loss(ld))                    # (= the sum over the four sub-dictionaries, engine/train.py:84-90)
        out["u3_network_graphed_ms_per_step"] = round(timed(eager_step), 3)
    except Exception as exc:                                # the cross-check never costs the figure above
        out["u3_crosscheck_error"] = f"{type(exc).__name__}: {exc}"[:160]
    FLAGS.reset()
    torch.set_num_threads(host_threads)
    return out


def cpu_baseline(n_points, sample_clouds):
    """time the CPU oracle (restatement of the reference, oracle/ref_cpu.py) on the host cores:
    same unit (HS stack fwd + bwd, train-mode BN, reference-shaped params), bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_cpu as oc                                      # checker / baseline only
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(0)
    sd = FaceRecon().state_dict()
    p = {k: v.detach().clone() for k, v in sd.items()}
    for k in p:
        if p[k].is_floating_point() and "running" not in k:
            p[k].requires_grad_(True)
    centred, obj, dfeat = make_inputs(sample_clouds, n_points, torch.device("cpu"))
    # MKL/oneDNN on these small per-cloud matrices stops scaling (and then degrades) beyond a few dozen
    # threads; use at most 32 and report the number actually used
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    torch.manual_seed(1)
    reps, dt = 0, 0.0
    while reps < 2 or (dt < 10.0 and reps < 6):               # >= 2 steps, ~10-30 s of CPU work in all
        for v in p.values():
            v.grad = None
        t0 = time.perf_counter()
        feat = oc.face_recon(p, centred, obj, oc.draw_pool_indices(n_points), train_heads=False, bn_training=True)["feat"]
        feat.backward(dfeat)
        dt += time.perf_counter() - t0
        reps += 1
    return {"value": round(reps * sample_clouds / dt, 4), "unit": "point-clouds/sec", "cores": cores, "kind": "port",
            "sample": f"{reps} fwd+bwd steps of the HS stack on {sample_clouds} clouds x N={n_points} fp32 "
                      f"(oracle/ref_cpu.py, torch CPU, {cores} threads), {dt:.1f} s"}


def side_run(extra_args, env_extra=None, timeout=600):
    """this script again, in a child process, for a secondary figure of the same run (another dtype / GEMM mode: module
    state, the captured graph and the tuned-library tables of the parent stay untouched); returns the child's JSON line"""
    import subprocess
    env = dict(os.environ)
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-u3", "--no-cpu-baseline", "--no-side"]
                         + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"side run {extra_args} failed (rc {out.returncode}): {out.stderr[-300:]}")
    return json.loads(lines[-1])


def relaunch_multi_gpu(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run and hand over;
    refuses when fewer than N devices are visible (a silent 1-rank run would print n_gpus: 1 for an N-GPU request)"""
    have = torch.cuda.device_count()
    if have < n:
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) visible")
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def dry_run(args, rank, world, device, result_out):
    """the launch / timing / reporting skeleton of main() with a stub step (tests/test_bench_contract.py runs it with two gloo
    ranks on the CPU): WORLD_SIZE == --gpus, warm-up, barrier + K timed steps + barrier, MAX over ranks, one JSON line from
    rank 0 with the weak-scaled global batch.  The stub step is one all-reduce of a gradient-sized buffer (3.1 M fp32: the HS
    stack's parameters); its rate means nothing and the line says so."""
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU (or plain `python bench.py --gpus N`)"
    B, N = args.batch, args.points
    grad = torch.ones(3_100_288, dtype=torch.float32, device=device)

    def step():
        if dist.is_initialized():
            dist.all_reduce(grad)
            grad.div_(world)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    if rank == 0:
        assert abs(grad[0].item() - 1.0) < 1e-6               # the mean of identical replicas is the replica
        line = {"metric": f"point-clouds/sec (N={N}) HS-layer fwd+bwd", "value": round(world * B * args.steps / dt, 2),
                "unit": "point-clouds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic", "dry_run": True,
                "config": {"workload": "DRY RUN (no kernels): stub step = one gradient-sized all-reduce; launcher / contract check only",
                           "global_batch": world * B, "points": N, "parallelism": f"dp{world}",
                           "backend": dist.get_backend() if dist.is_initialized() else "none",
                           "process_group": {"backend": dist.get_backend() if dist.is_initialized() else None,
                                             "world_size": dist.get_world_size() if dist.is_initialized() else 1}},
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), file=result_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def roofline_of(kname, kkey, kd, bf16, traffic, graph_calls=None):
    """the roofline object of one C-ABI call: algorithmic flops (GEMM-shaped calls) or bytes per launch / its HIP-event average.
    ``graph_calls``: the committed in-graph durations (tools/graph_call_us.py) -- when the call is in it, ``in_graph`` prices the same
    algorithmic work on the kernels' durations inside the replayed graph (no gaps between a call's kernels, no eager launch path)."""
    from hs_pose_amd import ops
    if kd.get("aflops", 0) > 0:           # GEMM-shaped kernel (feature-space distance tiles / weight gradient): MFMA roofline
        peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
        achieved = kd["aflops"] / (kd["avg_us"] * 1e-6) / 1e12
        roof = {"bound": "mfma", "kernel": f"{kname}[{kkey}]", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 5), "avg_us": round(kd["avg_us"], 2), "algorithmic_flops_per_launch": kd["aflops"],
                "algorithmic_bytes_per_launch": kd["abytes"], "traffic": None}
    else:
        achieved = kd["abytes"] / (kd["avg_us"] * 1e-6) / 1e9
        roof = {"bound": "hbm", "kernel": f"{kname}[{kkey}]", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "avg_us": round(kd["avg_us"], 2),
                "algorithmic_bytes_per_launch": kd["abytes"], "traffic": None}
    roof["traffic"] = traffic.get(f"{kname}[{kkey}]", {}).get("hbm_bytes_per_launch")
    useful = 0.0
    m = re.match(r"B(\d+)N(\d+)k(\d+)S(\d+)C(\d+)$", kkey) if kname in VALU_BOUND_CALLS else None
    if m:
        # the receptive-field forward pair sits at VALUBusy 95-100 % (profiles/r05/k_pmc_mfma_util.txt): HBM is the wrong roof for it.
        # Priced on vector issue: useful fp32 flops per (point, neighbour, column) -- theta = 3-term chain (5) + relu (1) + max (1),
        # + the product with the support value (1) in the HS layers -- against the 157.3 TFLOP/s vector peak, AND the kernel's own
        # instruction floor: VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU, committed) x 4 clocks / 1024 SIMDs / clock
        Bq, Nq, kq, Sq, Cq = (int(v) for v in m.groups())
        useful = float(Bq) * Nq * kq * Sq * Cq * VALU_BOUND_CALLS[kname]
        achieved = useful / (kd["avg_us"] * 1e-6) / 1e12
        roof = {"bound": "valu", "kernel": roof["kernel"], "achieved": round(achieved, 3), "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / VALU_F32_PEAK_TFLOPS, 5), "avg_us": roof["avg_us"], "useful_flops_per_launch": useful,
                "algorithmic_bytes_per_launch": kd["abytes"], "hbm_frac": roof["frac"], "traffic": roof["traffic"]}
        insts = traffic.get(f"{kname}[{kkey}]", {}).get("valu_wave_insts_per_launch")
        if insts:
            floor_us = insts * 4.0 / 1024.0 / (SHADER_CLOCK_GHZ * 1e3)
            roof["valu_wave_insts_per_launch"] = insts
            roof["valu_issue_floor_us"] = round(floor_us, 2)
            roof["issue_floor_frac"] = round(floor_us / kd["avg_us"], 4)   # 1.0 = the kernel runs at its own instruction count's floor
    gc = (graph_calls or {}).get(f"{kname}[{kkey}]")
    if gc:
        us = gc["in_graph_us"]
        work = kd["aflops"] / 1e12 if roof["bound"] == "mfma" else useful / 1e12 if roof["bound"] == "valu" else kd["abytes"] / 1e9
        roof["in_graph"] = {"avg_us": us, "achieved": round(work / (us * 1e-6), 3), "frac": round(work / (us * 1e-6) / roof["peak"], 5),
                            "kernels": [k_["kernel"].split("(")[0] for k_ in gc["kernels"]]}
    sb = ops.design_stream_bytes.get((kname, kkey))
    if sb:                                 # the bytes the kernel streams by design (uint16 winning-row slots, winners' support values)
        roof["kernel_stream_bytes_per_launch"] = sb
        roof["kernel_stream_frac"] = round(sb / (kd["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=1028)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="feature storage of the HS stack: f32 = BASELINE configs[1] (the headline); bf16 = configs[3] "
                         "(run it as --dtype bf16 --points 4096 --batch 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-u3", action="store_true", help="skip the full-training-step (unit U3) figure appended to config")
    ap.add_argument("--cpu-sample", type=int, default=16, help="clouds in the CPU-baseline sample (one per-GPU batch)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-gemm-tuning", action="store_true", help="leave hipBLASLt/rocBLAS on their default heuristics")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / contract check without a GPU: the ranks rendezvous (gloo on CPU), run K stub steps with one "
                         "gradient-sized all-reduce each, and rank 0 prints the line with \"dry_run\": true -- no throughput claim")
    ap.add_argument("--no-side", action="store_true",
                    help="skip the secondary figures appended to config (bf16 dense clouds, the no-BLAS-library step)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_multi_gpu(args.gpus)                   # does not return

    # stdout carries exactly ONE line, the JSON result: libraries that write to fd 1 themselves (RCCL prints a version
    # banner there, flushed at exit) are sent to stderr
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from hs_pose_amd import ops
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.FaceRecon import FaceRecon
    from hs_pose_amd.parallel import GradReducer, all_reduce_, choose_exchange_form, graphed_step_with_exchange, init_distributed
    from hs_pose_amd.parallel import describe as parallel_describe

    rank, world, device = init_distributed()
    if args.dry_run:
        return dry_run(args, rank, world, device, result_out)
    assert device.type == "cuda", "bench.py measures the HIP path; it needs a GPU"
    if os.environ.get("HSP_GEMM") == "library":         # the comparison figure (a child process of the default run): BLAS-library
        from tools import library_gemm                  # composites through tools/library_gemm.py, TunableOp solution selection
        library_gemm.enable(tune=not args.no_gemm_tuning)
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU (or plain `python bench.py --gpus N`)"
    B, N = args.batch, args.points

    FLAGS.reset(); FLAGS.train = 0                      # U1: backbone only (feat), no train-only heads
    torch.manual_seed(0)
    net = FaceRecon().to(device).train()
    bf16 = args.dtype == "bf16"
    if bf16:
        net.set_feature_dtype(torch.bfloat16)           # bf16 feature rows / fm / gradients, bf16 MFMA products (ops_bf16.py)
    params = [p for p in net.parameters()]
    reducer = None                                     # eager fallback only (hooks must not exist during capture)
    centred, obj, dfeat = make_inputs(B, N, device, seed=rank)
    if bf16:
        dfeat = dfeat.bfloat16()
    torch.manual_seed(1 + rank)                         # Pool_layer randperm stream (per rank)

    def eager_step():
        for p in params:
            p.grad = None
        _, _, feat = net(centred, obj)
        feat.backward(dfeat)
        if reducer is not None:
            reducer.finish()

    graphed = None
    exchange_info = {"reason": "eager step (--no-graph or no capture)"}
    use_dist = dist.is_initialized()
    if not args.no_graph:
        # hipGraph replay of zero_grad+fwd+bwd (hs_pose_amd/graph.py); the Pool_layer randperm draws stay on
        # the host, before each replay.  Data parallel: the captured step also packs all gradients into one
        # flat buffer, which is mean-all-reduced with a single RCCL collective after every replay.
        from hs_pose_amd.graph import GraphedStep
        # Two forms: ONE graph + one all-reduce after it, or the step captured as two graphs cut below the coarse levels, so that
        # the all-reduce of their gradients (78 % of the bytes) runs on RCCL's stream under the N=1028 layers' backward (the
        # second graph launch + the asynchronous work cost ~0.15 ms per step on a 1-rank group).
        # Data parallel (world > 1): BOTH forms are captured and a start-up probe (3 exchanged replays each way, MAX over the
        # ranks, rank 0's decision broadcast) picks one -- the overlapped two-graph form unless the single all-reduce is more
        # than 3 % faster; HSP_SPLIT_GRAPH=1 / 0 forces a form.  config.grad_exchange records what ran and why.
        forced = {"1": "split", "0": "single"}.get(os.environ.get("HSP_SPLIT_GRAPH", ""))
        probe_anyway = os.environ.get("HSP_SPLIT_GRAPH") == "probe"      # test hook: run the probe on a forced 1-rank group too
        if use_dist and world == 1 and forced is None and not probe_anyway:
            forced = "single"                              # a forced 1-rank group (test hook): nothing to overlap
        forms = {}
        for name in (["split", "single"] if use_dist else ["single"]):
            if forced is not None and name != forced:
                continue
            try:
                forms[name] = GraphedStep(net, centred, obj, dfeat, flat_grads=use_dist, split=(name == "split"))
            except Exception as exc:                       # capture unsupported -> other form / eager, say so
                print(f"[bench] hipGraph capture (split={name == 'split'}) failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        if use_dist and not forms and forced is not None:  # the forced form did not capture: try the other before going eager
            other = "single" if forced == "split" else "split"
            try:
                forms[other] = GraphedStep(net, centred, obj, dfeat, flat_grads=True, split=(other == "split"))
            except Exception as exc:
                print(f"[bench] hipGraph capture (split={other == 'split'}) failed ({type(exc).__name__}: {exc})", file=sys.stderr)
        if use_dist:
            chosen, exchange_info = choose_exchange_form(forms, world, sync=torch.cuda.synchronize, forced=forced,
                                                         probe_single_rank=probe_anyway)
            graphed = forms.get(chosen)
        else:
            graphed, exchange_info = forms.get("single"), {"reason": "one rank, no process group"}
        forms = None
    if graphed is None and use_dist and world > 1:
        reducer = GradReducer(params)                      # bucketed all-reduce launched from autograd hooks

    def graphed_step():
        if use_dist:
            graphed_step_with_exchange(graphed, world)         # hs_pose_amd/parallel.py
        else:
            graphed.run()

    step = graphed_step if graphed is not None else eager_step

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    timer = ops.KernelTimer() if rank == 0 else None
    if graphed is None:
        ops.set_timer(timer)                            # eager: per-kernel HIP events inside the timed region
    # per-step HIP events on the stream the step is issued on (hipEvent timing, median of >= 50 steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    extra = max(0, 50 - args.steps)                     # the median is always over >= 50 steps; `value` stays the K-step wall clock
    if extra:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(extra + 1)]
        evs[0].record()
        for i in range(extra):
            step()
            evs[i + 1].record()
        fence()
        step_ms += [evs[i].elapsed_time(evs[i + 1]) for i in range(extra)]
    step_ms.sort()
    median_ms = step_ms[len(step_ms) // 2]
    if graphed is not None and rank == 0:
        # HIP events cannot be recorded inside a graph replay: the per-kernel durations of the roofline
        # line come from the same K steps issued eagerly right after the timed region (same kernels, same
        # buffers); profiles/ holds the rocprofv3 trace of the graph replays themselves.
        ops.set_timer(timer)
        for _ in range(args.steps):
            eager_step()
        torch.cuda.synchronize()
        ops.set_timer(None)

    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist.is_initialized():
        all_reduce_(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    if rank == 0:
        summ = timer.summary()
        graph_calls = {}
        if graphed is not None and not bf16 and B == 16 and N == 1028:
            try:                           # the committed graph-replay trace of this configuration, per C-ABI call
                with open(GRAPH_CALLS_JSON) as f:
                    graph_calls = json.load(f)["calls"]
            except Exception:
                graph_calls = {}
        # dominant call = the single C-ABI launch with the longest duration (per launch, not per shape key: two layers that share a
        # shape are two launches) -- by its kernels' durations inside the replayed graph where the committed trace has the call
        # (the timed region IS the replay; HIP events around a call of the eager re-issue also count the gaps between its
        # kernels), else by the eager HIP-event average
        def _dur(item):
            (n_, k_), d_ = item
            gc = graph_calls.get(f"{n_}[{k_}]")
            return gc["in_graph_us"] if gc else d_["avg_us"]
        (kname, kkey), kd = max(summ.items(), key=_dur)
        hsp_ms = sum(d["total_ms"] for d in summ.values()) / args.steps
        traffic = {}
        try:                               # HBM bytes per launch measured with rocprofv3 PMC passes (committed with the profile)
            with open(TRAFFIC_JSON) as f:
                traffic = json.load(f)
        except Exception:
            pass
        roof = roofline_of(kname, kkey, kd, bf16, traffic, graph_calls)
        roof["dominant_by"] = ("kernel durations inside the replayed graph (" + os.path.relpath(GRAPH_CALLS_JSON, ROOT) + ")"
                               if graph_calls.get(f"{kname}[{kkey}]") else "HIP events around the call")
        roof["traffic_source"] = (os.path.relpath(TRAFFIC_JSON, ROOT) + " (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE passes of this command; "
                                  "committed with the round's profile, not re-measured by this run)")
        # the same object for the five longest calls of the step (the "dominant" one above can change between runs when two
        # calls are within noise of each other: hsp_knn_f32[C128] and hsp_rf_conv_fwd[N1028] both sit near 85-100 us)
        top = sorted(summ.items(), key=lambda kv: -_dur(kv))[:5]
        roof_all = [roofline_of(n_, k_, d_, bf16, traffic, graph_calls) for (n_, k_), d_ in top]
        roof["avg_us_source"] = ("HIP events around the call, in the timed region" if graphed is None else
                                 "HIP events around the call, the same K steps re-issued eagerly right after the timed "
                                 "graph replays (events cannot be recorded inside a replay)")
        ubytes, uflops = u1_algorithmic(N)
        if bf16:
            ubytes *= 0.5                                      # every feature tensor of the byte model is stored in 2 bytes
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if bf16 else MFMA_F32_PEAK_TFLOPS
        step_s = median_ms * 1e-3
        step_roof = {"algorithmic_bytes_per_cloud": round(ubytes), "gemm_flops_per_cloud": round(uflops),
                     "step_hbm_frac": round(B * ubytes / step_s / 1e9 / HBM_PEAK_GBS, 5),
                     "step_mfma_frac": round(B * uflops / step_s / 1e12 / mfma_peak, 5),
                     "mfma_peak_tflops": mfma_peak}
        # the MEASURED step traffic ("report both algorithmic and measured fractions and say which one the 40 % claim
        # uses"): sum over every dispatch of the committed PMC passes of 2 * FETCH_SIZE + WRITE_SIZE, per cloud, on THIS run's step time
        meas = traffic.get("__step_bf16_b64_n4096__" if bf16 else "__step_f32_b16_n1028__")
        if meas and meas.get("clouds_per_step") == B and (N == 4096 if bf16 else N == 1028):
            mb = meas["measured_hbm_bytes_per_cloud"]
            step_roof.update({"measured_bytes_per_cloud": mb, "measured_hbm_frac": round(B * mb / step_s / 1e9 / HBM_PEAK_GBS, 5),
                              "measured_over_algorithmic": round(mb / ubytes, 3),
                              "measured_source": os.path.relpath(TRAFFIC_JSON, ROOT) + " (rocprofv3 --pmc passes of the eager step, "
                                                 + str(meas["profiled_steps"]) + " steps; committed, not re-measured by this run)"})
        step_roof["north_star_40pct_reads"] = ("BASELINE.json's '>= 40 % HBM roofline' is read against measured_hbm_frac (bytes the memory "
                                               "system moved); step_hbm_frac prices only the algorithmic 93.3 MB/cloud of SURVEY 8(d), "
                                               "which reaches 40 % only at ~34 k clouds/s per GPU")
        if args.breakdown:
            for (n_, k_), d in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
                print(f"{n_:22s} {k_:28s} calls/step {d['calls'] / args.steps:4.1f}  avg {d['avg_us']:9.1f} us  "
                      f"{d['total_ms'] / args.steps:8.3f} ms/step  alg {d['abytes'] / 1e6:8.2f} MB "
                      f"-> {d['abytes'] / (d['avg_us'] * 1e-6) / 1e9:8.1f} GB/s", file=sys.stderr)
            print(f"libhsp kernels {hsp_ms:.3f} ms/step of {1e3 * dt / args.steps:.3f} ms/step", file=sys.stderr)
        line = {
            "metric": f"point-clouds/sec (N={N}) HS-layer fwd+bwd",     # BASELINE.json's metric at the default N=1028
            "value": round(world * B * args.steps / dt, 2),
            "unit": "point-clouds/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "ms_per_step_median": round(median_ms, 4),          # HIP-event median over max(K, 50) steps
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"HS stack (FaceRecon backbone -> feat) fwd+bwd, B={B}/GPU N={N} "
                                   + ("bf16 feature storage + bf16 MFMA products, fp32 geometry / accumulation / parameters, "
                                      "train-mode BN, random-init weights (BASELINE configs[3] shape when B=64 N=4096)" if bf16 else
                                      "fp32, train-mode BN, random-init weights (BASELINE configs[1] shape)"),
                       "global_batch": world * B, "points": N, "parallelism": f"dp{world}", "process_group": parallel_describe(), "hipgraph": graphed is not None, "split_graph": bool(graphed is not None and graphed.split), "grad_exchange": ("none" if not use_dist else "2 all-reduces, first overlapped with backward" if graphed is not None and graphed.split else "1 all-reduce after backward" if graphed is not None else "bucketed, hook-driven"), "grad_exchange_choice": exchange_info,
                       "libhsp_ms_per_step": round(hsp_ms, 4),
                       **({"bf16_tolerance": "feat vs the fp32 path: <= 6e-2 of scale / 8e-2 rms (tests/test_gpu_bf16.py); per layer vs the "
                                             "CPU oracle 1e-2; free-running neighbour-set agreement floors 0.05-0.2: a stress configuration, "
                                             "not a parity claim"} if bf16 else {}),
                       # dense per-point products: hand-written csrc/gemm_rows.hip vs the BLAS library, per composite shape
                       "dense_products": ("fp32 in / out / accumulation; products on the bf16 matrix cores from exact three-way bf16 "
                                          "splits of both operands, 6 of the 9 slice products (csrc/gemm_x3.hip: error vs fp64 within "
                                          "about 3x max / 2x rms of an fp32 library GEMM's, profiles/r04/x3_error.txt, "
                                          "tests/test_gpu_gemm_x3.py); feature-space distance tiles, K = 3 "
                                          "products and the eval-mode forward on the fp32 matrix cores" if not bf16 else
                                          "bf16 operands on the bf16 matrix cores, fp32 accumulation"),
                       "gemm": {"mode": getattr(ops, "gemm_mode", "own")}},
            "roofline": roof,
            "roofline_longest_calls": roof_all,
            "step_roofline": step_roof,
        }
        if world == 1 and not args.no_u3 and not bf16:
            try:                                        # BASELINE configs[1] as worded: the FULL training step, same run
                del graphed
                torch.cuda.empty_cache()
                line["config"].update(u3_full_step(B, N, device))
            except Exception as exc:                    # never lose the headline line to the extra figure
                line["config"]["u3_error"] = f"{type(exc).__name__}: {exc}"[:200]
        if world == 1 and not args.no_side and not bf16 and B == 16 and N == 1028:
            # secondary figures of the same run, each from a child process: BASELINE configs[3] (bf16 feature storage, dense
            # clouds) and the fp32 step with NO BLAS-library GEMM (HSP_GEMM=own: gemm_wave / gemm_rows / wgrad everywhere)
            try:
                torch.cuda.empty_cache()
                d = side_run(["--dtype", "bf16", "--points", "4096", "--batch", "64", "--steps", "10", "--warmup", "3"])
                line["config"].update({"bf16_b64_n4096_ms_per_step": d["ms_per_step"], "bf16_b64_n4096_clouds_per_s": d["value"],
                                       "bf16_b64_n4096_tolerance": d["config"].get("bf16_tolerance"),
                                       "bf16_b64_n4096_step_hbm_frac": d["step_roofline"]["step_hbm_frac"],
                                       "bf16_b64_n4096_roofline": d["roofline"]})
            except Exception as exc:
                line["config"]["bf16_b64_n4096_error"] = f"{type(exc).__name__}: {exc}"[:200]
            try:
                other = "library" if getattr(ops, "gemm_mode", "own") == "own" else "own"
                d = side_run(["--steps", "20", "--warmup", "5"], {"HSP_GEMM": other})
                line["config"].update({f"{other}_gemm_ms_per_step": d["ms_per_step"], f"{other}_gemm_clouds_per_s": d["value"]})
            except Exception as exc:
                line["config"]["other_gemm_error"] = f"{type(exc).__name__}: {exc}"[:200]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_sample)
        print(json.dumps(line), file=result_out, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
