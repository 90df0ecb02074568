"""bench.py's launch contract WITHOUT a GPU: two ranks under torch.distributed.run (gloo on the CPU) run `bench.py --gpus 2
--dry-run` -- the rendezvous from RANK / WORLD_SIZE / MASTER_*, the WORLD_SIZE == --gpus check, the barrier-bracketed timed
region with the MAX over ranks, and exactly ONE JSON line on rank 0's stdout with the weak-scaled global batch.  (The step is a
stub there: the line carries "dry_run": true and no throughput claim.)  Ready for the first multi-GPU lease."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(n, extra=()):
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
           "--dry-run", *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)


def test_bench_two_ranks_dry_run_contract():
    out = _run(2)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # ONE line, from rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 32 and line["config"]["parallelism"] == "dp2" and line["config"]["backend"] == "gloo"
    assert line["dry_run"] is True and line["vs_baseline"] is None and line["higher_is_better"] is True
    assert line["metric"].startswith("point-clouds/sec (N=1028)") and line["unit"] == "point-clouds/sec" and line["value"] > 0
    assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"} <= set(line)


def test_bench_rejects_a_rank_count_that_differs_from_gpus():
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--dry-run"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in out.stderr
