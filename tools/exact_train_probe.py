"""DESIGN 2.2 / round-5 review, weak point 1: what does the reference-order arithmetic (ops.exact_scope) buy under TRAIN-mode BatchNorm?
Free-running PoseNet9D on the fixture stack_refinit_trainbn_1028 with and without HSP_EXACT_TRAIN=1: rows per HS layer whose
ordered feature-space neighbour list equals the reference's, max error of the six pose / size outputs, forward time."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
    import time
    import numpy as np, torch
    import ref_cpu as ref
    from conftest import golden
    import test_gpu_stack as T
    from hs_pose_amd.config import FLAGS
    from hs_pose_amd.PoseNet9D import PoseNet9D
    dev = torch.device("cuda:0")
    g = golden("stack_refinit_trainbn_1028")
    _, B, N, seed, bn_training, wseed = (int(v) for v in g["meta"])
    FLAGS.reset(); FLAGS.train = 0
    torch.manual_seed(wseed)
    net = PoseNet9D().to(dev)
    net.train(True)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts, obj = T._inputs(ref, B, N, seed, dev)

    class MP:                                   # the two monkeypatch calls ForcedFeatKnn makes
        def setattr(self, o, n, v): setattr(o, n, v)
    watch = T.ForcedFeatKnn(MP(), g, dev, force=False)
    torch.manual_seed(1)
    with torch.no_grad():
        outs = dict(zip(T.OUT_NAMES, net(pts, obj)))
    errs = {n_: T._maxerr(outs[n_], g["out." + n_]) for n_ in T.OUT_NAMES[4:]}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(10):
            torch.manual_seed(1); net(pts, obj)
    torch.cuda.synchronize()
    print(json.dumps({"exact_train": os.environ.get("HSP_EXACT_TRAIN", "0"), "ordered_lists": [round(a, 4) for a in watch.agree],
                      "sets": [round(a, 4) for a in watch.agree_set], "max_err": {k: float(f"{v:.2e}") for k, v in errs.items()},
                      "forward_ms_eager": round(1e2 * (time.perf_counter() - t0), 3)}))
else:
    for flag in ("0", "1"):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, HSP_EXACT_TRAIN=flag),
                             capture_output=True, text=True)
        print((out.stdout.strip().splitlines() or [out.stderr[-800:]])[-1])
