"""probe (run two of these side by side on one GPU: tools/run_pair.sh tools/replay_race_probe.py): replay ONE captured step many times with identical inputs / pool draws; report parameters whose gradient deviates from the
first replay by more than 1e-4 of scale (an ordering race shows up here; an allocation-dependent read would not)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _dp_shared_gpu_check as w
dev = torch.device("cuda:0")
tag = sys.argv[1] if len(sys.argv) > 1 else "p"
for split in (False, True):
    net, gs = w.build(0, dev, split)
    names = {id(p): n for n, p in net.named_parameters()}
    first, nbad = None, 0
    for it in range(150):
        torch.manual_seed(100)
        if split:
            gs.run_first(); gs.run_second()
        else:
            gs.run()
        torch.cuda.synchronize()
        cur = {names[id(p)]: v.detach().clone() for p, v in zip(gs.params, gs.grad_views())}
        if first is None:
            first = cur
            continue
        for n, g in cur.items():
            e = float((g - first[n]).abs().max()) / (float(first[n].abs().max()) + 1e-30)
            if e > 1e-4:
                nbad += 1
                print(f"[{tag}] split {split} replay {it}: {n} deviates {e:.4f}", flush=True)
    print(f"[{tag}] split {split}: {nbad} deviations in 149 replays", flush=True)
    del gs, net
