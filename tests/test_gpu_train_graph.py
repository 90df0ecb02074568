"""GraphedTrainStep (the full engine/train.py:76-104 step as one hipGraph) against the eager step, at a small size
and at BASELINE configs[1] (B=16, N=1028).  Runs in a child process: the HIP runtime reads
DEBUG_CLR_GRAPH_PACKET_CAPTURE when it starts (hs_pose_amd/graph.py explains why the flag is needed)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(4, 256), (16, 1028)])
def test_graphed_train_step_matches_eager(B, N):
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_train_graph_check.py"), str(B), str(N)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_graphed_train_step_refuses_default_runtime(monkeypatch):
    import torch
    from hs_pose_amd.graph import GraphedTrainStep
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        GraphedTrainStep(None, None, {"PC": torch.zeros(1, 8, 3)})
