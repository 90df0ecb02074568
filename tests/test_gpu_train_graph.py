"""GraphedTrainStep (the full engine/train.py:76-104 step as one hipGraph) against the eager step, at a small size
and at BASELINE configs[1] (B=16, N=1028), under the DEFAULT HIP runtime settings (round 1 needed
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; hs_pose_amd/graph.py::GraphedTrainStep explains what changed) and with that flag.
Runs in a child process: the HIP runtime reads the flag when it starts."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,packet_capture", [(4, 256, None), (16, 1028, None), (16, 1028, "0")])
def test_graphed_train_step_matches_eager(B, N, packet_capture):
    env = dict(os.environ)
    env.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
    if packet_capture is not None:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = packet_capture
    r = subprocess.run([sys.executable, os.path.join(HERE, "_train_graph_check.py"), str(B), str(N)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
