"""
oracle/gen_golden_frontend.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Golden vectors for the steps either side of the network at inference:
  * PC_sample               network/point_sample/pc_sample.py:8-77
  * _depth_to_pcl / _sample_points      datasets/load_data.py:308-333
  * generate_RT(mode='vec') tools/geom_utils.py:232-244 (+ tools/rot_utils.py:39-100)
The reference functions are imported from /root/reference (import stubs under oracle/stubs/ for the
packages the image lacks; ``np.float`` is re-aliased because the loader predates numpy 1.24), run on
closed-form hash inputs, compared with oracle/ref_cpu.py, and their OUTPUTS written to
tests/golden/frontend_*.npz (inputs are regenerated from the same hash fills by the tests:
``ref_cpu.frontend_inputs``).

usage:  python oracle/gen_golden_frontend.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import numpy.ma  # noqa: F401  (must be imported before the alias below)
import matplotlib.pyplot  # noqa: F401
import torch

np.float = float            # removed in numpy 1.24; load_data.py:325 still uses it

import config.config  # noqa: F401,E402
from absl import flags  # noqa: E402

FLAGS = flags.FLAGS
from network.point_sample.pc_sample import PC_sample as RefPCSample  # noqa: E402
from tools.geom_utils import generate_RT as ref_generate_RT  # noqa: E402
from datasets.load_data import PoseDataset as RefDataset  # noqa: E402

import ref_cpu as oc  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)


frontend_inputs = oc.frontend_inputs


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    mpath = os.path.join(GOLD, "manifest.json")
    man = json.load(open(mpath))
    man["files"][name] = {k: [list(v.shape), str(v.dtype)] for k, v in arrs.items()}
    json.dump(man, open(mpath, "w"), indent=1, sort_keys=True)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


# ---- PC_sample --------------------------------------------------------------------------------------
print("PC_sample")
FLAGS.random_points = 1028
FLAGS.sample_method = "basic"
B, H, W = 3, 64, 80
radii = [30.0, 12.0, 22.0]                     # ~2600 / ~410 (< 1028: with replacement) / ~1400 valid pixels
mask, depth, camK, coor = frontend_inputs(B, H, W, 900, radii)
np.random.seed(7)
pc_ref = RefPCSample(mask, depth, camK, coor)
pc_orc = oc.pc_sample(mask, depth, camK, coor, 1028, np.random.RandomState(7))
assert torch.equal(pc_ref, pc_orc), (pc_ref - pc_orc).abs().max()
counts = np.array([oc.valid_pixels(mask[b], depth[b]).numel() for b in range(B)], np.int32)
print("  valid pixels per image:", counts)
# predicted-mask form: two logit channels (pc_sample.py:16-18)
logits = torch.cat([oc.hash_tensor((B, 1, H, W), 950, 1.0), 2.0 * mask - 1.0 + oc.hash_tensor((B, 1, H, W), 951, 0.5)], dim=1)
np.random.seed(8)
pc_ref2 = RefPCSample(logits, depth, camK, coor)
pc_orc2 = oc.pc_sample(logits, depth, camK, coor, 1028, np.random.RandomState(8))
assert torch.equal(pc_ref2, pc_orc2)
# an image with <= 1 valid pixel: the reference returns the pair (None, None)
empty = mask.clone()
empty[1] = 0
empty[1, 0, 3, 4] = 1.0
d2 = depth.clone()
d2[1, 0, 3, 4] = 700.0
np.random.seed(9)
r = RefPCSample(empty, d2, camK, coor)
assert isinstance(r, tuple) and r == (None, None)
assert oc.pc_sample(empty, d2, camK, coor, 1028, np.random.RandomState(9)) is None
save("frontend_pc_sample", counts=counts, pc=pc_ref.numpy(), pc_logits=pc_ref2.numpy(),
     radii=np.array(radii, np.float32))

# ---- loader-side cloud extraction ---------------------------------------------------------------------
print("_depth_to_pcl / _sample_points")
K64 = np.array([[591.0125, 0.0, 322.525], [0.0, 590.16775, 244.11084], [0.0, 0.0, 1.0]], dtype=np.float64)   # REAL275 intrinsics
outs = {}
for b, seed in ((0, 11), (1, 12)):
    dnp, xy, m = depth[b].numpy(), coor[b].numpy(), mask[b].numpy()
    pcl_ref = RefDataset._depth_to_pcl(None, dnp, K64, xy, m) / 1000.0
    pcl_orc = oc.depth_to_pcl(dnp, K64, xy, m) / 1000.0
    assert pcl_ref.dtype == np.float32 and np.array_equal(pcl_ref, pcl_orc), np.abs(pcl_ref - pcl_orc).max()
    np.random.seed(seed)
    samp_ref = RefDataset._sample_points(None, pcl_ref, 1028)
    samp_orc = oc.sample_points(pcl_orc, 1028, np.random.RandomState(seed))
    assert np.array_equal(samp_ref, samp_orc)
    outs[f"pcl{b}"] = samp_ref.astype(np.float32)
    print(f"  image {b}: {pcl_ref.shape[0]} valid -> {samp_ref.shape}")
save("frontend_depth_to_pcl", K=K64, **outs)

# ---- generate_RT -----------------------------------------------------------------------------------------
print("generate_RT")
pg, pr, fg, fr, T, sym = oc.generate_rt_inputs()
rt_ref = ref_generate_RT([pg, pr], [fg, fr], T, mode="vec", sym=sym)
rt_orc = oc.generate_rt(pg, pr, fg, fr, T, sym)
err = (rt_ref - rt_orc).abs().max().item()
print("  oracle vs reference:", err)
assert err < 1e-6
RtR = rt_ref[:, :3, :3].transpose(1, 2) @ rt_ref[:, :3, :3]
print("  |R^T R - I| max:", (RtR - torch.eye(3)).abs().max().item())
save("frontend_generate_rt", rt=rt_ref.numpy())
print("done")
