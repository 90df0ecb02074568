"""
oracle/gen_golden_eval_loop.py -- TEST INFRASTRUCTURE ONLY.  Runs ONLY in the build container.

Pinned SURROGATE of BASELINE configs[4] (the REAL275 loop of evaluation/evaluate.py: dataset, detections and the published
checkpoint are not available offline, so its accuracy figures stay unpinned).  What the loop DOES is reproduced end to end with
the reference's own code:

  1. a checkpoint the reference would write: its HSPose built under FLAGS.train = 1 and torch.manual_seed(0) (train heads
     included, 160 tensors), BatchNorm running statistics moved off their defaults by closed-form fills (a trained model's are not
     0 / 1), saved as {'seed', 'epoch', 'posenet_state_dict', ...} (engine/train.py:117-126) to a scratch file;
  2. reloaded the way evaluation/evaluate.py:39,58-73 does it: FLAGS.train = False BEFORE construction, the three train-only
     heads' keys dropped, 'resconv' -> 'STE_layer', load_state_dict(strict=True), .eval();
  3. per "image" n_inst in {1, 4, 6} instances x 1028 points (one instance per image is a TILED short crop, load_data.py:314-316):
     network(PC=, obj_id=, mean_shape=, sym=) -> generate_RT(mode='vec') -> pred_s = Pred_s + mean_shape (evaluate.py:91-106).

tests/golden/eval_loop_1028.npz holds the per-image pred_RT / pred_s and network outputs plus samples + sums of every tensor of
the checkpoint (proving the mirrored modules, built under the same seed, write the identical file).  Inputs are closed form.
The reference's source never enters this repo.

usage:  python oracle/gen_golden_eval_loop.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, HERE]

import numpy as np
import torch

import config.config  # noqa: F401
from absl import flags

FLAGS = flags.FLAGS
import network.HSPose as RH
import network.fs_net_repo.gcn3d as rg
from tools.geom_utils import generate_RT

import ref_cpu as oc

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)

def main():
    # 1. the checkpoint
    FLAGS.train = 1
    torch.manual_seed(0)
    trained = RH.HSPose('PoseNet_only')
    sd = trained.state_dict()
    oc.eval_loop_move_bn_stats(sd)
    assert len(sd) == 160, len(sd)
    tmp = os.path.join(tempfile.mkdtemp(), "model_149.pth")
    torch.save({'seed': 0, 'epoch': 149, 'posenet_state_dict': sd, 'scheduler': {}, 'optimizer': {}}, tmp)
    arrs = {"meta": np.array([oc.EVAL_LOOP_SEED, 1028, 160], np.int64)}
    for k, v in sd.items():
        if v.is_floating_point():
            flat = v.reshape(-1)
            arrs["wsample." + k] = flat[::997].numpy().copy()
            arrs["wsum." + k] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
    # 2. evaluate.py:39,58-73
    FLAGS.train = False
    network = RH.HSPose('PoseNet_only')
    state_dict = torch.load(tmp)['posenet_state_dict']
    unnecessary_nets = ['posenet.face_recon.conv1d_block', 'posenet.face_recon.face_head', 'posenet.face_recon.recon_head']
    for key in list(state_dict.keys()):
        for net_to_delete in unnecessary_nets:
            if key.startswith(net_to_delete):
                state_dict.pop(key)
        if 'resconv' in key:
            state_dict[key.replace("resconv", "STE_layer")] = state_dict.pop(key)
    network.load_state_dict(state_dict, strict=True)
    network = network.eval()
    assert len(state_dict) == 107, len(state_dict)
    # eval BatchNorm divides by at::sqrt(var + eps) evaluated by the HOST's vector maths library (within an ulp, not correctly
    # rounded: host-specific); the three BatchNorms whose outputs are RANKED by a feature-space search are recorded so that a test
    # on another host can tell a differing sqrt from a differing kernel
    for nm in ("bn1", "bn2", "bn3"):
        bn = getattr(network.posenet.face_recon, nm)
        arrs["invstd." + nm] = (1 / torch.sqrt(bn.running_var + bn.eps)).numpy()
    # 3. the loop body, evaluate.py:90-106
    torch.manual_seed(1)                                           # Pool_layer randperm stream, consumed image after image
    for n_inst in sorted(oc.EVAL_LOOP_IMAGES):
        pts, obj, mean_shape, sym = oc.eval_loop_inputs(n_inst)
        lists = []
        orig = rg.get_neighbor_index

        def rec(vertices, neighbor_num):
            o = orig(vertices, neighbor_num)
            if vertices.shape[-1] != 3:
                lists.append(o.clone())
            return o
        rg.get_neighbor_index = rec
        try:
            with torch.no_grad():
                output_dict = network(PC=pts, obj_id=obj, mean_shape=mean_shape, sym=sym)
        finally:
            rg.get_neighbor_index = orig
        if n_inst == 4:
            for li, fi in enumerate(lists):
                arrs[f"n4.featknn{li + 1}"] = fi.numpy().astype(np.int16)
        p_green_R_vec, p_red_R_vec = output_dict['p_green_R'].detach(), output_dict['p_red_R'].detach()
        p_T, p_s = output_dict['Pred_T'].detach(), output_dict['Pred_s'].detach()
        f_green_R, f_red_R = output_dict['f_green_R'].detach(), output_dict['f_red_R'].detach()
        pred_s = p_s + mean_shape
        pred_RT = generate_RT([p_green_R_vec, p_red_R_vec], [f_green_R, f_red_R], p_T, mode='vec', sym=sym)
        arrs[f"n{n_inst}.pred_RT"] = pred_RT.numpy()
        arrs[f"n{n_inst}.pred_s"] = pred_s.numpy()
        for k in ("p_green_R", "p_red_R", "f_green_R", "f_red_R", "Pred_T", "Pred_s"):
            arrs[f"n{n_inst}.{k}"] = output_dict[k].detach().numpy()
        assert output_dict['recon'] is None
        print(f"image with {n_inst} instances: pred_RT {tuple(pred_RT.shape)}, |pred_s| {pred_s.abs().max().item():.3f}")
    os.remove(tmp)
    np.savez_compressed(os.path.join(GOLD, "eval_loop_1028.npz"), **arrs)
    print(f"eval_loop_1028: {os.path.getsize(os.path.join(GOLD, 'eval_loop_1028.npz')) / 1024:.1f} KiB")
    FLAGS.train = 1


if __name__ == "__main__":
    main()
