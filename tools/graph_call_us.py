"""In-graph duration of the step's longest C-ABI calls, from the committed per-shape kernel table of a graph-replay trace:
    python tools/graph_call_us.py profiles/r05/k_per_shape_kernel_us.txt > profiles/r05/graph_calls.json
A C-ABI call is one or more kernels; bench.py times calls with HIP events around them in an EAGER re-issue of the step (events
cannot be recorded inside a replay), which adds the gaps between a call's kernels to it.  This table gives bench.py the kernels'
own durations inside the replayed graph (sum over the call's kernels), so that the `roofline` object is priced -- and the dominant
call chosen -- on what the timed region actually ran.  The map below (call key -> kernels by name prefix and grid) covers the calls
that can lead the B=16, N=1028 fp32 step."""
import json
import sys

CALLS = {
    "hsp_knn_f32[B16N1028C128k20]": [("hsp::knn_feat_kernel<21, true>", "8192x16"), ("hsp::knn_feat_sym_tail_kernel", "256x16"),
                                     ("hsp::quad32_kernel", "526336x1")],
    "hsp_rf_conv_fwd[B16N1028k20S7C128]": [("hsp::rf_fwd_pipe_kernel<false, 1, true, float>", "524288x1")],
    "hsp_rf_conv_bwd_scatter[B16N1028S7C128]": [("hsp::rf_bwd_tile_kernel<16, false, true, float, 1>", "28672x16")],
    "hsp_rf_surface_fwd[B16N1028k20S7C128]": [("hsp::rf_fwd_pipe_kernel<true, 1, false, float>", "524288x1")],
    "hsp_geometry_all_f32[B16N1028/257/64k20]": [("hsp::knn3_wave_kernel<17>", "16640x16"), ("hsp::geometry_levels_kernel<5>", "504320x1")],
}


def main(path):
    rows = []
    for line in open(path):
        if " grid " not in line:
            continue
        name = line.split(" grid ")[0].replace("void ", "").strip()
        rest = line.split(" grid ")[1].split()
        rows.append((name, rest[0], float(rest[4]), float(rest[6])))         # name, grid, calls/step, avg us
    out = {}
    for call, kernels in CALLS.items():
        tot, found = 0.0, []
        for prefix, grid in kernels:
            hit = [r for r in rows if r[0].startswith(prefix[:40]) and r[1] == grid]
            if hit:
                # (a kernel shared by several calls of the step -- quad32 at this grid runs once per N = 1028 feature search)
                tot += hit[0][3]
                found.append({"kernel": hit[0][0], "grid": grid, "avg_us": hit[0][3]})
        if found:
            out[call] = {"in_graph_us": round(tot, 1), "kernels": found}
    json.dump({"source": path, "calls": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
