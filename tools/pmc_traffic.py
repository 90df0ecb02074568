"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into profiles/<round>/traffic.json, keyed by the
bench.py roofline kernel key.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv \
                                gpurun_out/pmc_write/w_counter_collection.csv profiles/r01/traffic.json

Units / corrections (guide, section HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads, so reads are doubled:  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.
"""
import collections
import csv
import json
import sys

# rocprof kernel name fragment + flattened grid size  ->  bench.py key   (B=16, N=1028 workload)
MAP = {
    ("knn_feat_kernel<21, true", "131072"): "hsp_knn_f32[B16N1028C128k20]",
    ("knn_feat_kernel<9, true", "8192"): "hsp_knn_f32[B16N64C256k8]",
    ("knn3_wave_kernel<17", "266240"): "hsp_geometry_all_f32[B16N1028/257/64k20]",
    ("rf_fwd_pipe_kernel<true, 1, false, float", "524288"): "hsp_rf_surface_fwd[B16N1028k20S7C128]",
    ("rf_fwd_pipe_kernel<false, 1, true, float", "524288"): "hsp_rf_conv_fwd[B16N1028k20S7C128]",
    ("rf_fwd_pipe_kernel<false, 2, false, float", "524288"): "hsp_rf_conv_fwd[B16N257k20S7C256]",
    ("rf_fwd_pipe_kernel<false, 4, false, float", "262144"): "hsp_rf_conv_fwd[B16N64k8S7C512]",
    ("rf_bwd_tile_kernel<16, true, false, float", "458752"): "hsp_rf_surface_bwd[B16N1028S7C128]",
    ("rf_bwd_tile_kernel<16, false, true, float", "458752"): "hsp_rf_conv_bwd_scatter[B16N1028S7C128]",
    ("rf_bwd_tile_kernel<32, false, false, float", "458752"): "hsp_rf_conv_bwd_scatter[B16N257S7C256]",
    ("rf_bwd_tile_kernel<64, false, false, float", "458752"): "hsp_rf_conv_bwd_scatter[B16N64S7C512]",
    ("concat_rows_kernel<float", "2097152"): "hsp_concat_rows[B16N1028W1286]",
    # bf16 feature storage, B=64, N=4096 (BASELINE configs[3])
    ("knn_feat_bf16_kernel<21", "2097152"): "hsp_knn_bf16[B64N4096C128k20]",
    ("rf_fwd_pipe_kernel<false, 1, true, unsigned short", "524288"): "hsp_rf_conv_fwd_bf16[B64N4096k20S7C128]",
    ("rf_fwd_pipe_kernel<true, 1, false, unsigned short", "524288"): "hsp_rf_surface_fwd_bf16[B64N4096k20S7C128]",
    ("rf_bwd_tile_kernel<16, false, true, unsigned short, 2>", "3670016"): "hsp_rf_conv_bwd_scatter_bf16[B64N4096S7C128]",
}       # (the weight-gradient and N=257 feature-KNN launches share one grid size between shapes: not separable here)


def agg(path, counter):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a[(r["Kernel_Name"], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in a.items()}


def total(path, counter):
    """(sum of the counter over EVERY dispatch of the run, dispatches of libhsp kernels)"""
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            t += float(r["Counter_Value"])
            n += 1 if ("hsp::" in r["Kernel_Name"] or "gather_rows" in r["Kernel_Name"]) else 0
    return t, n


def main(fetch_csv, write_csv, out_json, valu_csv=None, steps=None, clouds=None, tag="step"):
    """steps / clouds: the profiled run's step count (bench.py: warm-up + K + the steps added for the >= 50-step median) and clouds
    per step -- with them the file also carries the STEP-level measured traffic (SURVEY 8d asks for the measured fraction next to the
    algorithmic one): sum over every dispatch of 2 * FETCH_SIZE + WRITE_SIZE, per step and per cloud.  valu_csv: a third pass with
    SQ_INSTS_VALU -> VALU wave-instructions per launch of the issue-bound kernels (bench.py prices them on the instruction floor)."""
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    v = agg(valu_csv, "SQ_INSTS_VALU") if valu_csv else {}
    out = {}
    for (name, grid), fv in f.items():
        for (frag, g), key in MAP.items():
            if frag in name and g == grid:
                wv = w.get((name, grid), 0.0)
                out[key] = {"FETCH_SIZE_KiB": round(fv, 1), "WRITE_SIZE_KiB": round(wv, 1),
                            "hbm_bytes_per_launch": int((2 * fv + wv) * 1024),
                            "correction": "reads doubled (gfx950 FETCH_SIZE counts 64 B per 128 B request)"}
                if (name, grid) in v:
                    out[key]["valu_wave_insts_per_launch"] = int(v[(name, grid)])
    if steps and clouds:
        ft, _ = total(fetch_csv, "FETCH_SIZE")
        wt, _ = total(write_csv, "WRITE_SIZE")
        steps, clouds = int(steps), int(clouds)
        per_step = (2 * ft + wt) * 1024 / steps
        out["__%s__" % tag] = {"profiled_steps": steps, "clouds_per_step": clouds, "FETCH_SIZE_KiB_total": round(ft, 1),
                               "WRITE_SIZE_KiB_total": round(wt, 1), "measured_hbm_bytes_per_step": int(per_step),
                               "measured_hbm_bytes_per_cloud": int(per_step / clouds),
                               "how": "sum over every dispatch of the PMC runs of (2 * FETCH_SIZE + WRITE_SIZE) KiB / profiled steps"}
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(*sys.argv[1:8])
