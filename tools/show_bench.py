"""print the headline fields of a bench.py JSON line (last line of the file)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(d["value"], d["unit"], d["ms_per_step"], "ms; gemm", c["gemm"]["mode"], {k: v for k, v in c.items() if "ms_per" in k or "clouds_per" in k},
      "roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "avg_us", "traffic")}, "cpu", d.get("cpu_baseline", {}).get("value"))
