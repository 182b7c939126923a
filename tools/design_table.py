#!/usr/bin/env python3
"""bench.py JSON line -> the markdown kernel table of DESIGN.md section 3 (every librltime_hip kernel of one learner step
with its bound, its roofline time and what it reaches).  usage: python tools/design_table.py profiles/r06_bench_final.json"""
import json
import sys


def main(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print("| kernel | launches / step | avg µs | ms / step | bound | roofline µs | fraction of its roofline |")
    print("|---|---|---|---|---|---|---|")
    for k in d["roofline_all"]["kernels"]:
        bound = k.get("bound", "")
        if bound == "mfma":
            bound = "MFMA (%s: %.0f TFLOP/s)" % (k.get("pipe"), k.get("peak_TFLOPs", 0))
        elif bound == "hbm":
            bound = "HBM (8 TB/s)"
        frac = k.get("frac_of_roofline")
        extra = ""
        if k.get("achieved_TFLOPs"):
            extra = " (%.0f TFLOP/s)" % k["achieved_TFLOPs"]
        elif k.get("achieved_GBps"):
            extra = " (%.2f TB/s)" % (k["achieved_GBps"] / 1e3)
        print("| `%s` | %g | %.1f | %.3f | %s | %s | %s |" % (
            k["kernel"], k["launches_per_step"], k["avg_us"], k["ms_per_step"], bound,
            ("%.1f" % k["roofline_us"]) if k.get("roofline_us") is not None else "—",
            ("**%.2f**%s" % (frac, extra)) if frac is not None else "—"))
    rs = d.get("roofline_step") or {}
    print("\nstep %.2f ms; librltime_hip kernels %.1f ms measured / %.1f ms of roofline time; whole-step fraction %.3f" % (
        d["ms_per_step"], rs.get("librltime_hip_measured_ms", 0), rs.get("librltime_hip_roofline_ms", 0), rs.get("frac", 0)))


if __name__ == "__main__":
    main(sys.argv[1])
