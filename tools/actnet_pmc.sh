#!/bin/bash
# MFMA-busy counters of the acting network's kernels (own pass, kernel-trace only): tools/actnet_pmc.sh <outdir>
set -u
OUT="${1:-gpurun_out/actnet_pmc}"; mkdir -p "$OUT"
R="$(pwd)"; export TMPDIR=/tmp
for E in 32 256; do
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_$E" -o p -- python "$R/tools/actnet_probe.py" $E 1 > "$R/$OUT/probe_$E.jsonl" 2> "$R/$OUT/probe_$E.err"); echo "pmc E=$E rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = {}
for E in (32, 256):
    f = glob.glob(os.path.join(out, "pmc_%d" % E, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        res[str(E)] = "no counter csv"; continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "k_act_" not in n and "Cijk" not in n and "k_lstm_cell" not in n: continue
        n = n.split("(")[0][:60]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[n] += 1; dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res[str(E)] = {n: {"dispatches": calls[n], "avg_us_under_counters": round(dur[n] / max(calls[n], 1), 2),
                       "mfma_busy_per_gui_active_cycle": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["GRBM_GUI_ACTIVE"], 3) if c.get("GRBM_GUI_ACTIVE") else None,
                       "sq_busy_per_gui_active_cycle": round(c.get("SQ_BUSY_CYCLES", 0) / c["GRBM_GUI_ACTIVE"], 3) if c.get("GRBM_GUI_ACTIVE") else None}
                   for n, c in acc.items()}
res["how"] = "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace (own pass) over tools/actnet_probe.py E 1; ratios of the summed counters per kernel; SQ_VALU_MFMA_BUSY_CYCLES is reported per SIMD-quad on gfx950 as in profiles/r04_conv_backward_pmc.json"
print(json.dumps(res, indent=1)); json.dump(res, open(os.path.join(out, "actnet_pmc.json"), "w"), indent=1)
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
