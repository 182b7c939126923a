#!/bin/bash
# round 4, session U: PMC counters (own pass, kernel-trace only) for the conv stack's backward kernels, both pipes
set -u
R="$(pwd)"; OUT=gpurun_out/r04u; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH="$R"
(cd /tmp && timeout 300 python "$R/tools/conv_bwd_probe.py" 40960 1 > /dev/null 2>&1)      # MIOpen find outside the counter run
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$R/$OUT/pmc" -o c -- python "$R/tools/conv_bwd_probe.py" 40960 3 > /dev/null 2> "$R/$OUT/pmc.err"); echo "pmc rc=$?"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = {}
keys = ("k_conv2_bwd_data", "k_conv3_bwd_data", "k_conv_wrw_b3", "k_conv1_u8_wrw", "igemm_bwd", "igemm_wrw")
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n_disp = collections.Counter(); seen = set(); dur = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not any(k in n for k in keys): continue
        n = n.split("(")[0][:60]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); n_disp[n] += 1; dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for n, c in acc.items():
        d = {k: v / n_disp[n] for k, v in c.items()}
        d["avg_ms_under_counters"] = round(dur[n] / n_disp[n], 4); d["dispatches"] = n_disp[n]
        if d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_fraction"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / d["GRBM_GUI_ACTIVE"] / 128.0, 3)   # same normalisation as profiles/r03_conv3_pmc.json
            d["clock_GHz"] = round(d["GRBM_GUI_ACTIVE"] / (d["avg_ms_under_counters"] * 1e6) / 8.0, 3)   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        res[n] = d
print(json.dumps(res, indent=1)); json.dump(res, open(os.path.join(out, "conv_bwd_pmc.json"), "w"), indent=1)
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
