#!/bin/bash
# Counters of the persistent LSTM sweeps (own passes, kernel-trace only): tools/lstm_pmc.sh <outdir>
# pass 1: how busy the matrix pipe is inside a sweep; pass 2: what the waves wait on.
set -u
OUT="${1:-gpurun_out/lstm_pmc}"; mkdir -p "$OUT"
R="$(pwd)"; export TMPDIR=/tmp
SHAPES="80x512x512 40x512x512 80x64x512"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/pmc_a" -o p -- python "$R/tools/lstm_probe.py" $SHAPES > "$R/$OUT/probe_a.jsonl" 2> "$R/$OUT/probe_a.err"); echo "pass a rc=$?"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$R/$OUT/pmc_b" -o p -- python "$R/tools/lstm_probe.py" $SHAPES > "$R/$OUT/probe_b.jsonl" 2> "$R/$OUT/probe_b.err"); echo "pass b rc=$?"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
res = {}
for tag in ("a", "b"):
    f = glob.glob(os.path.join(out, "pmc_" + tag, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        res[tag] = "no counter csv: " + open(os.path.join(out, "probe_%s.err" % tag)).read()[-400:]; continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "k_lstm_seq" not in n and "k_lstm_cell" not in n and "Cijk" not in n: continue
        n = n.split("(")[0][:48] + " grid=" + r.get("Grid_Size", "?")
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[n] += 1; dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tab = {}
    for n, c in acc.items():
        e = {"dispatches": calls[n], "avg_us_under_counters": round(dur[n] / max(calls[n], 1), 1)}
        if tag == "a" and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_per_gui_active_cycle"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["GRBM_GUI_ACTIVE"], 3)
            e["sq_busy_per_gui_active_cycle"] = round(c.get("SQ_BUSY_CYCLES", 0) / c["GRBM_GUI_ACTIVE"], 3)
        if tag == "b" and c.get("SQ_WAVE_CYCLES"):
            e["wait_any_per_wave_cycle"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
            e["wait_lds_per_wave_cycle"] = round(c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"], 3)
            e["lds_bank_conflict_per_wave_cycle"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_WAVE_CYCLES"], 4)
        tab[n] = e
    res[tag] = tab
res["how"] = ("rocprofv3 --pmc <counters> --kernel-trace (two own passes) over tools/lstm_probe.py 80x512x512 40x512x512 80x64x512; ratios of the "
              "summed counters per kernel and grid; SQ_VALU_MFMA_BUSY_CYCLES is reported per SIMD-quad on gfx950 as in profiles/r04_conv_backward_pmc.json")
print(json.dumps(res, indent=1)); json.dump(res, open(os.path.join(out, "lstm_pmc.json"), "w"), indent=1)
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete
