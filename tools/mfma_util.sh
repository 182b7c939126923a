#!/bin/bash
# MFMA utilisation of the PyTorch GEMM / conv kernels inside the learner step
# (BASELINE north_star: "MFMA used only for the conv/linear GEMMs ... evidenced by
# MFMA-utilisation counters").  Own PMC run, kernel-trace only.
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out/mfma"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
timeout 600 python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-acting --replay-size 120000 > "$OUT/prewarm.log" 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc" -o b -- \
  python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-acting --replay-size 120000 > "$OUT/run.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter csv; tail of run.log:"); print(open(os.path.join(out, "run.log")).read()[-1500:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); dur = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    fam = "hipBLASLt GEMM" if n.startswith("Cijk") else "MIOpen conv" if ("igemm" in n or "Conv" in n) else "librltime_hip" if "mirl::" in n else "other"
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); calls[fam] += 1; dur[fam] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
res = {}
for fam, c in acc.items():
    mf, busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0)
    res[fam] = {"dispatches": calls[fam], "kernel_ms": round(dur[fam], 2), "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CYCLES": busy,
                "GRBM_GUI_ACTIVE": gui, "mfma_busy_per_gui_active_cycle": (mf / gui) if gui else None}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(out, "mfma_util.json"), "w"), indent=1)
PY
find "$OUT" -name "*.csv" -size +1M -delete; find "$OUT" -name "*.db" -delete; du -sh "$OUT"
