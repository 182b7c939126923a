"""Where is the GPU idle inside a learner step?  Reads a rocprofv3 --kernel-trace csv (one row per dispatch with
Start_Timestamp / End_Timestamp), cuts the trace into learner steps at every `k_per_sample` (the step's first kernel),
and for the last steps prints busy time (union of dispatch intervals), idle time, and the largest idle gaps with the
kernels on either side."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "k_per_sample" in r[2]]
    print("dispatches", len(rows), "steps", len(cuts))
    for s in range(max(0, len(cuts) - 1 - last), len(cuts) - 1):
        seg = rows[cuts[s]:cuts[s + 1]]
        t0, t1 = seg[0][0], rows[cuts[s + 1]][0]
        busy, end = 0, t0
        gaps = []
        prev = None
        for a, b, name in seg:
            if a > end:
                gaps.append((a - end, prev, name))
                busy += b - a
            else:
                busy += max(0, b - end)
            if b > end:
                end, prev = b, name
        if t1 > end:
            gaps.append((t1 - end, prev, "next step k_per_sample"))
        print(f"\nstep {s}: wall {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, dispatches {len(seg)}")
        hist = defaultdict(lambda: [0, 0])
        for gdur, before, after in gaps:
            key = (before or "")[:50] + "  ->  " + after[:50]
            hist[key][0] += 1
            hist[key][1] += gdur
        for key, (n, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"   {tot / 1e3:9.1f} us in {n:4d} gaps   {key}")
        small = sum(gd for gd, _, _ in gaps if gd < 5000)
        print(f"   gaps < 5 us: {small / 1e3:.1f} us total of {sum(gd for gd, _, _ in gaps) / 1e3:.1f}")


if __name__ == "__main__":
    main()
