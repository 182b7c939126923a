"""Ordered kernel list of the last full bench step in a rocprofv3 --kernel-trace csv: python tools/trace_window.py <dir> [marker-kernel-substring]
(the marker is a kernel that runs once per step; default k_gather_scalars)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_gather_scalars"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
print("step window: %d kernels, %.1f us wall" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    g = "%sx%sx%s/%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"))
    print("%9.1f %8.1f  %-22s %s" % ((s - t0) / 1e3, (e - s) / 1e3, g, r["Kernel_Name"][:110]))
print("busy us", busy / 1e3)
