"""development aid: prints the kernel timeline of the timed bench steps from a rocprofv3 --kernel-trace csv
usage: python tools/step_timeline.py <dir with *_kernel_trace.csv> [first row] [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = [r for r in csv.DictReader(open(f)) if "icpx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:8.1f} us) {r['Kernel_Name'][11:45]}")
