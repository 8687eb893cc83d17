"""Development aid: timeline (start, end, gap to the previous kernel, duration in us) of the kernels of a rocprofv3
--kernel-trace run, from the LAST launch of the kernel whose name contains <anchor> on.
usage: kernel_timeline.py <rocprof output dir> [anchor = grid_bbox] [max rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "grid_bbox"
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
s = idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
prev_end = t0
for r in rows[s:s + limit]:
    st = int(r["Start_Timestamp"]) - t0
    en = int(r["End_Timestamp"]) - t0
    print(f"{st / 1e3:8.1f} {en / 1e3:8.1f} gap={(st - prev_end) / 1e3:6.1f} dur={(en - st) / 1e3:6.1f} {r['Kernel_Name'][:70]}")
    prev_end = en
