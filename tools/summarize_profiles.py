"""Turns the rocprofv3 outputs that tools/collect_profiles.sh left under gpurun_out/<tag>_* into the
small summaries committed under profiles/: <tag>_bench.json (copy), <tag>_bench_kernel_stats.csv
(copy) and <tag>_counters.json (per-kernel, per-launch averages of every PMC counter, with the HBM
byte figures derived as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB).
usage: python tools/summarize_profiles.py <tag>"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
shutil.copy(os.path.join(go, f"{tag}_bench.json"), os.path.join(pr, f"{tag}_bench.json"))
stats = glob.glob(os.path.join(go, f"{tag}_trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(pr, f"{tag}_bench_kernel_stats.csv"))
stats = glob.glob(os.path.join(go, f"{tag}_c4_trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(pr, f"{tag}_c4_grid_kernel_stats.csv"))
for d, name in (("c4conv_trace", "c4_converge_kernel_stats.csv"), ("queen_trace", "queen_kernel_stats.csv")):
    stats = glob.glob(os.path.join(go, f"{tag}_{d}", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(pr, f"{tag}_{name}"))
per = defaultdict(lambda: defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_valu", "pmc_lds", "c4_pmc_fetch", "c4_pmc_write"):
    for f in glob.glob(os.path.join(go, f"{tag}_{d}", "*", "*_counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if "icpx::" in row["Kernel_Name"]:
                    per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {"_doc": "rocprofv3 --pmc passes (one counter group per run) over `bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-extras --verify-pairs 0 --no-executed-work` (pair kernels: the launches of the timed steps "
               "and the ten launched alone after them, same work each) and over `tools/c4_grid_probe.py 3` (grid kernels); per-launch averages. FETCH_SIZE / WRITE_SIZE are in KB; "
               "on gfx950 FETCH_SIZE counts half the bytes of wide coalesced streaming reads "
               "(MI355X_MICROARCH.md), so the raw and the fetch-doubled figure are both given.",
       "kernels": {}}
for k, cs in per.items():
    e = {c + "_per_launch": sum(v) / len(v) for c, v in cs.items()}
    e.update({c + "_median": sorted(v)[len(v) // 2] for c, v in cs.items()})
    e["launches_seen"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        fz, wz = e["FETCH_SIZE_per_launch"] * 1024, e["WRITE_SIZE_per_launch"] * 1024
        e["hbm_bytes_per_launch_raw"] = fz + wz
        e["hbm_bytes_per_launch_fetch_doubled"] = 2 * fz + wz
    if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs:
        e["valu_instructions_per_launch"] = e.get("SQ_INSTS_VALU_per_launch")
    out["kernels"][k] = e
with open(os.path.join(pr, f"{tag}_counters.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out, indent=1)[:3000])
