import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last occurrence of grid_bbox: print timeline from there
idx=[i for i,r in enumerate(rows) if "grid_bbox" in r["Kernel_Name"]]
s=idx[-1]
t0=int(rows[s]["Start_Timestamp"])
prev_end=t0
for r in rows[s:]:
    st=int(r["Start_Timestamp"])-t0; en=int(r["End_Timestamp"])-t0
    print(f"{st/1e3:8.1f} {en/1e3:8.1f} gap={(st-prev_end)/1e3:6.1f} dur={(en-st)/1e3:6.1f} {r['Kernel_Name'][:70]}")
    prev_end=en
