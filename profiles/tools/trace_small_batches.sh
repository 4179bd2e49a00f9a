export TMPDIR=/tmp
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o p -- python3 profiles/tools/batch_over_cus.py 4,3,5 8,32 0 > /tmp/pt.log 2>&1 < /dev/null
grep "^[345] " /tmp/pt.log | awk '{print $1,$2,$3,$6,$7,$9,$10}'
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("/tmp/pt/p_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 400 dispatches per attention kernel family: print durations and gap to the following kernel
by=collections.defaultdict(list)
for i,r in enumerate(rows[:-1]):
    n=r["Kernel_Name"]
    if "k_attend" in n:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
        nx=rows[i+1]
        gap=(int(nx["Start_Timestamp"])-int(r["End_Timestamp"]))/1e3
        by[(n.split("(")[0][-60:],r.get("Grid_Size_X") or r.get("Grid_Size"),r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append((d,gap,nx["Kernel_Name"].split("(")[0][-30:]))
for k,v in by.items():
    v=v[-20:]
    ds=sorted(x[0] for x in v); gs=sorted(x[1] for x in v)
    print(k,len(v),"dur med %.1f us, gap to next med %.1f us, next=%s"%(ds[len(ds)//2],gs[len(gs)//2],v[-1][2]))
PY
