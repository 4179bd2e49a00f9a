#!/bin/bash
# kernel durations and grids of a ragged batch: bash profiles/tools/trace_ragged.sh <schemes> <n_seq> <lo> <hi> [DIST]
export TMPDIR=/tmp
rm -rf /tmp/pt2
DIST=${5:-tail} timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt2 -o p -- python3 profiles/tools/batch_ragged.py ${1:-4} ${2:-64} ${3:-1024} ${4:-32768} 0 > /tmp/pt2.log 2>&1 < /dev/null
grep "^[345] " /tmp/pt2.log
python3 - <<'PY'
import csv,collections
rows=list(csv.DictReader(open("/tmp/pt2/p_kernel_trace.csv")))
by=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "k_attend" in n:
        by[(n.split("(")[0][-48:],r.get("Grid_Size_X"),r.get("Grid_Size_Y"),r.get("Workgroup_Size_X"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in by.items():
    v=sorted(v); print(k,len(v),"med %.1f us"%v[len(v)//2])
PY
