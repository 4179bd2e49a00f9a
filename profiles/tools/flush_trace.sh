#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
OUT=$R/gpurun_out/flush_trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t -- python3 $R/profiles/tools/flush_trace.py > $OUT/log.txt 2>&1
cat $OUT/log.txt | grep flush
python3 - $OUT <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
ev = []
for fn in glob.glob(os.path.join(out, "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
for fn in glob.glob(os.path.join(out, "t", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# last 30 events
t_prev = None
for s, e, n in ev[-30:]:
    gap = (s - t_prev) / 1e3 if t_prev else 0
    print(f"{n:62s} dur {(e - s) / 1e3:8.1f} us  gap {gap:8.1f} us")
    t_prev = e
PY
