#!/bin/bash
# read requests in flight and their latency for one kernel of a script: inflight.sh <name-substring> <tag> -- <script args>
# (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ = average latency in cycles; LEVEL / (GRBM_GUI_ACTIVE / 8 XCDs) = requests in flight)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
KN=$1; TAG=$2; shift 3
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/p -- python3 $R/$* > $OUT/p.log 2>&1
python3 - $OUT "$KN" <<'PY'
import csv, sys, glob, collections, os
out, kn = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for fn in glob.glob(os.path.join(out, "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
if m:
    gui = m["GRBM_GUI_ACTIVE"] / 8.0
    print(kn, "launches", len(agg["GRBM_GUI_ACTIVE"]), "cycles/XCD %.0f" % gui, "| reads %.2f M, in flight %.1f k, latency %.0f cycles" % (m["TCC_EA0_RDREQ_sum"] / 1e6, m["TCC_EA0_RDREQ_LEVEL_sum"] / gui / 1e3, m["TCC_EA0_RDREQ_LEVEL_sum"] / max(1, m["TCC_EA0_RDREQ_sum"])),
          "| writes %.2f M, in flight %.1f k, latency %.0f" % (m.get("TCC_EA0_WRREQ_sum", 0) / 1e6, m.get("TCC_EA0_WRREQ_LEVEL_sum", 0) / gui / 1e3, m.get("TCC_EA0_WRREQ_LEVEL_sum", 0) / max(1, m.get("TCC_EA0_WRREQ_sum", 1))))
else:
    print("no kernel matching", kn)
PY
