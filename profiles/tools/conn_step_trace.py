"""Timeline of one steady connector decode step from a rocprofv3 --kernel-trace CSV of profiles/tools/conn_step.py:
kernels in start order with duration and the idle gap in front of each.  usage: conn_step_trace.py <kernel_trace.csv> [step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("speckv::", "").replace("(anonymous namespace)::", "")
    return n[:70]
# steps are separated by host synchronisations: find k_flush_* groups as step starts
starts = [i for i, r in enumerate(rows) if "k_flush_candidates" in r["Kernel_Name"] or "k_flush_small" in r["Kernel_Name"]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# collapse consecutive flush kernels into one start
marks = [starts[0]] if starts else []
for a, b in zip(starts, starts[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 500000: marks.append(b)
lo, hi = marks[step], marks[step + 1]
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %8.1f  gap %6.1f  %s  grid %s wg %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(r["Kernel_Name"]), r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
    prev_end = max(prev_end, e); busy += e - s
print("step span %.1f us, kernel time %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
