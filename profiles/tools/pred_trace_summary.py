"""Durations and gaps of the prediction kernels in a rocprofv3 --kernel-trace CSV of profiles/tools/pred_trace.py."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "k_predict_small" in r["Kernel_Name"]]
# the 20 timed back-to-back single predictions: the last 40 k_predict_small* dispatches before the batch kernels
pairs = [(a, b) for a, b in zip(ks, ks[1:]) if "merge" not in a["Kernel_Name"] and "merge" in b["Kernel_Name"]]
pairs = pairs[-20:]
d0 = [int(a["End_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in pairs]
d1 = [int(b["End_Timestamp"]) - int(b["Start_Timestamp"]) for a, b in pairs]
g = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in pairs]
gn = [int(n[0]["Start_Timestamp"]) - int(p[1]["End_Timestamp"]) for p, n in zip(pairs, pairs[1:])]
avg = lambda v: sum(v) / max(1, len(v)) / 1e3
print("k_predict_small %.2f us | gap %.2f | k_predict_small_merge %.2f | gap to the next prediction %.2f | period %.2f" % (avg(d0), avg(g), avg(d1), avg(gn), avg(d0) + avg(g) + avg(d1) + avg(gn)))
# the batch path (256 requests): the last 20 rounds of its four kernels
names = ("k_lstm_hidden", "k_lstm_logits_topk", "k_lstm_logits", "k_softmax_topk_small", "k_softmax_topk_merge", "k_predict_small_merge")
first_batch = next(i for i, r in enumerate(rows) if "k_lstm_hidden" in r["Kernel_Name"])
bk = [r for r in rows[first_batch:] if any(n + "<" in r["Kernel_Name"] or n + "(" in r["Kernel_Name"] for n in names)]
per = 3 if any("k_lstm_logits_topk" in r["Kernel_Name"] for r in bk) else 4
last = bk[-20 * per:]
import collections
dur = collections.defaultdict(list)
for r in last:
    n = next(n for n in names if n + "<" in r["Kernel_Name"] or n + "(" in r["Kernel_Name"])
    dur[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(" | ".join("%s %.2f us" % (n, avg(dur[n])) for n in names if dur[n]), "| span per round %.2f us" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 20e3))
