#!/usr/bin/env python3
"""Copy one evidence run from gpurun_out/ into profiles/ under a tag.

    python profiles/publish_r02.py r02h prof_r02h r2h_bench_n1.json r2h_bench_n2fake.json [r2h_bench_driver_cmd.json]

prof dir = output of profiles/collect_r02.sh (kernel traces + PMC passes + summary.json); the bench files are the
stdout of `python bench.py` (default), of the two-ranks-on-one-GPU run, and of the driver's command line."""
import glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prof, bench1, bench2 = sys.argv[1:5]
drv = sys.argv[5] if len(sys.argv) > 5 else None
g = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def one_line(src, dst):
    s = open(src).read()
    s = s[s.index('{"metric'):]
    json.loads(s.strip().splitlines()[-1])
    open(dst, "w").write(s.strip().splitlines()[-1] + "\n")


s = json.load(open(os.path.join(g, prof, "summary.json")))
pm = s["pmc"]


def m(t, c):
    for k, v in pm.items():
        if k.startswith("pmc_" + t + "_") and c in v:
            return v[c]["mean"]
    raise KeyError((t, c))


fp = {}
for t, name in (("4096x32", "cfg1 (131072 blocks)"), ("16384x32", "4 x cfg1 (524288 blocks)"), ("8192x80", "70B-shaped sequence (655360 blocks)")):
    fp[name] = {"FETCH_SIZE_KiB": m(t, "FETCH_SIZE"), "WRITE_SIZE_KiB": m(t, "WRITE_SIZE"), "TCC_HIT": m(t, "TCC_HIT_sum"),
                "TCC_MISS": m(t, "TCC_MISS_sum"), "TCC_EA0_RDREQ": m(t, "TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ": m(t, "TCC_EA0_WRREQ_sum")}
c = fp["cfg1 (131072 blocks)"]
rd, wr = int(c["FETCH_SIZE_KiB"] * 1024 * 2), int(c["WRITE_SIZE_KiB"] * 1024)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on profiles/tools/footprint.py 4096 32 (the bench's cfg1 launch), profiles/collect_r02.sh",
       "units": "FETCH_SIZE and WRITE_SIZE in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B)",
       "pmc": {"k_fetch_decompress<2, 0, false>": {"FETCH_SIZE_KiB": c["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": c["WRITE_SIZE_KiB"],
                                                    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                                                    "hbm_traffic_bytes_per_launch": rd + wr}},
       "footprints": fp}
json.dump(out, open(os.path.join(P, f"{tag}_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(g, prof, "summary.json"), os.path.join(P, f"{tag}_summary.json"))
one_line(os.path.join(g, prof, "bench_ascalled.json"), os.path.join(P, f"{tag}_bench_ascalled.json"))
one_line(os.path.join(g, prof, "bench_full.json"), os.path.join(P, f"{tag}_bench_full.json"))
one_line(os.path.join(g, bench1), os.path.join(P, f"{tag}_bench.json"))
one_line(os.path.join(g, bench2), os.path.join(P, f"{tag}_bench_2ranks_one_gpu.json"))
if drv:
    one_line(os.path.join(g, drv), os.path.join(P, f"{tag}_bench_driver_cmd.json"))
for t in ("ascalled", "full"):
    f = sorted(glob.glob(os.path.join(g, prof, f"trace_{t}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    shutil.copy(f[-1], os.path.join(P, f"{tag}_kernel_stats_{t}.csv"))      # the newest, should an older run's output remain
if tag.startswith("r02"):                                            # (round-2 side files; later rounds publish their own)
    for extra, name in (("r2h_conn_step.txt", "connector_step.txt"), ("r2h_batch_rule_ab.txt", "fp8_batch_split_rule_ab.txt")):
        if os.path.exists(os.path.join(g, extra)):
            shutil.copy(os.path.join(g, extra), os.path.join(P, f"{tag}_{name}"))
print("published", tag)
