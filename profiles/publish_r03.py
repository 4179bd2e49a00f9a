#!/usr/bin/env python3
"""Copy one round-3 evidence run (profiles/tools/r3_final.sh <run>) from gpurun_out/ into profiles/ under a tag:
    python profiles/publish_r03.py r03a r3a
Reuses publish_r02.py for the common files, then adds the 8-ranks-on-one-GPU line (the 7-pool cfg4 leg executes) and the
striped-attention table."""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, run = sys.argv[1:3]
g, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(P, "publish_r02.py"), tag, f"prof_{run}", f"{run}_bench_n1.json", f"{run}_bench_n2fake.json",
                       f"{run}_bench_driver_cmd.json"])
s = open(os.path.join(g, f"{run}_bench_n8fake.json")).read()
s = s[s.index('{"metric'):].strip().splitlines()[-1]
open(os.path.join(P, f"{tag}_bench_8ranks_one_gpu.json"), "w").write(s + "\n")
for src, name in ((f"{run}_conn_step.txt", "connector_step.txt"), (f"{run}_striped.txt", "striped_attention.txt")):
    if os.path.exists(os.path.join(g, src)):
        shutil.copy(os.path.join(g, src), os.path.join(P, f"{tag}_{name}"))
print("published", tag)
