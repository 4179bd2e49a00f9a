"""The fused attention over a pool striped across 7 same-GPU "peers" (the 1 + 7 layout of BASELINE configs[3] on a one-GPU
box): striped form (computed addresses) against the page-table form (SPECKV_ATTEND_GENERAL=1) and the one-pool linear form.
    python profiles/tools/striped_bench.py [T=32768] [layers=80]"""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
T = sys.argv[1] if len(sys.argv) > 1 else "32768"
L = sys.argv[2] if len(sys.argv) > 2 else "80"
rows = []
for tool in ("int4_bench.py", "fp8_bench.py"):
    for label, env in (("linear (one pool)", {}), ("striped x7, computed addresses", {"SPECKV_POOL_DEVICES": "0,0,0,0,0,0,0"}),
                       ("striped x7, table form", {"SPECKV_POOL_DEVICES": "0,0,0,0,0,0,0", "SPECKV_ATTEND_GENERAL": "1"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(HERE, tool), T, L], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith(("int4", "fp8"))]
        rows.append({"tool": tool, "placement": label, "result": line[-1] if line else out.stderr[-300:]})
        print(rows[-1], flush=True)
