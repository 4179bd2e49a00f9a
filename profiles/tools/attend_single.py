"""The fused attention of one 70B-shaped sequence (32k x 80 layers) for one pool format: python profiles/tools/attend_single.py [scheme=5] [T=32768] [layers=80]
env: SPECKV_POOL_DEVICES (striping), DUMMY_GB=a,b,.. (allocations made first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
dummy = [kv.lib.alloc(int(float(g) * 2 ** 30)) for g in os.environ.get("DUMMY_GB", "").split(",") if g]      # allocations in front of the measured one (placement experiments)
sch = int(sys.argv[1]) if len(sys.argv) > 1 else 5
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
L = int(sys.argv[3]) if len(sys.argv) > 3 else 80
r = bench.fp8_scores_extra(torch, kv, T, L) if sch == 4 else bench.int4_attention_extra(torch, kv, T, L, scheme=sch)
r = {k: v for k, v in r.items() if 'fused_attention' in k}
print({k: (v.get("ms_all_layers"), v.get("frac_hbm")) for k, v in r.items()})
kv.close()
