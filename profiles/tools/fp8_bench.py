import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
r = bench.fp8_scores_extra(torch, kv, T, L)
print("fp8", T, L, {k: (v.get("ms_all_layers"), v.get("frac_hbm")) for k, v in r.items()})
