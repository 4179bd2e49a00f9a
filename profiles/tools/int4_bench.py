import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
if os.environ.get("SPLITS"): os.environ["SPECKV_ATTEND_SPLITS"] = os.environ["SPLITS"]
r = bench.int4_attention_extra(torch, kv, T, L)["int4_fused_attention"]
print("int4", os.environ.get("SPECKV_LIB_PATH", "default").split("/")[-2:-1], "T", T, "L", L, "splits", os.environ.get("SPLITS"), "ms", r.get("ms_all_layers"), "frac", r.get("frac_hbm"), r.get("error", ""))
