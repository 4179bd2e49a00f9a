import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
os.environ["SPECKV_ATTEND_SPLITS"] = os.environ.get("SPLITS", "64")
r = bench.fp8_scores_extra(torch, kv, 32768, 80)
a = r["fp8_fused_attention"]
print(os.environ.get("SPECKV_LIB_PATH", "base").split("/")[-2] if "SPECKV_LIB_PATH" in os.environ else "base", a.get("ms_all_layers"), a.get("frac_hbm"), a.get("error"), "qk", r["fp8_qk_scores_mfma"].get("ms_all_layers"))
