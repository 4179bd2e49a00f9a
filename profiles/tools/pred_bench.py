"""One-request and 256-request prediction latency (bench.predictor_extra): python profiles/tools/pred_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for _ in range(3):
    print("predictor", os.environ.get("SPECKV_LIB_PATH", "default").split("/")[-2:-1], bench.predictor_extra(torch, kv.lib)["token_predictor_top4"], flush=True)
