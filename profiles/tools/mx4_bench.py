"""MXFP4 fused attention at BASELINE configs[4] / configs[3] shapes: python profiles/tools/mx4_bench.py [single|batch|both] [n_seq] [T]
env: SPLITS (single form), TPS (batch form), SPECKV_LIB_PATH (an A/B build of the library)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
what = sys.argv[1] if len(sys.argv) > 1 else "both"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
kv = pkg.CxlSpeckvKVAllocator(os.environ.get("SPECKV_LIB_PATH", pkg.library_path()), "hip:0")
tag = os.environ.get("TAG", "base")
if what in ("single", "both"):
    for splits in os.environ.get("SPLITS", "0").split(","):
        bench.set_tuning("attend_splits", int(splits))
        r = bench.int4_attention_extra(torch, kv, 32768, 80, scheme=5)["mxfp4_fused_attention"]
        bench.set_tuning("attend_splits", 0)
        print(tag, "single 32k x 80 splits", splits, r.get("ms_all_layers"), r.get("frac_hbm"), r.get("error"), flush=True)
if what in ("batch", "both"):
    for tps in os.environ.get("TPS", "0").split(","):
        bench.set_tuning("attend_tiles_per_split", int(tps))
        r = list(bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=5).values())[0]
        bench.set_tuning("attend_tiles_per_split", 0)
        print(tag, f"batch {n_seq} x {T} tps", tps, r.get("ms_per_layer"), r.get("frac_hbm"), "planned", r.get("planned_ms_per_layer"), r.get("planned_frac_hbm"), r.get("error"), flush=True)
kv.close()
