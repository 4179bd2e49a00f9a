"""Batch attention of more sequences than CUs, by pieces per sequence: python profiles/tools/batch_over_cus.py [schemes] [n_seqs] [tps list] [T]
(schemes / n_seqs / tps comma-separated; tps 0 = the engine's own rule)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
schemes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4,3").split(",")]
seqs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "256,260,300,384,512").split(",")]
tpss = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
T = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for scheme in schemes:
    for n in seqs:
        for tps in tpss:
            bench.set_tuning("attend_tiles_per_split", tps)
            r = list(bench.batch_attention_extra(torch, kv, n_seq=n, T=T, scheme=scheme).values())[0]
            bench.set_tuning("attend_tiles_per_split", 0)
            print(scheme, n, T, "tps", tps, r.get("ms_per_layer"), r.get("frac_hbm"), "planned", r.get("planned_ms_per_layer"), r.get("planned_frac_hbm"), r.get("error"), flush=True)
kv.close()
