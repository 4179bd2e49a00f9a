"""Batch attention (one layer) at short contexts, FP8 and INT4: 256 sequences x {1k, 2k, 4k}, 512 x 1k.
python profiles/tools/batch_short.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for scheme in (4, 3):
    for n_seq, T in ((256, 1024), (256, 2048), (256, 4096), (512, 1024), (128, 2048)):
        r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=scheme)
        v = list(r.values())[0]
        print("fp8" if scheme == 4 else "int4", n_seq, T, v.get("ms_per_layer"), v.get("frac_hbm"), v.get("error", ""), flush=True)
