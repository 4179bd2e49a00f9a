"""The batch split rule against the plain 512-workgroup target over many shapes: python profiles/tools/batch_rule_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
shapes = ((128, 2048), (64, 8192), (32, 4096), (16, 8192), (32, 32768), (8, 32768), (100, 4096), (200, 8192), (256, 8192), (48, 16384),
          (4, 131072), (256, 2048), (512, 1024), (24, 8192), (300, 4096), (1, 131072), (7, 16384), (160, 4096), (96, 8192), (256, 32768 // 4))
for n_seq, T in shapes:
    row = []
    for t in (os.environ.get("AB_COSTS", "16,6").split(",")):
        os.environ["SPECKV_FP8_MERGE_COST"] = t
        r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=4)
        v = list(r.values())[0]
        row.append("%.3f" % v.get("frac_hbm", 0))
    print(n_seq, T, "merge cost", os.environ.get("AB_COSTS", "16,6"), row)
