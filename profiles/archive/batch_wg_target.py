"""Workgroup target sweep of the batch attention split rule: python profiles/tools/batch_wg_target.py [scheme]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
shapes = ((128, 2048), (64, 8192), (32, 4096), (16, 8192), (32, 32768), (8, 32768), (100, 4096), (200, 8192), (256, 8192), (48, 16384), (4, 131072))
targets = (192, 256, 320, 384, 512, 768, 1024)
print("shape", targets)
for n_seq, T in shapes:
    row = []
    for t in targets:
        os.environ["SPECKV_ATTEND_WG_TARGET"] = str(t)
        r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=scheme)
        v = list(r.values())[0]
        row.append("%.3f" % v.get("frac_hbm", 0))
    print(n_seq, T, " ".join(row))
