"""Split length sweep of the batch attention (one layer): python profiles/tools/fp8_batch_tps.py [scheme]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
shapes = ((128, 2048), (64, 8192), (512, 1024), (32, 4096), (16, 8192), (256, 2048))
if os.environ.get("SHAPES") == "odd":
    shapes = ((100, 4096), (48, 16384), (24, 8192), (7, 16384), (96, 8192), (200, 8192), (32, 32768))
for n_seq, T in shapes:
    row = []
    for tps in (None, 8, 12, 16, 24, 32, 43, 48, 64, 86, 96, 128, 171, 256):
        if tps is None: os.environ.pop("SPECKV_ATTEND_TILES_PER_SPLIT", None)
        else: os.environ["SPECKV_ATTEND_TILES_PER_SPLIT"] = str(tps)
        if tps is not None and tps > T // 32: continue
        r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=scheme)
        v = list(r.values())[0]
        row.append("%s:%.3f" % (tps, v.get("frac_hbm", 0)))
    print(n_seq, T, " ".join(row))
