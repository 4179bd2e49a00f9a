# one shape of the FP8 / INT4 batch attention (bench.batch_attention_extra): python profiles/tools/fp8_batch_one.py <n_seq> <T> [scheme=4]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
n_seq, T = int(sys.argv[1]), int(sys.argv[2])
scheme = int(sys.argv[3]) if len(sys.argv) > 3 else 4
r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=scheme)
print("batch", n_seq, T, scheme, {k: v for k, v in list(r.values())[0].items() if k in ("ms_per_layer", "frac_hbm", "error")})
