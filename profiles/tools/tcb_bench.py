"""The batched tensor codec at the reference's call size: python profiles/tools/tcb_bench.py [n_tensors] [n]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
r = bench.tensor_codec_batch_extra(torch, kv.lib, nt, n)
print(json.dumps(r, indent=1))
kv.close()
