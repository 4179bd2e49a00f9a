# the whole-tensor codec alone (bench.tensor_codec_extra): python profiles/tools/tc_bench.py [n]
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072 * 256
r = list(bench.tensor_codec_extra(torch, kv.lib, n).values())[0]
print("tensor codec n", n, json.dumps({k: r[k] for k in r if k != "note"}))
