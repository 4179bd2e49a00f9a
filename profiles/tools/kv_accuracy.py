"""The pool formats' cost in attention accuracy on KV-like data: python profiles/tools/kv_accuracy.py [T] [seed]  (cxl-speckv_amd/kv_accuracy.py)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cxl_speckv_amd as pkg
from cxl_speckv_amd.kv_accuracy import kv_format_accuracy, format_table
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7001
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
try:
    acc = kv_format_accuracy(kv, T=T, seed=seed)
finally:
    kv.close()
print(json.dumps(acc))
print(format_table(acc))
