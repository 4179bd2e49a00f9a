import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(os.environ.get("SPECKV_LIB_PATH", pkg.library_path()), "hip:0")
for n in (131072 * 256, 1342177280):
    r = bench.tensor_codec_extra(torch, kv.lib, n)
    v = list(r.values())[0]
    print(n, {k: (v[k].get("frac_hbm") if isinstance(v.get(k), dict) else None) for k in ("compress", "decompress", "compress_fp32_source", "decompress_fp32_output")}, flush=True)
kv.close()
