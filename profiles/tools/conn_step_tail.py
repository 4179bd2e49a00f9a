import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for sch in ("fp8", "mxfp4"):
    for given in (0, 1):
        bench.set_tuning("attend_order_as_given", given)
        r = bench.connector_decode_extra(torch, kv, ctx=16384, T=16384 + 64, scheme=sch, tail=True)
        v = list(r.values())[0]
        print(sch, "as_given" if given else "rules", {k: v.get(k) for k in ("ms_per_step", "frac_hbm", "ms_per_step_layers_in_one_call", "frac_hbm_layers_in_one_call", "error", "context")}, flush=True)
bench.set_tuning("attend_order_as_given", 0)
kv.close()
