"""Decode step of 256 sequences x 8 layers x 2k context through the connector, per pool format: python profiles/tools/conn_step_schemes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for sch in (sys.argv[1:] or ["fp8", "mxfp4", "int4"]):
    r = bench.connector_decode_extra(torch, kv, scheme=sch)
    k = list(r)[0]
    print(k, {x: r[k].get(x) for x in ("ms_per_step", "ms_fastest_step", "ms_slowest_step", "frac_hbm", "ms_per_step_layers_in_one_call", "frac_hbm_layers_in_one_call", "error")}, flush=True)
kv.close()
