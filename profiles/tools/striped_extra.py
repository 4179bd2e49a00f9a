"""bench.striped_attention_extra alone (pool striped x7 on one GPU: computed-address / class forms, table forms, batch): python profiles/tools/striped_extra.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
r = bench.striped_attention_extra(torch, pkg)
for k, v in r["fused_attention_striped_x7"].items():
    print(k, json.dumps(v) if isinstance(v, dict) else v)
