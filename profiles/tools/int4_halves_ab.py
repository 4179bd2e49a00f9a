"""INT4 batch attention: two-halves workgroups (16 waves, one per CU) against one-run workgroups (8 waves, two per CU).
python profiles/tools/int4_halves_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for n_seq, T in ((512, 1024), (384, 1024), (1024, 1024), (512, 2048), (256, 1024), (256, 2048), (128, 2048)):
    for one_half, tps in ((0, 0), (1, 0), (1, 16), (1, 32)):
        os.environ.pop("SPECKV_INT4_W8_ONE_HALF", None); os.environ.pop("SPECKV_ATTEND_TILES_PER_SPLIT", None)
        if one_half: os.environ["SPECKV_INT4_W8_ONE_HALF"] = "1"
        if tps: os.environ["SPECKV_ATTEND_TILES_PER_SPLIT"] = str(tps)
        r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=3)
        v = list(r.values())[0]
        print(n_seq, T, "one_half" if one_half else "halves", "tps", tps or "rule", v.get("ms_per_layer"), v.get("frac_hbm"), v.get("error", ""), flush=True)
