"""FP8 batch attention: the register-staged kernel (k_attend_fp8_linear) against the LDS-DMA kernel (k_attend_fp8_dma), by
SPECKV_FP8_BATCH_KERNEL=reg|dma python profiles/tools/fp8_batch_kernel_ab.py  (one process per choice)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
k = os.environ.get("SPECKV_FP8_BATCH_KERNEL", "default (reg)")          # (read once by the library: one process per choice)
for n_seq, T in ((256, 1024), (256, 2048), (256, 8192), (64, 8192), (512, 1024)):
    r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=4)
    v = list(r.values())[0]
    print(k, n_seq, T, v.get("ms_per_layer"), v.get("frac_hbm"), v.get("error", ""), flush=True)
