"""FP8 batch attention (one layer) over a few (sequences, context) shapes.  SPECKV_FP8_BATCH_KERNEL=dma|reg forces a kernel,
SPECKV_ATTEND_TILES_PER_SPLIT the split length.  python profiles/tools/fp8_batch_shapes.py [scheme]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
scheme = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
for n_seq, T in ((256, 2048), (256, 4096), (256, 8192), (64, 8192), (32, 32768), (512, 1024), (128, 2048)):
    r = bench.batch_attention_extra(torch, kv, n_seq=n_seq, T=T, scheme=scheme)
    v = list(r.values())[0]
    print(n_seq, T, v.get("ms_per_layer"), v.get("frac_hbm"), v.get("error", ""))
