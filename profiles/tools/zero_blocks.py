"""Decode of structured INT8_DELTA_RLE blocks (all zeros; runs of 32) through the raw operator: store-bound launches.
SPECKV_WGS_PER_CU=<n> changes the grid cap (blocks per wave).  python profiles/tools/zero_blocks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
n_blocks = 131072
g = torch.Generator(device="cuda"); g.manual_seed(1)
src = torch.randn((n_blocks, 2048), generator=g, device="cuda").to(torch.float16)
dst = torch.empty_like(src)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ex = bench.run_extras(torch, pkg, lib, src, dst, n_blocks, s.cuda_stream)
for k in ("rle_ref_exact", "rle_all_zero_blocks", "rle_piecewise_runs_of_32", "fp16_copy"):
    print(os.environ.get("SPECKV_WGS_PER_CU", "default"), k, ex[k]["decompress_frac_hbm"], ex[k]["decompress_GBps"])
lib.finalize()
