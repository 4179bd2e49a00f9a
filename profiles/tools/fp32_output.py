"""Every scheme of the block codec decoded to fp16 and to fp32 (out_f32), 131 072 N(0,1) blocks: python profiles/tools/fp32_output.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cxl_speckv_amd as pkg
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0"); raw = lib.lib
n, E, PAGE = 131072, 2048, 4096
g = torch.Generator(device="cuda"); g.manual_seed(5)
src = torch.randn((n, E), generator=g, device="cuda").to(torch.float16)
recs = torch.empty((n, PAGE), dtype=torch.uint8, device="cuda"); lens = torch.empty(n, dtype=torch.int32, device="cuda"); scales = torch.empty(n, dtype=torch.float32, device="cuda")
dst32 = torch.empty((n, E), dtype=torch.float32, device="cuda"); dst16 = torch.empty((n, E), dtype=torch.float16, device="cuda")
s = torch.cuda.Stream(); sp = s.cuda_stream
with torch.cuda.stream(s):
    for scheme in (2, 1, 4, 3, 0):
        stride = {1: 2048, 3: 1152, 4: 2048}.get(scheme, PAGE)
        raw.speckv_ext_codec_compress(src.data_ptr(), n, recs.data_ptr(), stride, lens.data_ptr(), scales.data_ptr(), scheme, 0, sp); torch.cuda.synchronize()
        comp = int(lens.to(torch.int64).sum().item())
        for f32, dst in ((0, dst16), (1, dst32)):
            dec = lambda: raw.speckv_ext_codec_decompress(recs.data_ptr(), stride, lens.data_ptr(), scales.data_ptr(), n, dst.data_ptr(), f32, scheme, 0, sp)
            for _ in range(300): dec()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); [dec() for _ in range(30)]; b.record(s); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 30
            byt = comp + n * (4 + E * (4 if f32 else 2))
            print(f"scheme {scheme} f32={f32}: {ms*1e3:.1f} us  {byt/ms/1e6/8000:.3f}", flush=True)
lib.finalize()
