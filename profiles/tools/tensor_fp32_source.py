"""The tensor compressor on fp16 and fp32 sources of the same values, noise and long runs (SPECKV_TC_NO_SPLIT_TILES=1: without the SPLIT tile form): python profiles/tools/tensor_fp32_source.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0"); raw = kv.lib.lib
n = 128 * 2**20
g = torch.Generator(device="cuda"); g.manual_seed(2001)
def long_runs():
    m = n // 200 + 1
    x = torch.repeat_interleave(torch.randn(m, generator=g, device="cuda"), torch.randint(200, 900, (m,), generator=g, device="cuda"))
    return x[:n].to(torch.float16).contiguous()
for name, make in (("noise", lambda: torch.randn(n, generator=g, device="cuda", dtype=torch.float32).to(torch.float16)), ("long_runs", long_runs)):
    x16 = make(); x32 = x16.to(torch.float32)
    ws_bytes = int(raw.speckv_ext_codec_tensor_workspace_bytes(n)); ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda"); wsp = (ws.data_ptr() + 255) & ~255
    rle = torch.empty(2 * n + 32, dtype=torch.uint8, device="cuda"); meta = torch.zeros(4, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream(); sizes = {}
    for f32, x in ((0, x16), (1, x32)):
        enc = lambda: raw.speckv_ext_codec_compress_tensor(x.data_ptr(), n, f32, rle.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, wsp, ws_bytes, 0, s.cuda_stream)
        for _ in range(10): assert enc() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); [enc() for _ in range(5)]; b.record(s); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        size = int(meta[0].item()); sizes[f32] = (size, int(rle[:size].to(torch.int64).sum().item()))
        print(f"tensor compress {name} src_f32={f32}: {ms:.4f} ms  stream {size}", flush=True)
    assert sizes[0] == sizes[1], sizes
    del x16, x32, ws, rle
