"""The tensor decoder to fp16 and to fp32, noise (one-pass kernel) and long runs (by output): python profiles/tools/tensor_fp32_output.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0"); raw = kv.lib.lib
n = 256 * 2**20
g = torch.Generator(device="cuda"); g.manual_seed(2001)
def long_runs():
    m = n // 200 + 1
    x = torch.repeat_interleave(torch.randn(m, generator=g, device="cuda"), torch.randint(200, 900, (m,), generator=g, device="cuda"))
    return x[:n].to(torch.float16).contiguous()
for name, make in (("noise", lambda: torch.randn(n, generator=g, device="cuda", dtype=torch.float32).to(torch.float16)), ("long_runs", long_runs)):
    x = make()
    ws_bytes = int(raw.speckv_ext_codec_tensor_workspace_bytes(n)); ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda"); wsp = (ws.data_ptr() + 255) & ~255
    rle = torch.empty(2 * n + 32, dtype=torch.uint8, device="cuda"); meta = torch.zeros(4, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    assert raw.speckv_ext_codec_compress_tensor(x.data_ptr(), n, 0, rle.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, wsp, ws_bytes, 0, s.cuda_stream) == 0
    torch.cuda.synchronize()
    size = int(meta[0].item()); scale = float(meta[1:2].view(torch.float32)[0].item())
    del ws
    dws_bytes = int(raw.speckv_ext_codec_tensor_decode_workspace_bytes(size)); dws = torch.empty(dws_bytes + 256, dtype=torch.uint8, device="cuda"); dwsp = (dws.data_ptr() + 255) & ~255
    y16 = torch.empty(n, dtype=torch.float16, device="cuda"); y32 = torch.empty(n, dtype=torch.float32, device="cuda")
    for f32, y in ((0, y16), (1, y32)):
        dec = lambda: raw.speckv_ext_codec_decompress_tensor(rle.data_ptr(), size, scale, y.data_ptr(), n, f32, meta.data_ptr() + 16, dwsp, dws_bytes, 0, s.cuda_stream)
        for _ in range(20): assert dec() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); [dec() for _ in range(10)]; b.record(s); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = size + n * (4 if f32 else 2)
        print(f"tensor {name} f32={f32}: {ms:.4f} ms  {byt/ms/1e6/8000:.3f}  stream {size}", flush=True)
    assert torch.equal(y32.to(torch.float16), y16), "fp32 and fp16 outputs differ"
    del x, rle, y16, y32, dws
