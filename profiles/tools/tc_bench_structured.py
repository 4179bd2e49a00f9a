"""The whole-tensor codec on data that COMPRESSES (piecewise constant: runs of 16 .. 4000 equal values; a third noise): the chunks of
the decoder then span many 4096-element windows, the compressor's tiles sit inside long stretches.
python profiles/tools/tc_bench_structured.py [n]      (SPECKV_TC_MULTIPASS=1 for the multi-launch forms)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cxl_speckv_amd as pkg
kv = pkg.CxlSpeckvKVAllocator(pkg.library_path(), "hip:0")
raw = kv.lib.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 2**20
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 16
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
noise_third = len(sys.argv) <= 4 or sys.argv[4] != "0"
g = torch.Generator(device="cuda"); g.manual_seed(7)
m = n // max(1, (lo + hi) // 4) + 1
lens = torch.randint(lo, hi, (m,), generator=g, device="cuda")
vals = torch.randn(m, generator=g, device="cuda")
x = torch.repeat_interleave(vals, lens)[:n]
if x.numel() < n: x = torch.cat([x, torch.zeros(n - x.numel(), device="cuda")])
if noise_third:
    noise = torch.randn(n // 3, generator=g, device="cuda")
    x[n // 3: n // 3 + noise.numel()] = noise
x = x.to(torch.float16).contiguous()
ws_bytes = int(raw.speckv_ext_codec_tensor_workspace_bytes(n))
ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device="cuda"); wsp = (ws.data_ptr() + 255) & ~255
rle = torch.empty(2 * n + 32, dtype=torch.uint8, device="cuda")
meta = torch.zeros(4, dtype=torch.int64, device="cuda")
y = torch.empty(n, dtype=torch.float16, device="cuda")
s = torch.cuda.Stream()
def enc(): assert raw.speckv_ext_codec_compress_tensor(x.data_ptr(), n, 0, rle.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, wsp, ws_bytes, 0, s.cuda_stream) == 0
enc(); torch.cuda.synchronize()
size = int(meta[0].item()); scale = float(meta[1:2].view(torch.float32)[0].item())
dws_bytes = int(raw.speckv_ext_codec_tensor_decode_workspace_bytes(size))
dws = torch.empty(dws_bytes + 256, dtype=torch.uint8, device="cuda"); dwsp = (dws.data_ptr() + 255) & ~255
def dec(): assert raw.speckv_ext_codec_decompress_tensor(rle.data_ptr(), size, scale, y.data_ptr(), n, 0, meta.data_ptr() + 16, dwsp, dws_bytes, 0, s.cuda_stream) == 0
dec(); torch.cuda.synchronize()
ok = bool((y.float() - x.float()).abs().max() <= scale / 127 * 0.51 + 1e-3)
for name, fn, byt in (("compress", enc, 2 * n + size), ("decompress", dec, size + 2 * n)):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(5): fn()
    b.record(s); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"structured runs {lo}..{hi} n={n} stream={size} ({2*n/size:.2f}x) {name} {ms:.4f} ms  {byt/ms/1e6:.0f} GB/s  frac {byt/ms/1e6/8000:.3f}  form={'multipass' if os.environ.get('SPECKV_TC_MULTIPASS') else 'one pass'}  roundtrip_ok={ok}", flush=True)
