"""raw k_compress<2,REF_EXACT> on 131072 N(0,1) blocks: python enc_bench.py [reps] [scheme] [mode]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from tests._gpu import load_raw_lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scheme = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = load_raw_lib()
n = 131072
g = torch.Generator(device="cuda"); g.manual_seed(2001)
src = torch.randn((n, 2048), generator=g, device="cuda", dtype=torch.float32).to(torch.float16)
stride = {1: 2048, 3: 1152, 4: 2048}.get(scheme, 4096)
recs = torch.empty((n, stride), dtype=torch.uint8, device="cuda")
lens = torch.empty(n, dtype=torch.int32, device="cuda"); scales = torch.empty(n, dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
def enc():
    assert lib.speckv_ext_codec_compress(src.data_ptr(), n, recs.data_ptr(), stride, lens.data_ptr(), scales.data_ptr(), scheme, mode, s.cuda_stream) == 0
for _ in range(reps // 4 + 10): enc()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(s)
for _ in range(reps): enc()
b.record(s); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
comp = int(lens.to(torch.int64).sum().item())
byt = n * 4096 + comp + n * 8
print(f"compress scheme={scheme} mode={mode} lib={os.environ.get('SPECKV_LIB_PATH','default')} us={ms*1e3:.1f} GB/s={byt/ms/1e6:.1f} frac={byt/ms/1e6/8000:.4f} checksum={int(recs[:, :64].to(torch.int64).sum().item())} comp={comp}", flush=True)
