"""Decode of INT8_DELTA_RLE blocks of structured data through the raw operator: noise, zeros, runs of 32, long runs (200..900),
runs of 8, a third-each mix; plain and with SPECKV_CODEC_HINT_STRUCTURED.  Write-bound launches: frac = (records + 4 + 4096 B) / t.
    python profiles/tools/structured_decode.py [dataset] [hint 0|1] [reps]      (one dataset + hint: the form for --pmc runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import cxl_speckv_amd as pkg
lib = pkg.SpeckvLib(pkg.library_path(), "hip:0")
raw = lib.lib
n_blocks, E, PAGE = 131072, 2048, 4096
only = sys.argv[1] if len(sys.argv) > 1 else None
only_hint = int(sys.argv[2]) if len(sys.argv) > 2 else None
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = torch.Generator(device="cuda"); g.manual_seed(77)
def runs_of(k): return torch.randn((n_blocks, E // k), generator=g, device="cuda").repeat_interleave(k, dim=1).to(torch.float16)
def long_runs(lo, hi):
    m = n_blocks * E // lo + 1
    x = torch.repeat_interleave(torch.randn(m, generator=g, device="cuda"), torch.randint(lo, hi, (m,), generator=g, device="cuda"))
    return x[:n_blocks * E].to(torch.float16).reshape(n_blocks, E).contiguous()
sets = {"noise": lambda: torch.randn((n_blocks, E), generator=g, device="cuda").to(torch.float16),
        "zeros": lambda: torch.zeros((n_blocks, E), dtype=torch.float16, device="cuda"),
        "runs32": lambda: runs_of(32), "runs8": lambda: runs_of(8), "runs128": lambda: runs_of(128),
        "long_200_900": lambda: long_runs(200, 900), "odd_20_60": lambda: long_runs(20, 60)}
recs = torch.empty((n_blocks, PAGE), dtype=torch.uint8, device="cuda")
lens = torch.empty(n_blocks, dtype=torch.int32, device="cuda")
scales = torch.empty(n_blocks, dtype=torch.float32, device="cuda")
dst = torch.empty((n_blocks, E), dtype=torch.float16, device="cuda")
s = torch.cuda.Stream(); sp = s.cuda_stream
with torch.cuda.stream(s):
    for name, make in sets.items():
        if only and name != only: continue
        data = make()
        raw.speckv_ext_codec_compress(data.data_ptr(), n_blocks, recs.data_ptr(), PAGE, lens.data_ptr(), scales.data_ptr(), 2, 0, sp)
        torch.cuda.synchronize()
        comp = int(lens.to(torch.int64).sum().item())
        nbytes = comp + n_blocks * (4 + PAGE)
        out = [name, f"rec {comp / n_blocks:7.1f} B"]
        for hint in (0, 1):
            if only_hint is not None and hint != only_hint: continue
            dec = lambda: raw.speckv_ext_codec_decompress(recs.data_ptr(), PAGE, lens.data_ptr(), scales.data_ptr(), n_blocks, dst.data_ptr(), 0, 2, 0x100 * hint, sp)
            for _ in range(600): dec()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); [dec() for _ in range(reps)]; b.record(s); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            out.append(f"hint {hint}: {ms * 1e3:7.1f} us  {nbytes / (ms * 1e-3) / 8e12:.3f}")
        print("  ".join(out), flush=True)
lib.finalize()
