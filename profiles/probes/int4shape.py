import ctypes, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from importlib import import_module
b = import_module("cxl-speckv_amd.build")
b._preload_torch_hip_runtime()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libint4shape.so"))
T2, L = 16384, 80                        # records per region and layer (32k positions), layers
out = torch.zeros(4, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
V = ctypes.c_void_p
def timeit(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    c.record(); torch.cuda.synchronize()
    return a.elapsed_time(c) / reps

def shipped_table(stride):
    """k_attend_int4_wg's own assignment: wave w fetches rows 8w..8w+7 of K and of V (two instructions of 4 rows, lane l -> row l/16, piece (l & 15) ^ row)
    and the scale lines of 8 pages of K (waves 0, 1) or V (waves 2, 3), pieces xor-ed by the page for K."""
    region = T2 * stride
    tab = np.zeros((2, 4, 5, 64), dtype=np.uint64)
    for g in range(2):
        for w in range(4):
            for l in range(64):
                r0 = 8 * w + (l >> 4); r1 = r0 + 4
                def row_in(r): return 128 + ((r & 1) * 8 + 4 * g) * 64 + (((l & 15) ^ (r & 15)) * 16)
                gr0 = (r0 >> 1) * stride + row_in(r0); gr1 = (r1 >> 1) * stride + row_in(r1)
                spage = 8 * (w & 1) + (l >> 3)
                sin = ((l & 7) ^ ((spage & 7) if w < 2 else 0)) * 16
                tab[g, w, 0, l] = gr0; tab[g, w, 1, l] = gr1; tab[g, w, 2, l] = region + gr0; tab[g, w, 3, l] = region + gr1
                tab[g, w, 4, l] = (0 if w < 2 else region) + spage * stride + sin
    return tab, 5

def table_for(stride, groups, chunk_lists):
    if chunk_lists == "shipped": return shipped_table(stride)
    """chunk_lists[g] = list of byte offsets (16-byte chunks) one workgroup of group g reads per tile, K region relative to the tile's first K record;
    V chunks are the same + region distance.  Dealt to the 4 waves in runs of 64 (one wave instruction = 64 consecutive list entries)."""
    region = T2 * stride
    tabs = []
    ni = None
    for g in range(groups):
        lst = np.array(chunk_lists[g], dtype=np.uint64)
        both = np.concatenate([lst, lst + np.uint64(region)])
        n_instr = (len(both) + 63) // 64
        n_instr = (n_instr + 3) // 4 * 4
        pad = np.full(n_instr * 64 - len(both), both[0], dtype=np.uint64)           # padding re-reads the first chunk (cache hit)
        both = np.concatenate([both, pad]).reshape(n_instr, 64)
        ni = n_instr // 4
        per_wave = np.stack([both[w::4] for w in range(4)])                           # [wave][ni][lane]
        tabs.append(per_wave)
    return np.stack(tabs), ni

def pieces(stride, offs, piece=128, recs=16):
    return [r * stride + o + c for r in range(recs) for o in offs for c in range(0, piece, 16)]

def kernel_order(stride, half):            # as k_attend_int4_wg issues them: 32 rows x 256 B, then the 16 scale lines
    rows = [r * stride + 128 + slot * 512 + half * 256 + c for r in range(16) for slot in range(2) for c in range(0, 256, 16)]
    scales = [r * stride + c for r in range(16) for c in range(0, 128, 16)]
    return rows + scales

def kernel_swizzled(stride, half):         # the shipped kernel: lane l of a row instruction fetches piece (l & 15) ^ row; wave w: rows 8w..8w+7 of K, of V, one scale instruction
    def row_instr(region_off, r_first):
        return [region_off + (r >> 1) * stride + 128 + (r & 1) * 512 + half * 256 + (((l & 15) ^ (r & 15)) * 16) for l in range(64) for r in [r_first + (l >> 4)]]
    return row_instr
cases = {
    "shipped kernel's exact assignment and swizzle": (1152, 2, "shipped", 1152),
    "int4 kernel shape (2 x 256 B per half record + scale line)": (1152, 2, [kernel_order(1152, 0), kernel_order(1152, 1)], 1152),
    "same without the scale line": (1152, 2, [pieces(1152, [128, 256, 640, 768]), pieces(1152, [384, 512, 896, 1024])], 1024),
    "half scale lines (64 B each)": (1152, 2, [pieces(1152, [128, 256, 640, 768]) + [r * 1152 + c for r in range(16) for c in range(0, 64, 16)],
                                            pieces(1152, [384, 512, 896, 1024]) + [r * 1152 + 64 + c for r in range(16) for c in range(0, 64, 16)]], 1152),
    "whole records, 8 heads per workgroup": (1152, 1, [[c for c in range(0, 16 * 1152, 16)]], 1152),
    "stride 1024, halves": (1024, 2, [pieces(1024, [0, 128, 512, 640]), pieces(1024, [256, 384, 768, 896])], 1024),
    "stride 1024, whole records": (1024, 1, [[c for c in range(0, 16 * 1024, 16)]], 1024),
}
for name, (stride, groups, lists, counted) in cases.items():
    tab, ni = table_for(stride, groups, lists)
    d_tab = torch.from_numpy(tab.astype(np.int64).reshape(-1)).cuda()
    nbytes = T2 * 2 * L * stride
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda"); buf.random_(0, 255)
    useful = T2 * 2 * L * counted
    for splits, depth, lds in ((8, 1, 0), (8, 2, 40960), (8, 3, 40960), (8, 3, 32768), (8, 3, 0), (16, 3, 40960)):
        if True:
            rc = []
            def go():
                rc.append(lib.probe_shape(V(buf.data_ptr()), V(d_tab.data_ptr()), ni, depth, groups, ctypes.c_uint64(T2 * 2 * stride), 16 * stride, T2 // 16, L, splits, lds, V(out.data_ptr()), V(st)))
            ms = timeit(go)
            print(f"{name}: ni={ni} splits={splits} wgs={splits * L * groups} depth={depth} lds={lds}: {ms:.3f} ms  record bytes {useful/ms/1e9:.2f} TB/s = {useful/ms/1e9/8:.3f}  rc={rc[0]}")
    del buf
