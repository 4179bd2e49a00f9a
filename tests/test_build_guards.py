"""Not -m gpu: properties of the COMPILED gfx950 code that a source review does not show.

Record pointers reach the kernels through page-table entries and pointer lists.  A plain C++ dereference of such a pointer --
or of one that is merely initialised as nullptr and assigned under a template condition -- compiles to a FLAT memory
instruction: it counts in lgkmcnt as well as vmcnt, so every wait in front of an LDS or scalar read also waits for the record
loads in flight, and the compiler's vmcnt(N) bookkeeping collapses to vmcnt(0).  Round 3 lost 15 % of the linear FP8
attention kernel that way without any test noticing.  The hot kernels therefore go through explicit global-address-space
pointers (ldg16 / ld16 / gload ...), and this test disassembles the built objects and checks that they stay free of flat
accesses."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "cxl-speckv_amd", "lib", "obj")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# (object, kernel name pattern, flat instructions allowed)
HOT = [
    ("kernels.o", r"k_fetch_decompressILi\dELi\dELb[01]ELi[01]E", 0),      # plain and extension-1 forms of every scheme (EXT 2 / 3: the flush forms keep two bookkeeping accesses)
    ("kernels.o", r"k_compressILi[0134]ELi\dE", 0),
    ("kernels.o", r"k_compressILi2ELi\dE", 1),                              # (one tail store of the RLE record)
    ("attend.o", r"k_attend_fp8_linearILb0E", 0),
    ("attend.o", r"k_attend_fp8_linearILb1E", 1),                           # (the run-base table is read once per workgroup)
    ("attend.o", r"k_attend_fp8ENS", 0),
    ("attend.o", r"k_attend_fp8_dma", 0),
    ("attend.o", r"k_qk_scores_fp8_linear", 0),
    ("attend_int4.o", r"k_attend_int4_wgILb0E", 0),
    ("attend_int4.o", r"k_attend_int4_wgILb1E", 1),
    ("attend_int4.o", r"k_attend_int4ILb0E", 0),
]


def _disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="speckv_objx_")
    try:
        shutil.copy(os.path.join(OBJ, obj), tmp)
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        assert co, f"no gfx950 code object in {obj}"
        text = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = 0
        elif cur and re.search(r"\bflat_(load|store|atomic)", line):
            funcs[cur] += 1
    return funcs


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_hot_kernels_use_no_flat_memory_instructions():
    import __graft_entry__ as entry
    if not os.path.exists(os.path.join(OBJ, "kernels.o")):
        entry.build()
    cache = {}
    for obj, pattern, allowed in HOT:
        funcs = cache.setdefault(obj, _disassemble(obj))
        hits = {name: n for name, n in funcs.items() if re.search(pattern, name)}
        assert hits, f"no kernel matching {pattern} in {obj}"
        for name, n in hits.items():
            assert n <= allowed, f"{name}: {n} flat memory instructions (allowed {allowed}) -- a record pointer is dereferenced without the global-address-space helpers"
