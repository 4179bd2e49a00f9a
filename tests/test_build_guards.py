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
    ("kernels.o", r"k_fetch_decompress_flat", 0),                           # the kernel of launches hinted "structured" (fp16 and fp32 outputs)
    ("kernels.o", r"k_compressILi[0134]ELi\dE", 0),
    ("kernels.o", r"k_compressILi2ELi\dE", 1),                              # (one tail store of the RLE record)
    ("attend.o", r"k_attend_fp8_linearILb0E", 0),                           # linear, table and residue-class forms
    ("attend.o", r"k_attend_fp8_linearILb1E", 0),                           # striped, address per page (the run-base table through the global address space since round 6)
    ("attend.o", r"k_attend_fp8ENS", 0),
    ("attend.o", r"k_attend_fp8_dma", 0),
    ("attend.o", r"k_qk_scores_fp8_linear", 0),
    ("attend_int4.o", r"k_attend_int4_wgILb0E", 0),
    ("attend_int4.o", r"k_attend_int4_wgILb1E", 1),
    ("attend_int4.o", r"k_attend_int4_wg8ILi1E", 0),
    ("attend_int4.o", r"k_attend_int4_wg8ILi2E", 0),
    ("tensor_codec.o", r"k_td_fusedILi\dELb[01]E", 0),                    # the one-pass tensor kernels (their out-of-line fall-backs are not listed)
    ("tensor_codec.o", r"k_tc_fusedILi\dELb[01]E", 0),
    ("tensor_codec.o", r"k_td_expand_scatter", 0),
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


def _function_text(obj, pattern):
    tmp = tempfile.mkdtemp(prefix="speckv_objx_")
    try:
        shutil.copy(os.path.join(OBJ, obj), tmp)
        subprocess.run([OBJDUMP, "--offloading", obj], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f and "gfx950" in f]
        text = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp, check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1) if re.search(pattern, m.group(1)) else None
            if cur:
                out[cur] = []
        elif cur:
            ins = line.split("//")[0].strip()
            if ins:
                out[cur].append(ins)
    return out


def _vregs(s):
    regs = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", s):
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", s):
        regs.add(int(m.group(1)))
    return regs


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_table_form_int4_kernel_reads_its_page_table_entries_only_behind_their_wait():
    """k_attend_int4_wg<false, true> fetches page-table entries with hand-counted inline-assembly loads; to the compiler those
    loads are complete the moment they are issued, so nothing but the operand list of the following s_waitcnt keeps it from
    scheduling a use in front of the wait (the first version of the kernel computed addresses from entries that had not
    arrived).  In the compiled kernel no instruction may read a register of an entry load before a vmcnt wait that covers it
    (vmcnt(5) or less: everything older than the five newest DMAs)."""
    funcs = _function_text("attend_int4.o", r"k_attend_int4_wgILb0ELb1E")
    assert len(funcs) == 1
    (name, lines), = funcs.items()
    pending, loads = [], 0
    for ins in lines:
        m = re.match(r"global_load_dwordx4 (v\[\d+:\d+\])", ins)
        if m and "lds" not in ins:
            pending.append(_vregs(m.group(1)))
            loads += 1
            continue
        w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
        if w:
            if int(w.group(1)) <= 5:
                pending = []
            continue
        parts = ins.split(None, 1)
        if len(parts) < 2 or not pending:
            continue
        ops = parts[1].split(",")
        srcs = parts[1] if parts[0].startswith(("global_store", "ds_write", "global_load_lds")) else ",".join(ops[1:])
        bad = _vregs(srcs) & set().union(*pending)
        assert not bad, f"{name}: `{ins}` reads v{sorted(bad)} while its entry load is in flight"
    assert loads >= 20, loads                              # the prologue's and the loop's entry loads were found


def test_no_environment_walks_behind_the_entry_points():
    """the library reads its environment at open / first use only.  getenv may appear in csrc/tuning.cpp (the
    one-time table), csrc/engine_internal.hpp (helpers of Engine::open and the log switch, a static initialiser) and in
    Engine::open / Engine::init_hip of csrc/engine.cpp -- nowhere behind speckv_ext_attend_*, fetch_range or the codec
    operators: their translation units contain no getenv at all."""
    csrc = os.path.join(ROOT, "cxl-speckv_amd", "csrc")
    allowed = {"tuning.cpp", "engine_internal.hpp", "engine.cpp", "tuning.hpp"}
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".cpp", ".hip", ".hpp")) or fn in allowed:
            continue
        text = open(os.path.join(csrc, fn)).read()
        assert not re.search(r"\bgetenv\s*\(", text), f"{fn} reads the environment"
    # engine.cpp: only inside open() / init_hip() (everything in front of the first function that is neither)
    text = open(os.path.join(csrc, "engine.cpp")).read()
    end_of_open = text.index("int Engine::wait_event")
    assert "getenv" not in text[end_of_open:], "engine.cpp reads the environment outside Engine::open / Engine::init_hip"
    # the header documents the entry point tests use instead
    assert "speckv_ext_set_tuning" in open(os.path.join(ROOT, "include", "speckv_ext.h")).read()


HEADLINE = r"k_fetch_decompressILi2ELi0ELb0ELi0E"                     # k_fetch_decompress<INT8_DELTA_RLE, REF_EXACT, fp16 out, EXT 0>: bench.py's dominant kernel


def headline_kernel_hash():
    """(mangled name, SHA-256 of its instructions one per line without addresses / encodings, instruction count) of the headline
    kernel as built.  profiles/publish_r06.py writes it into <tag>_pmc.json beside the PMC traffic measured on that build."""
    import hashlib
    funcs = _function_text("kernels.o", HEADLINE)
    assert len(funcs) == 1, list(funcs)
    (name, lines), = funcs.items()
    return name, hashlib.sha256("\n".join(lines).encode()).hexdigest(), len(lines)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_committed_pmc_traffic_belongs_to_the_headline_kernel_as_built():
    """bench.py's roofline.traffic is a constant read from the newest profiles/r*_pmc.json -- PMC cannot be
    read from inside the run.  That file names the kernel it was measured on and the hash of its instructions; a build whose
    headline kernel differs (any change to k_fetch_decompress<2, 0, false, 0>) fails here until the PMC passes are collected
    again (profiles/tools/r6_final.sh + profiles/publish_r06.py), so the figure cannot go stale silently."""
    import glob
    import json
    import __graft_entry__ as entry
    if not os.path.exists(os.path.join(OBJ, "kernels.o")):
        entry.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import bench
    traffic, src = bench.pmc_traffic(2, 0)
    assert traffic and src, "no committed PMC file with the headline kernel's traffic"
    d = json.load(open(os.path.join(root, src)))
    name, sha, n = headline_kernel_hash()
    assert d.get("kernel") == name, (src, d.get("kernel"), name)
    assert d.get("kernel_instructions_sha256") == sha, f"{src} was measured on another build of {name} ({d.get('kernel_instructions')} instructions then, {n} now): collect the PMC passes again"
