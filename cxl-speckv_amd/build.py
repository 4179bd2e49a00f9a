"""Build driver for libcxlspeckv.so (hipcc, --offload-arch=gfx950, in-tree)."""
import ctypes
import importlib.util
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "lib", "libcxlspeckv.so")


def build_library(force=False, jobs=4):
    """Compile every HIP source for gfx950 into cxl-speckv_amd/lib/libcxlspeckv.so."""
    if force:
        subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-s", "-C", CSRC, f"-j{jobs}", "ARCH=gfx950"])
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build produced no libcxlspeckv.so")
    return LIB_PATH


def library_path():
    """Path of the built library; raises (loudly) when it has not been built.
    SPECKV_LIB_PATH overrides it (A/B builds during kernel tuning)."""
    override = os.environ.get("SPECKV_LIB_PATH")
    if override:
        if not os.path.exists(override):
            raise RuntimeError(f"SPECKV_LIB_PATH={override} does not exist")
        return override
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: build in-tree once (hipcc cross-compiles gfx950 anywhere)
        try:
            build_library()
        except Exception as e:
            raise RuntimeError(
                f"{LIB_PATH} is missing and could not be built ({e!r}). Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU fallback for the KV data path.") from e
    return LIB_PATH


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP
    runtimes in one process cannot both own the GPU, so when torch is installed but
    not imported yet, map ITS runtime first: libcxlspeckv.so (NEEDED libamdhip64.so.7)
    then binds to that copy by SONAME, and a later ``import torch`` reuses it."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def load_library(path=None):
    """ctypes.CDLL of libcxlspeckv.so with a single HIP runtime in the process."""
    _preload_torch_hip_runtime()
    return ctypes.CDLL(path or library_path())
