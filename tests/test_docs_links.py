"""Not -m gpu: every repository path the documents quote in backticks exists (evidence files under profiles/, tests, sources)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "DESIGN_HISTORY.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"))
PREFIXES = ("profiles/", "tests/", "cxl-speckv_amd/", "include/", "oracle/", "examples/")
REFERENCE_PATHS = {"tests/test_c_api.c"}                 # paths of the reference repository that the text names as such


def test_documented_paths_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`([A-Za-z0-9_./*<>\-]+)`", text):
            p = m.group(1)
            if not p.startswith(PREFIXES) or "<" in p or ">" in p or p in REFERENCE_PATHS:
                continue
            q = p.split("::")[0].rstrip(".,")
            if q.endswith("_build/") or "/_build/" in q or q.startswith("oracle/_ref"):
                continue                                   # build outputs (git-ignored)
            hit = glob.glob(os.path.join(ROOT, q)) if "*" in q else os.path.exists(os.path.join(ROOT, q))
            if not hit:
                missing.append((doc, p))
    assert not missing, missing
