"""Rewrite the C entry point count quoted in DESIGN.md / INTEGRATION.md / README.md from include/rectools_hip.h (tests/test_abi.py checks them)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(ROOT, "include", "rectools_hip.h")).read()
n = len(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S))))   # as tests/test_abi.py counts
for name, pat, fmt in (("DESIGN.md", r"\((\d+) `extern \"C\"` entry points", '({} `extern "C"` entry points'),
                       ("INTEGRATION.md", r"\((\d+) `extern \"C\"` functions", '({} `extern "C"` functions'),
                       ("README.md", r"\((\d+) C entry points\)", "({} C entry points)")):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    open(p, "w").write(re.sub(pat, fmt.format(n), s))
print(n)
