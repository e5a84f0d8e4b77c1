#!/usr/bin/env python
"""TEST INFRASTRUCTURE ONLY.  Stage the UNMODIFIED reference package for the GPU box.

`/root/reference` exists only in the build container; bench.py's `cpu_baseline` leg wants to time the reference ITSELF on the GPU
box's host cores ("the unmodified reference through the shims on the same box").  This recipe copies the reference's
pure-Python package (`/root/reference/rectools`, ~1 MB of .py files) into the git-ignored `oracle/_ref/` — never into history, never
into the product package — so that it travels with the gpurun snapshot like the built `.so` does.  `oracle/ref_shims.py` then finds it
there when `/root/reference` is absent.  Nothing under `rectools_amd/` imports it; `__graft_entry__.build()` runs this when the
reference tree is present.

    python oracle/make_ref.py            # -> oracle/_ref/rectools + oracle/_ref/SOURCE.txt
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("RECTOOLS_REFERENCE_SOURCE", "/root/reference")
DST = os.path.join(HERE, "_ref")


def stage(verbose: bool = True) -> bool:
    pkg = os.path.join(SRC, "rectools")
    if not os.path.isdir(pkg):
        if verbose:
            print(f"[make_ref] {pkg} not found: nothing staged (the GPU box uses what the build container staged)")
        return False
    if os.path.isdir(os.path.join(DST, "rectools")):
        shutil.rmtree(os.path.join(DST, "rectools"))
    os.makedirs(DST, exist_ok=True)
    shutil.copytree(pkg, os.path.join(DST, "rectools"), ignore=shutil.ignore_patterns("__pycache__", "*.pyc", "*.so", "*.c", "*.pyx"))
    version = "unknown"
    try:
        for line in open(os.path.join(pkg, "version.py")):
            if "VERSION" in line and "=" in line:
                version = line.split("=", 1)[1].strip().strip("\"'")
    except OSError:
        pass
    with open(os.path.join(DST, "SOURCE.txt"), "w") as f:
        f.write(f"unmodified copy of {pkg} (RecTools {version}), staged by oracle/make_ref.py for bench.py's cpu_baseline leg; git-ignored\n")
    if verbose:
        n = sum(len(fs) for _, _, fs in os.walk(os.path.join(DST, "rectools")))
        print(f"[make_ref] staged {n} files of RecTools {version} under {DST}")
    return True


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
