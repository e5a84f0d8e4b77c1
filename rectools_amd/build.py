"""Build recipe for `librectools_hip.so` (gfx950 only): explicit `hipcc -shared -fPIC`, in-tree output.

`hipcc` cross-compiles without a GPU, so this runs in the build container; the resulting `.so` is
git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librectools_hip.so")
OBJ_DIR = os.path.join(CSRC, "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
FLAGS += os.environ.get("RT_EXTRA_HIPCC_FLAGS", "").split()   # diagnostic builds (e.g. -DRT_ATTN_TRACE); part of the digests


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest(path: str) -> str:
    h = hashlib.sha1()
    for p in [path] + sorted(glob.glob(os.path.join(CSRC, "*.h"))):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _lib_digest() -> str:
    h = hashlib.sha1()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip to an object (cached by content hash) and link the shared library."""
    lib_stamp = LIB_PATH + ".sha1"
    lib_dig = _lib_digest()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(lib_stamp) and open(lib_stamp).read() == lib_dig:
        return LIB_PATH  # prebuilt library matches the sources (the case on the GPU box)
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    relink = True   # we only get here when the library is missing or its digest differs (e.g. a source was removed)
    procs = []
    for src in _sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OBJ_DIR, base + ".o")
        stamp = obj + ".sha1"
        dig = _digest(src)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), src, stamp, dig))
        relink = True
    for p, src, stamp, dig in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed for {src}")
        with open(stamp, "w") as f:
            f.write(dig)
    if relink:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(lib_stamp, "w") as f:
        f.write(lib_dig)
    return LIB_PATH


def build_ablation(verbose: bool = False) -> str:
    """The diagnostic twin `librectools_hip_ablation.so` (git-ignored, loaded through `RT_LIB_PATH`): the sources that carry
    `RT_ABLATION_BUILD` switches compiled with `-DRT_ABLATION_BUILD`, every other object shared with the product build."""
    build(verbose=verbose)
    out = os.path.join(PKG_DIR, "librectools_hip_ablation.so")
    abl_dir = os.path.join(OBJ_DIR, "abl")
    os.makedirs(abl_dir, exist_ok=True)
    objs, procs = [], []
    for src in _sources():
        base = os.path.splitext(os.path.basename(src))[0]
        with open(src) as f:
            switched = "RT_ABLATION_BUILD" in f.read()
        if not switched:
            objs.append(os.path.join(OBJ_DIR, base + ".o"))
            continue
        obj = os.path.join(abl_dir, base + ".o")
        stamp, dig = obj + ".sha1", _digest(src)
        objs.append(obj)
        if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        cmd = [HIPCC] + FLAGS + ["-DRT_ABLATION_BUILD", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), src, stamp, dig))
    for p, src, stamp, dig in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(o.decode(errors="replace"))
            raise RuntimeError(f"hipcc failed for {src}")
        with open(stamp, "w") as f:
            f.write(dig)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    if "--ablation" in sys.argv:
        print(build_ablation(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
