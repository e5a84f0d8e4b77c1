"""The ctypes mirrors of the header's descriptor structs (`_lib.SasrecBlock / PreLNBlock / SasrecStep`) have the C compiler's layout:
sizeof and the offset of every field, from a gcc-built probe that includes include/rectools_hip.h (plain C)."""
import ctypes
import os
import shutil
import subprocess

import pytest

from rectools_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = {"rt_sasrec_block": _lib.SasrecBlock, "rt_preln_block": _lib.PreLNBlock, "rt_sasrec_step": _lib.SasrecStep}


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_ctypes_structs_match_the_header(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/rectools_hip.h"', "int main(void) {"]
    for cname, cls in STRUCTS.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, field, value = line.split()
        cls = STRUCTS[cname]
        if field == "sizeof":
            assert ctypes.sizeof(cls) == int(value), f"sizeof({cname}): C {value}, ctypes {ctypes.sizeof(cls)}"
        else:
            assert getattr(cls, field).offset == int(value), f"{cname}.{field}: C offset {value}, ctypes {getattr(cls, field).offset}"
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in STRUCTS.values())
