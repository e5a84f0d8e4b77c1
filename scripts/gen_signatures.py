"""Regenerate the `SIGNATURES` table of rectools_amd/_lib.py from include/rectools_hip.h (run after editing the header)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {"int32_t": "c_i32", "int64_t": "c_i64", "float": "c_f32", "double": "c_f64", "uint64_t": "c_u64", "size_t": "c_sz",
         "rt_stream_t": "c_vp", "int": "c_i32", "void": "None", "uint32_t": "c_u32"}


def ctype(decl: str) -> str:
    decl = decl.strip()
    if "*" in decl:
        return "c_vp"
    base = decl.replace("const", "").split()[0]
    return TYPES[base]


def main() -> None:
    src = open(os.path.join(ROOT, "include", "rectools_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    rows = []
    for ret, name, args in re.findall(r"\b([a-z_0-9]+(?:\s+[a-z]+)?\s*\*?)\s*\b(rt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret = ret.strip()
        restype = "ctypes.c_char_p" if "char" in ret else ctype(ret)
        a = [ctype(x) for x in args.split(",") if x.strip() and x.strip() != "void"]
        rows.append(f'    "{name}": ({restype}, [{", ".join(a)}]),')
    path = os.path.join(ROOT, "rectools_amd", "_lib.py")
    text = open(path).read()
    start = text.index("SIGNATURES: ")
    brace = text.index("{", start)
    end = text.index("\n}\n", brace)
    text = text[: brace + 1] + "\n" + "\n".join(rows) + text[end:]
    open(path, "w").write(text)
    print(f"{len(rows)} entry points")


if __name__ == "__main__":
    main()
