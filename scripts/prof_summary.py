"""Summarise a rocprofv3 rocpd database (kernel-trace) into a short per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main(db: str, top: int = 25) -> None:
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x/workgroup_x) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B | wgs |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        print(f"| `{short(r[0])}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
              f"{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
