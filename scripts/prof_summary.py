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


def timeline(db: str, marker: str = "adam_segs_kernel", max_rows: int = 140) -> None:
    """ONE step as the GPU saw it: every dispatch between two consecutive `marker` kernels in the middle of the run, in start order — offset, duration, the idle
    gap since the latest end seen so far (negative: overlap with a kernel of another stream), queue, workgroups, name."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    pick = lambda *names: next((n for n in names if n in cols), None)   # noqa: E731
    st, en, q = pick("start", "start_time", "start_timestamp"), pick("end", "end_time", "end_timestamp"), pick("queue_id", "stream_id", "queue")
    if st is None or en is None:
        print("no start / end columns in `kernels`:", cols)
        return
    marks = c.execute(f"select {st} from kernels where name like ? order by {st}", (f"%{marker}%",)).fetchall()
    if len(marks) < 3:
        print("marker kernel seen fewer than 3 times")
        return
    mid = len(marks) // 2                    # a step from the middle of the run: bench.py's timed region (its last steps carry timing events)
    t0, t1 = marks[mid - 1][0], marks[mid][0]
    rows = c.execute(f"select {st}, {en}, name, grid_x / workgroup_x, {q or '0'}, lds_size from kernels where {st} > ? and {st} <= ? order by {st}",
                     (t0, t1)).fetchall()
    print(f"one step: {len(rows)} dispatches, {(t1 - t0) / 1e3:.1f} us from marker to marker; columns: start us | dur us | gap us | queue | wgs | lds | name")
    last_end = rows[0][0] if rows else 0
    busy = 0.0
    for r in rows[:max_rows]:
        gap = (r[0] - last_end) / 1e3
        busy += (r[1] - r[0]) / 1e3
        print(f"{(r[0] - t0) / 1e3:9.1f} {(r[1] - r[0]) / 1e3:8.1f} {gap:8.1f}  q{r[4]}  {r[3]:6d} {r[5]:7d}  {short(r[2])[:70]}")
        last_end = max(last_end, r[1])
    print(f"sum of durations {busy:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--timeline":
        timeline(sys.argv[1], *(sys.argv[3:4]))
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
