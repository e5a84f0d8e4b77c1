"""Print the headline numbers of a bench.py JSON line (used by scripts/gpu/visit.sh)."""
import json
import sys

try:
    j = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e); sys.exit(0)
print("value", j.get("value"), j.get("unit"), "ms/step", j.get("ms_per_step"), "loss", j.get("final_loss"), "env", j.get("env"))
for k in ("train_exact_gemm", "recommend", "topk5m", "topk5m_u4096"):
    if isinstance(j.get(k), dict):
        r = j[k].get("roofline") or {}
        cb = j[k].get("cpu_baseline") or {}
        print(" ", k, j[k].get("value"), j[k].get("unit"), "frac", r.get("frac"), r.get("achieved"), r.get("unit"),
              {a: b for a, b in j[k].items() if a.startswith("phase")}, "kernel", (j[k].get("ranker_kernel") or {}).get("value"),
              "cpu", cb.get("kind"), cb.get("value"))
for k, v in (j.get("families") or {}).items():
    print("  family", k, v.get("value"), v.get("unit"), "ms/step", v.get("ms_per_step"), "roofline", (v.get("roofline") or {}).get("kernel", "")[:30], (v.get("roofline") or {}).get("frac"))
cb = j.get("cpu_baseline") or {}
print("  cpu_baseline", cb.get("kind"), cb.get("value"), cb.get("unit"), "cores", cb.get("cores"))
r = j.get("roofline") or {}
print("  roofline", r.get("kernel", "")[:50], r.get("achieved"), r.get("unit"), "frac", r.get("frac"), "host_issue_ms", r.get("host_issue_ms_per_step"),
      "host_only", r.get("host_only"), "step_TF", r.get("step_TFLOPs"), "gemm launches/step", r.get("launches_per_step"))
for k, v in list((j.get("kernel_breakdown") or {}).items())[:24]:
    print("   %-30s %8.4f ms  x%5.1f  single %8.4f" % (k, v["ms_per_step"], v["calls_per_step"], v["single_stream_ms"]))
