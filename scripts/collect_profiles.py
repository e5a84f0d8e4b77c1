"""Copy the evidence of `scripts/gpu_profile_r2.sh` (gpurun_out/r2/) into profiles/ (tracked) and rebuild profiles/traffic.json
from the PMC passes.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so fetches are
doubled (MI355X_MICROARCH.md, HBM section)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r2")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"

KEEP = [
    ("bench_auto.json", "bench_auto.json"), ("bench_bert4rec.json", "bench_bert4rec.json"), ("bench_hstu.json", "bench_hstu.json"),
    ("bench_esasrec.json", "bench_esasrec.json"), ("bench_auto_2ranks_on_1gpu_gloo.json", "bench_auto_2ranks_on_1gpu_gloo.json"),
] + [(f"rocprof_kernel_trace_{n}.md", f"rocprof_kernel_trace_{n}.md") for n in
     ("train", "train_single_stream", "topk5m", "recommend", "bert4rec", "hstu", "esasrec")] + [
    (f"pmc_{w}_{c}.txt", f"pmc_{w}_{c}.txt") for w in ("topk5m", "train") for c in ("FETCH_SIZE", "WRITE_SIZE")] + [
    (f"sq_{n}.md", f"sq_counters_{n}.md") for n in ("attention", "attention_l512", "topk5m", "recommend")]


def parse_pmc(path):
    """-> {kernel prefix: (calls, avg KiB)}"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"\w+ (.+?)\s+calls=(\d+) avg=([\d.]+) max=([\d.]+) sum=([\d.]+)", line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(5)))
    return out


def main():
    for src, dst in KEEP:
        p = os.path.join(SRC, src)
        if os.path.exists(p):
            shutil.copyfile(p, os.path.join(DST, f"{TAG}_{dst}"))
        else:
            print("missing", src)
    traffic = {"_note": "HBM bytes per launch of the dominant kernel of each bench leg, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                        "passes (kernel-trace only); FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request), "
                        "WRITE_SIZE in KiB.  topk5m: whole rt_topk_score call (seeding prefix + main pass of topk_stream16_kernel + the two "
                        "selection kernels), algorithmic 10.24e9.  train_gemm: average rt_gemm launch (the three gemm_dma_kernel operand "
                        "layouts)."}
    f, w = parse_pmc(os.path.join(SRC, "pmc_topk5m_FETCH_SIZE.txt")), parse_pmc(os.path.join(SRC, "pmc_topk5m_WRITE_SIZE.txt"))
    if f:
        calls = [v for k, v in f.items() if "topk_select_kernel<false>" in k]
        n_calls = calls[0][0] if calls else 1
        fetch = sum(v[2] for k, v in f.items() if "topk_" in k) / n_calls * 1024 * 2
        write = sum(v[2] for k, v in w.items() if "topk_" in k) / n_calls * 1024
        traffic["topk5m"] = int(fetch + write)
    f, w = parse_pmc(os.path.join(SRC, "pmc_train_FETCH_SIZE.txt")), parse_pmc(os.path.join(SRC, "pmc_train_WRITE_SIZE.txt"))
    if f:
        g_f = [(v[0], v[2]) for k, v in f.items() if "gemm_dma_kernel" in k]
        g_w = [(v[0], v[2]) for k, v in w.items() if "gemm_dma_kernel" in k]
        n = sum(c for c, _ in g_f)
        traffic["train_gemm"] = int((sum(s for _, s in g_f) * 2 + sum(s for _, s in g_w)) / n * 1024)
        for name, key in (("sampled_fwd_kernel", "train_rt_sampled_loss_fwd_train"), ("sampled_bwd_rows_kernel", "train_rt_sampled_loss_bwd")):
            ff = [v for k, v in f.items() if name in k]
            ww = [v for k, v in w.items() if name in k]
            if ff:
                traffic[key] = int((ff[0][1] * 2 + (ww[0][1] if ww else 0)) * 1024)
    json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1)[:600])


if __name__ == "__main__":
    main()
