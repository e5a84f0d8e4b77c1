"""Copy the evidence of a `scripts/gpu/visit.sh r3 ...` visit (gpurun_out/r3/, see profiles/README.md for the command) into profiles/
(tracked), derive the matrix-pipe busy shares from the SQ pass and rebuild profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes.
FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so fetches of wide streaming reads are doubled
(MI355X_MICROARCH.md, HBM section)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r3"
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = os.path.join(ROOT, "profiles")
KEEP = [("1_bench.json", "bench_auto.json"), ("2_prof.md", "rocprof_kernel_trace_train.md"), ("3_bench.json", "bench_train_single_stream.json"),
        ("4_pmc.txt", "pmc_train_FETCH_SIZE.txt"), ("5_pmc.txt", "pmc_train_WRITE_SIZE.txt"), ("6_pmc.txt", "sq_counters_train_raw.txt"),
        ("7_prof.md", "rocprof_kernel_trace_topk5m.md"), ("8_pmc.txt", "pmc_topk5m_FETCH_SIZE.txt"), ("9_pmc.txt", "pmc_topk5m_WRITE_SIZE.txt"),
        ("10_prof.md", "rocprof_kernel_trace_topk5m_u4096.md"), ("11_prof.md", "rocprof_kernel_trace_bert4rec.md"),
        ("12_prof.md", "rocprof_kernel_trace_hstu.md"), ("13_prof.md", "rocprof_kernel_trace_recommend.md"),
        ("14_pmc.txt", "pmc_topk5m_u4096_FETCH_SIZE.txt"), ("15_prof.md", "rocprof_kernel_trace_esasrec_kpm.md"),
        ("16_pytest.txt", "pytest_gpu_final_tree.txt"), ("17_pmc.txt", "sq_counters_topk5m_u4096_raw.txt"), ("2_timeline.txt", "timeline_train.txt"),
        ("13_prof.md", "rocprof_kernel_trace_recommend.md"), ("18_prof.md", "rocprof_kernel_trace_esasrec.md"),
        ("19_pmc.txt", "sq_counters_hstu_raw.txt")]


def parse(path):
    """-> {kernel: {counter: (n, avg)}}"""
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(.+?)\s{2,}(\w+: n=.*)$", line.rstrip())
        if not m:
            continue
        d = {}
        for c in re.finditer(r"(\w+): n=(\d+) avg=([\d.e+-]+)", m.group(2)):
            d[c.group(1)] = (int(c.group(2)), float(c.group(3)))
        out[m.group(1).strip()] = d
    return out


def main():
    for src, dst in KEEP:
        p = os.path.join(SRC, src)
        if os.path.exists(p):
            text = open(p).read()
            if src.endswith(".json"):
                text = "\n".join(l for l in text.splitlines() if l.startswith("{")) + "\n"
            open(os.path.join(DST, f"{TAG}_{dst}"), "w").write(text)
        else:
            print("missing", src)
    sq = parse(os.path.join(SRC, "6_pmc.txt"))
    rows = ["GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: matrix-pipe busy share = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024); "
            "SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (x 4 = cycles).",
            "| kernel | launches | GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES | matrix-pipe busy share | VALU active share | LDS active share | mean waves per SIMD | LDS bank conflict / LDS active |",
            "|---|---|---|---|---|---|---|---|---|"]
    for k, c in sq.items():
        if not any(t in k for t in ("v2_", "v3_", "gemm_", "wgrad_", "ffn_", "sampled_", "layernorm", "adam", "embed_bwd_rows")):
            continue
        g = c.get("GRBM_GUI_ACTIVE", (0, 0))[1]
        if g <= 0:
            continue
        simd_cycles = g / 8 * 1024
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]
        valu = c.get("SQ_ACTIVE_INST_VALU", (0, 0))[1] * 4
        lds = c.get("SQ_ACTIVE_INST_LDS", (0, 0))[1] * 4
        wav = c.get("SQ_WAVE_CYCLES", (0, 0))[1] * 4
        conf = c.get("SQ_LDS_BANK_CONFLICT", (0, 0))[1]
        rows.append(f"| `{k[:60]}` | {c['GRBM_GUI_ACTIVE'][0]} | {g:.0f} | {mf:.0f} | {mf / simd_cycles:.3f} | {valu / simd_cycles:.3f} | {lds / (g / 8 * 256):.3f} | "
                    f"{wav / simd_cycles:.2f} | {conf / max(lds / 4, 1):.4f} |")
    open(os.path.join(DST, f"{TAG}_sq_counters_train.md"), "w").write("\n".join(rows) + "\n")
    sq2 = parse(os.path.join(SRC, "17_pmc.txt"))      # the 4,096-user top-k launch (round 5)
    if sq2:
        rows2 = rows[:3]
        for k, c in sq2.items():
            if not any(t in k for t in ("topk_", "to_hm_rows", "one_plane")):
                continue
            g = c.get("GRBM_GUI_ACTIVE", (0, 0))[1]
            if g <= 0:
                continue
            simd_cycles = g / 8 * 1024
            mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]
            valu = c.get("SQ_ACTIVE_INST_VALU", (0, 0))[1] * 4
            lds = c.get("SQ_ACTIVE_INST_LDS", (0, 0))[1] * 4
            wav = c.get("SQ_WAVE_CYCLES", (0, 0))[1] * 4
            conf = c.get("SQ_LDS_BANK_CONFLICT", (0, 0))[1]
            rows2.append(f"| `{k[:60]}` | {c['GRBM_GUI_ACTIVE'][0]} | {g:.0f} | {mf:.0f} | {mf / simd_cycles:.3f} | {valu / simd_cycles:.3f} | {lds / (g / 8 * 256):.3f} | "
                         f"{wav / simd_cycles:.2f} | {conf / max(lds / 4, 1):.4f} |")
        open(os.path.join(DST, f"{TAG}_sq_counters_topk5m_u4096.md"), "w").write("\n".join(rows2) + "\n")
    f, w = parse(os.path.join(SRC, "4_pmc.txt")), parse(os.path.join(SRC, "5_pmc.txt"))
    tf, tw = parse(os.path.join(SRC, "8_pmc.txt")), parse(os.path.join(SRC, "9_pmc.txt"))

    def kib(tab, pat, counter):
        return [(v[counter][0], v[counter][1]) for k, v in tab.items() if pat in k and counter in v]

    traffic = {"_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace only, "
                        f"`scripts/gpu/visit.sh {TAG}`); FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request), "
                        "WRITE_SIZE in KiB.  topk5m: whole two-stage call at 16 users = 2 launches of the one-plane stream kernel + merge / seed / replay (image 5.12e9); "
                        "topk5m_single_stage: rt_topk_score on the fp32 rows (10.24e9); topk5m_u4096: the fragment-major coarse pass + merge + replay (fetch only).  train_gemm: average launch over the GEMM kernel family of the step "
                        "(gemm_wp_kernel forward / dgrad products, gemm_dma_kernel weight gradients)."}
    # the 16-user launch: the two-stage call = 2 launches of the one-plane stream kernel (seeding prefix + main pass) + merge / seed / replay;
    # the bench's `single_stage` extra = 2 launches of topk_stream16_kernel over the fp32 rows.  (FETCH x 2: the gfx950 correction)
    def per_call(tab_f, tab_w, main_pat, small_pats, launches_per_call=2):
        m = kib(tab_f, main_pat, "FETCH_SIZE")
        if not m:
            return None
        calls = m[0][0] / launches_per_call
        fetch = sum(v["FETCH_SIZE"][0] * v["FETCH_SIZE"][1] for k, v in tab_f.items() if (main_pat in k or any(p in k for p in small_pats)) and "FETCH_SIZE" in v)
        write = sum(v["WRITE_SIZE"][0] * v["WRITE_SIZE"][1] for k, v in tab_w.items() if (main_pat in k or any(p in k for p in small_pats)) and "WRITE_SIZE" in v)
        return int((fetch * 2 + write) / calls * 1024)

    v = per_call(tf, tw, "topk_stream_kernel<1, 6", ("topk_merge_kernel", "topk_seed_kernel", "topk_replay_kernel"))
    if v:
        traffic["topk5m"] = v
    v = per_call(tf, tw, "topk_stream16_kernel", ())
    if v:
        traffic["topk5m_single_stage"] = v
    u4 = parse(os.path.join(SRC, "14_pmc.txt"))
    v = per_call(u4, {}, "topk_coarse_frag_kernel", ("topk_merge_kernel", "topk_replay_kernel"), launches_per_call=1)
    if v:
        traffic["topk5m_u4096"] = v
    fam = ("gemm_wp_kernel", "gemm_dma_kernel", "gemm_dma_group", "wgrad_group_kernel", "ffn_kernel")
    gf = [x for pat in fam for x in kib(f, pat, "FETCH_SIZE")]
    gw = [x for pat in fam for x in kib(w, pat, "WRITE_SIZE")]
    if gf:
        n = sum(c for c, _ in gf)
        traffic["train_gemm"] = int((sum(c * a for c, a in gf) * 2 + sum(c * a for c, a in gw)) / n * 1024)
    for key, pat in (("train_rt_sampled_loss_fwd_train", "sampled_fwd_kernel"), ("train_rt_sampled_loss_bwd", "sampled_bwd_rows_kernel"),
                     ("train_v3_fwd_kernel", "v3_fwd_kernel"), ("train_v3_bwd_dq_kernel", "v3_bwd_dq_kernel"), ("train_v3_bwd_dkv_kernel", "v3_bwd_dkv_kernel"),
                     ("train_ffn_kernel_fwd", "ffn_kernel<0>"), ("train_ffn_kernel_bwd", "ffn_kernel<1>"), ("train_wgrad_group_kernel", "wgrad_group_kernel")):
        a, b = kib(f, pat, "FETCH_SIZE"), kib(w, pat, "WRITE_SIZE")
        if a:
            traffic[key] = int((a[0][1] * 2 + (b[0][1] if b else 0)) * 1024)
    json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print(open(os.path.join(DST, f"{TAG}_sq_counters_train.md")).read())


if __name__ == "__main__":
    main()
