"""Does a family leg of bench.py's auto run depend on what ran before it in the process?  (r5: families.hstu 17.4 k in the auto line, 21.7 k
from `--workload hstu` on the same box.)  Runs the HSTU leg alone, after the 4,096-user top-k leg, and after the BERT4Rec leg."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench      # noqa: E402

args = argparse.Namespace(gpus=1, steps=20, warmup=6, workload="auto", users_per_pass=0, users_per_step=0, n_negatives=128, rec_steps=20,
                          topk_steps=2, no_cpu_baseline=True, no_families=False, no_host_only=True)


def hstu(tag, kind="hstu"):
    t = time.time()
    v, wall, roof, info = bench.run_train(args, 0, 1, kind)
    print(f"{tag}: {kind} {v:.0f} seqs/s, {wall / args.steps * 1e3:.3f} ms/step, gemm frac {roof.get('frac')}  ({time.time() - t:.0f} s)", flush=True)
    return info


which = sys.argv[1:] or ["alone", "topk", "bert"]
LATE = "late" in which
if "alone" in which:
    del_info = hstu("alone"); del del_info; torch.cuda.empty_cache()
    del_info = hstu("alone again"); del del_info; torch.cuda.empty_cache()
if "train" in which:
    a2 = argparse.Namespace(**vars(args)); a2.steps, a2.warmup = 60, 10
    v, wall, roof, info = bench.run_train(a2, 0, 1, "train")
    print("train", v, flush=True)
    if "exact" in which:
        os.environ["RT_GEMM_SPLIT"] = "exact"
        for _ in range(45):
            info["loop"].step()
        torch.cuda.synchronize()
        os.environ.pop("RT_GEMM_SPLIT", None)
        del_info = hstu("after train + exact leg (train model alive)"); del del_info; torch.cuda.empty_cache()
    if "e2e" in which:
        print("e2e", bench.run_recommend_e2e(info)["value"], flush=True)
    del info; torch.cuda.empty_cache()
    del_info = hstu("after train" + (" + e2e" if "e2e" in which else "")); del del_info; torch.cuda.empty_cache()
if "rec" in which:
    r = bench.topk_leg("recommend", args, 0, 1, False)
    print("recommend leg", r["value"], flush=True)
    del_info = hstu("after the recommend top-k leg"); del del_info; torch.cuda.empty_cache()
if "topk16" in which:
    a3 = argparse.Namespace(**vars(args)); a3.topk_steps = 40
    r = bench.topk_leg("topk5m", a3, 0, 1, False)
    print("topk5m leg", r["value"], flush=True)
    del_info = hstu("after the 16-user top-k leg"); del del_info; torch.cuda.empty_cache()
if "topk" in which:
    big = argparse.Namespace(**vars(args)); big.users_per_step = 4096
    r = bench.topk_leg("topk5m", big, 0, 1, False)
    print("u4096", r["value"], r["ms_per_step"], flush=True)
    del_info = hstu("after u4096"); del del_info; torch.cuda.empty_cache()
if "bert" in which:
    info = hstu("bert", "bert4rec")
    print("bert recommend", bench.family_recommend(info, "bert4rec")["value"], flush=True)
    del info; torch.cuda.empty_cache()
    del_info = hstu("after bert4rec + its recommend"); del del_info
