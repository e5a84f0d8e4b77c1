"""Where the HOST spends a C2 training step (python + autograd + ctypes + HIP runtime): cProfile over 300 steps of the product loop.
   python scripts/host_profile.py [n_steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from rectools_amd.models import SASRecModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ds = bench.make_ml20m_dataset()
model = SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, dropout_rate=0.2, loss="sampled_softmax", n_negatives=128,
                    batch_size=128, lr=1e-3, epochs=1, seed=32)
model._build_model_from_dataset(ds)
loop = model.training_loop()
model.lightning_model.train()
loop.begin_epoch(0)
for _ in range(30):
    loop.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    loop.step()
issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print("compiled step (lightning.NativeSasrecStep):", loop._native is not None)
print(f"{n} steps: host issue {issue / n * 1e3:.3f} ms/step, wall {wall / n * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    loop.step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    txt = s.getvalue()
    print("\n".join(l[:170] for l in txt.splitlines()[:48]))
