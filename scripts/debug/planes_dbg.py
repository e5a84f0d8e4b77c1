"""Which products of a LiGR / STU stack find their weight planes (ops._gemm_w)?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rectools_amd import nn as hnn, ops, lightning as hl

torch.manual_seed(0)
for name, stack in (("ligr", hnn.LiGRLayers(2, 512, 4, 0.2)), ("stu", hnn.STULayers(2, 256, 4, 64, 64, 512, True, True))):
    stack = stack.cuda()
    opt = hl.FlatAdam(stack, lr=1e-3)
    pl = stack._fresh_planes()
    print(name, "planes:", None if pl is None else (pl.ok, pl.n, pl.stride, pl.lo % 32))
    with ops.active_planes(pl):
        for n, p in stack.named_parameters():
            if p.dim() == 2:
                r = ops._planes_of(p)
                print("   ", n, tuple(p.shape), None if r is None else (r[0] % 16, r[1]))
