#!/usr/bin/env python3
"""The student's output layer: exact-fp32 kernel + l2norm_rows_ (rounds 1-3) against gp_embed_head_f16x3 on the planes the last
3x3x3 layer writes.  usage: bench_embed_head.py [nv]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
X = torch.relu(torch.randn(nv, 512, device="cuda"))
W = torch.randn(512, 128, device="cuda") * 0.04
p2 = 2.0 ** int(np.floor(np.log2(16384.0 / float(W.abs().max()))))
hi, lo = ops.conv_weights_split(W.reshape(1, 512, 128).contiguous(), p2)
xs = ops.split_f16(X, 512, per_row=True)


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


out = torch.empty((nv, 128), device="cuda")
t_old = timeit(lambda: ops.l2norm_rows_(ops.sparse_conv(X, None, W, out=out)))
t_new = timeit(lambda: ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2], out=out))
byt = nv * (512 * 4 + 128 * 4)
print(f"nv {nv}: fp32 kernel + l2norm {t_old:7.1f} us   embed_head_f16x3 {t_new:7.1f} us = {byt / t_new / 1e6:6.2f} TB/s of {byt / 1e6:.0f} MB "
      f"({byt / t_new / 1e6 / 8.0:.3f} of 8 TB/s)")
a = ops.l2norm_rows_(ops.sparse_conv(X, None, W))
b = ops.embed_head_f16x3(xs[:2], hi, lo, 1.0 / p2, x_row_inv=xs[2])
print("max |difference| of the normalised embeddings:", float((a - b).abs().max()))
