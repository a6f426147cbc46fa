import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geopurify_amd import ops
N, Nv, D, C = 150000, 133933, 512, 20
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(Nv, D, device="cuda", generator=g)
idx = torch.randint(0, Nv, (N,), device="cuda", generator=g).sort().values
rm = torch.randperm(Nv, device="cuda", generator=g).to(torch.int32)
text = torch.nn.functional.normalize(torch.randn(C, D, device="cuda", generator=g), dim=1)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def two():
    o = ops.gather_rows(X, D, idx, row_map=rm)
    return ops.classify_argmax(o, text, 14.0)
for rep in range(3):
    print(f"gather_rows + classify_argmax: {t(two):.1f} us   gather_rows_classify: {t(lambda: ops.gather_rows_classify(X, D, idx, text, 14.0, row_map=rm)):.1f} us   gather alone {t(lambda: ops.gather_rows(X, D, idx, row_map=rm)):.1f} us")
