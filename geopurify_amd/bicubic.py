"""Host-side tap tables for F.interpolate(mode="bicubic", align_corners=False, antialias=True)
(models/affinity_module.py:527-533).  The resized mask logits are only needed at the sampled pixels,
so the HIP kernel evaluates the separable filter there; this module produces the per-output-index
first tap and (<=4) fp32 weights exactly as the torch CPU kernel computes them: Keys a=-0.5 kernel
in fp32 with fused multiply-adds, mixed float/double index arithmetic, renormalised weights.
"""
import numpy as np

_f = np.float32
_d = np.float64


def _fma(a, b, c):
    return _f(_d(a) * _d(b) + _d(c))


def _cubic(x):
    A = _f(-0.5)
    x = _f(abs(x))
    if x < 1:
        t = _fma(_f(A + _f(2)), x, -_f(A + _f(3)))
        t = _f(t * x)
        return _fma(t, x, _f(1))
    if x < 2:
        t = _fma(A, x, -_f(_f(5) * A))
        t = _fma(t, x, _f(_f(8) * A))
        return _fma(t, x, -_f(_f(4) * A))
    return _f(0)


def aa_bicubic_taps(in_size, out_size, max_taps=4):
    """Returns (first index int32 [out], weights fp32 [out, max_taps] zero padded)."""
    scale = _f(_f(in_size) / _f(out_size))
    if scale > 1:
        raise ValueError("antialias bicubic down-sampling needs more than 4 taps; only up-sampling "
                         "(mask logits -> mask_shape) is on the hot path")
    support = _f(2.0)
    x0 = np.zeros(out_size, np.int32)
    w = np.zeros((out_size, max_taps), _f)
    for i in range(out_size):
        center = _f(_d(scale) * (i + 0.5))
        xmin = max(0, int(_d(_f(center - support)) + 0.5))
        xmax = min(in_size, int(_d(_f(center + support)) + 0.5))
        ws = [_cubic(_f(_d(_f(_f(j + xmin) - center)) + 0.5)) for j in range(xmax - xmin)]
        assert 0 < len(ws) <= max_taps
        tot = _f(0)
        for v in ws:
            tot = _f(tot + v)
        x0[i] = xmin
        for j, v in enumerate(ws):
            w[i, j] = _f(v / tot)
    return x0, w
