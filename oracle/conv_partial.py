"""Checker for a BUILD-INTERNAL byte format (TEST INFRASTRUCTURE, like the rest of oracle/): the partial rows between the two phases of
the build's sparse convolution (geopurify_amd/csrc/sparse_conv_v2.hip).  The reference has no such object -- its
convolution is MinkowskiEngine's (affinity_module.py:36-66) -- so this restates the build's own stated format, in numpy, for the tests
to hold the kernels to byte for byte.

Per pair row and 128-column quarter:  E  = exponent field of (largest |v| of the quarter, one unit in the last place added) -- an
                                           all-ones mantissa takes the next exponent;  255 when the quarter holds an Inf or a NaN
                                      u  = rint(v * 2^(148 - E)) + 2^22      (round half to even; 0 < u < 2^23), three little-endian bytes
A quarter's 384 bytes: the first 16 bytes of its 16 lanes (lane f = columns 8 f .. 8 f + 7, 24 bytes), then their last 8.
The exponent bytes follow the rows of the chunk at the next multiple of 16 bytes.  Decoded value = (u - 2^22) * 2^(E - 148): within half a
unit, 2^(E - 149) <= 2^-22 of the quarter's largest magnitude.  (The kernel clamps E to >= 22: values below 2^-104 -- not restated.)
"""
import numpy as np

QUARTER = 128


def exponents(v):
    """v f32 [P, cout] -> E int64 [P, cout / 128]"""
    P, cout = v.shape
    m = np.abs(v.reshape(P, cout // QUARTER, QUARTER).astype(np.float32)).max(axis=2)
    bits = m.view(np.uint32).astype(np.int64)
    nan = np.isnan(v.reshape(P, cout // QUARTER, QUARTER)).any(axis=2)
    E = (bits + 1) >> 23
    return np.where(nan | (E >= 255), 255, E)


def encode(v):
    """v f32 [P, cout] -> (rows uint8 [P * cout * 3], E uint8 [P * cout / 128]) in the kernel's layout (finite quarters only)"""
    P, cout = v.shape
    nq = cout // QUARTER
    E = exponents(v)
    assert (E >= 22).all() and (E < 255).all(), "the restatement covers finite quarters above 2^-104"
    u = np.rint(v.reshape(P, nq, QUARTER).astype(np.float64) * np.exp2(148.0 - E)[:, :, None]).astype(np.int64) + (1 << 22)
    assert u.min() > 0 and u.max() < (1 << 23)
    lanes = u.reshape(P, nq, 16, 8)
    b = np.stack([(lanes >> (8 * t)) & 255 for t in range(3)], axis=-1).astype(np.uint8).reshape(P, nq, 16, 24)
    rows = np.concatenate([b[..., :16].reshape(P, nq, 256), b[..., 16:].reshape(P, nq, 128)], axis=2).reshape(-1)
    return rows, E.astype(np.uint8).reshape(-1)


def decode(rows, E, P, cout):
    """the inverse: -> f64 [P, cout] (NaN where E == 255)"""
    nq = cout // QUARTER
    r = rows.reshape(P, nq, 384)
    b = np.concatenate([r[..., :256].reshape(P, nq, 16, 16), r[..., 256:].reshape(P, nq, 16, 8)], axis=3).reshape(P, nq, 16, 8, 3).astype(np.int64)
    u = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    Ei = E.reshape(P, nq).astype(np.int64)
    out = (u - (1 << 22)).astype(np.float64) * np.exp2(Ei - 148.0)[:, :, None, None]
    out[Ei == 255] = np.nan
    return out.reshape(P, cout)


def exponent_offset(P, cout):
    """byte offset of the exponent bytes inside the chunk's buffer"""
    return (P * cout * 3 + 15) & ~15
