"""Tensor-level wrappers over the C-ABI (include/geopurify_hip.h).

Every function takes/returns torch CUDA tensors but hands the library raw device pointers, sizes
and the current HIP stream.  PyTorch is plumbing here: allocation and stream ownership only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _chk(t, dtype, name):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous CUDA {dtype}, got {t.dtype} cuda={t.is_cuda} contig={t.is_contiguous()}")
    return t


# host time spent BLOCKED in the hot path's device -> host read-backs (voxel count, per-view counts, chunk plan, pair bounds, union
# rows): a scheduler that runs them ahead (bench.py's look-ahead) reports its own enqueue time net of these waits
READBACK = {"seconds": 0.0, "calls": 0}


def readback(t):
    """t.tolist() of a small device tensor, the wait accounted in READBACK (one host synchronisation of the calling stream)"""
    import time
    t0 = time.perf_counter()
    v = t.tolist()
    READBACK["seconds"] += time.perf_counter() - t0
    READBACK["calls"] += 1
    return v


def _dbl16(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(16))
    return (ctypes.c_double * 16)(*a.tolist())


# ------------------------------------------------------------------------------------------ rows 1-3
def voxelize(coords, rigid):
    """coords f64 [N,3] cuda; rigid 4x4 (host).  Returns dict with coords_aug f64 [Nv,3], inds,
    inds_reconstruct, order, seg_start (CSR of each voxel's points), nv.  One host sync (nv)."""
    lib = _lib.load()
    _chk(coords, torch.float64, "coords")
    n = coords.shape[0]
    dev = coords.device
    ws = _ws(lib.gp_voxelize_workspace_bytes(n), dev)
    ca = torch.empty((n, 3), dtype=torch.float64, device=dev)
    inds = torch.empty(n, dtype=torch.int64, device=dev)
    inv = torch.empty(n, dtype=torch.int64, device=dev)
    order = torch.empty(n, dtype=torch.int64, device=dev)
    seg = torch.empty(n + 1, dtype=torch.int64, device=dev)
    nvd = torch.zeros(1, dtype=torch.int64, device=dev)
    check(lib.gp_voxelize_f64(_ptr(coords), n, _dbl16(rigid), _ptr(ca), _ptr(inds), _ptr(inv), _ptr(nvd),
                              _ptr(order), _ptr(seg), _ptr(ws), ws.numel(), _stream()), "gp_voxelize_f64")
    nv = int(readback(nvd)[0])
    return {"coords_aug": ca[:nv], "inds": inds[:nv], "inds_reconstruct": inv, "order": order,
            "seg_start": seg[:nv + 1], "nv": nv}


def fnv_hash(coords):
    lib = _lib.load()
    _chk(coords, torch.float64, "coords")
    out = torch.empty(coords.shape[0], dtype=torch.int64, device=coords.device)   # bit pattern of uint64
    check(lib.gp_fnv_hash_f64(_ptr(coords), coords.shape[0], _ptr(out), _stream()), "gp_fnv_hash_f64")
    return out


def project_points(coords, w2c, fx, fy, cx, cy, depth, width, height, cut_bound, vis_thres, want_weight=False):
    lib = _lib.load()
    _chk(coords, torch.float64, "coords")
    n = coords.shape[0]
    if depth is not None:
        _chk(depth, torch.float64, "depth")
        assert depth.shape == (height, width)
    mapping = torch.empty((n, 3), dtype=torch.int64, device=coords.device)
    weight = torch.empty(n, dtype=torch.float64, device=coords.device) if want_weight else None
    check(lib.gp_project_points_f64(_ptr(coords), n, _dbl16(w2c), float(fx), float(fy), float(cx), float(cy),
                                    _ptr(depth), int(width), int(height), int(cut_bound), float(vis_thres),
                                    _ptr(mapping), _ptr(weight), _stream()), "gp_project_points_f64")
    return (mapping, weight) if want_weight else mapping


def render_depth(coords, w2c, fx, fy, cx, cy, width, height, cut_bound):
    """z-buffer of the cloud (ScanNet mapper, depth given as a str: fusion_util.py:126-130) -> f64 [H,W]."""
    lib = _lib.load()
    _chk(coords, torch.float64, "coords")
    depth = torch.empty((height, width), dtype=torch.float64, device=coords.device)
    check(lib.gp_render_depth_f64(_ptr(coords), coords.shape[0], _dbl16(w2c), float(fx), float(fy), float(cx), float(cy),
                                  int(width), int(height), int(cut_bound), _ptr(depth), _stream()), "gp_render_depth_f64")
    return depth


# ------------------------------------------------------------------------------------------ order / grid
def minmax_i32(coords):
    """Per-axis (min xyz, max xyz) of int32 coordinates [n,3] -> int32 [6] on the device, no sync."""
    lib = _lib.load()
    _chk(coords, torch.int32, "coords")
    mm = torch.empty(6, dtype=torch.int32, device=coords.device)
    check(lib.gp_minmax_i32(_ptr(coords), coords.shape[0], _ptr(mm), _stream()), "gp_minmax_i32")
    return mm


def morton_order(coords_i32):
    lib = _lib.load()
    _chk(coords_i32, torch.int32, "coords")
    nv = coords_i32.shape[0]
    dev = coords_i32.device
    ws = _ws(lib.gp_morton_order_workspace_bytes(nv), dev)
    perm = torch.empty(nv, dtype=torch.int32, device=dev)
    rank = torch.empty(nv, dtype=torch.int32, device=dev)
    check(lib.gp_morton_order(_ptr(coords_i32), nv, _ptr(perm), _ptr(rank), _ptr(ws), ws.numel(), _stream()),
          "gp_morton_order")
    return perm, rank


class Grid:
    """Opaque lattice grid buffer (device) + the host-side origin/extent it was built with."""

    def __init__(self, buf, origin, extent, nv):
        self.buf, self.origin, self.extent, self.nv = buf, origin, extent, nv

    def status(self):
        return int(self.buf[36:40].view(torch.int32).item())      # GpGridHeader.status


def grid_build(coords_sorted, origin=None, extent=None):
    """coords_sorted i32 [nv,3] in Morton order.  origin/extent (host ints) are computed with one
    sync if not given."""
    lib = _lib.load()
    _chk(coords_sorted, torch.int32, "coords")
    nv = coords_sorted.shape[0]
    if origin is None or extent is None:
        lo = coords_sorted.amin(0)
        hi = coords_sorted.amax(0)
        lohi = torch.stack([lo, hi]).cpu().tolist()
        origin = lohi[0]
        extent = [lohi[1][a] - lohi[0][a] + 1 for a in range(3)]
    o = (ctypes.c_int32 * 3)(*[int(v) for v in origin])
    e = (ctypes.c_int32 * 3)(*[int(v) for v in extent])
    nbytes = lib.gp_grid_bytes(nv, e)
    if nbytes == 0:
        raise _lib.GeoPurifyHipError("gp_grid_bytes: invalid extent")
    buf = _ws(nbytes, coords_sorted.device)
    check(lib.gp_grid_build(_ptr(coords_sorted), nv, o, e, _ptr(buf), buf.numel(), _stream()), "gp_grid_build")
    return Grid(buf, list(origin), list(extent), nv)


def kernel_map_build(grid, coords_sorted):
    lib = _lib.load()
    nv = coords_sorted.shape[0]
    nm = torch.empty((27, nv), dtype=torch.int32, device=coords_sorted.device)
    check(lib.gp_kernel_map_build(_ptr(grid.buf), _ptr(coords_sorted), nv, _ptr(nm), _stream()), "gp_kernel_map_build")
    return nm


# ------------------------------------------------------------------------------------------ rows 8-12
def scatter_mean_csr(src, d, order, seg_start, nv, out, col0=0, row_map=None):
    lib = _lib.load()
    _chk(src, torch.float32, "src")
    check(lib.gp_scatter_mean_csr(_ptr(src), src.stride(0), int(d), _ptr(order), _ptr(seg_start), int(nv),
                                  _ptr(row_map), _ptr(out), out.stride(0), int(col0), _stream()),
          "gp_scatter_mean_csr")
    return out


def gather_rows(src, d, index, out=None, row_map=None):
    lib = _lib.load()
    n = index.shape[0]
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=src.device)
    check(lib.gp_gather_rows(_ptr(src), src.stride(0), int(d), _ptr(index), n, _ptr(row_map), _ptr(out),
                             out.stride(0), _stream()), "gp_gather_rows")
    return out


def gather_rows_classify(src, d, index, text_norm, logit_scale, row_map=None):
    """gather_rows + classify_argmax in one pass (gp_gather_rows_classify): returns (out [n, d], pred i64 [n], zero u8 [n]).
    d a multiple of 64 up to 512 and C * d * 4 <= 64 KiB (can_gather_rows_classify)."""
    lib = _lib.load()
    n = index.shape[0]
    out = torch.empty((n, d), dtype=torch.float32, device=src.device)
    pred = torch.empty(n, dtype=torch.int64, device=src.device)
    zero = torch.empty(n, dtype=torch.uint8, device=src.device)
    check(lib.gp_gather_rows_classify(_ptr(src), src.stride(0), int(d), _ptr(index), n, _ptr(row_map), _ptr(out), out.stride(0),
                                      _ptr(text_norm), int(text_norm.shape[0]), float(logit_scale), _ptr(pred), _ptr(zero), _stream()),
          "gp_gather_rows_classify")
    return out, pred, zero


def can_gather_rows_classify(d, num_classes):
    return d % 64 == 0 and d <= 512 and num_classes * d * 4 <= 64 * 1024


def sparse_conv(x, nbr_map, w, scale=None, shift=None, residual=None, relu=False, out=None):
    """x fp32 [nv, >=cin] (row stride = x.stride(0)); w fp32 [27,cin,cout] or [cin,cout]."""
    lib = _lib.load()
    nv = x.shape[0]
    if w.dim() == 3:
        kv, cin, cout = w.shape
    else:
        kv, (cin, cout) = 1, w.shape
    _chk(w, torch.float32, "w")
    if out is None:
        out = torch.empty((nv, cout), dtype=torch.float32, device=x.device)
    check(lib.gp_sparse_conv(_ptr(x), x.stride(0), _ptr(nbr_map) if kv > 1 else None, nv, _ptr(w), int(kv),
                             int(cin), int(cout), _ptr(scale), _ptr(shift), _ptr(residual),
                             residual.stride(0) if residual is not None else 0, int(bool(relu)), _ptr(out),
                             out.stride(0), _stream()), "gp_sparse_conv")
    return out


class ConvPairs:
    """Compacted kernel map of one scene (shared by every 3x3x3 layer)."""

    def __init__(self, pair_in, pair_pos, seg_off, tile_start, nseg, num_pairs, nv):
        self.pair_in, self.pair_pos, self.pair_off, self.tile_start = pair_in, pair_pos, seg_off, tile_start
        self.nseg, self.num_pairs, self.nv = nseg, num_pairs, nv
        self.partial = None
        self.num_chunks, self.chunk_row_off, self.chunk_tile_off, self.chunk_pair_off = 0, None, None, None   # 0 = not chunked

    def _set_chunks(self, rows, tiles, pairs):
        n = len(rows) - 1
        self.num_chunks = n
        self.chunk_row_off = (ctypes.c_int32 * (n + 1))(*rows)
        self.chunk_tile_off = (ctypes.c_int32 * (n + 1))(*tiles)
        self.chunk_pair_off = (ctypes.c_int32 * (n + 1))(*pairs)
        self.max_chunk_pairs = max(pairs[i + 1] - pairs[i] for i in range(n))

    def regroup(self, g):
        """The same pair ORDER (chunk-major: a chunk's 27 offset segments gather from the same few thousand input rows, which
        keeps the gathered operand in the XCDs' L2) executed `g` chunks per LAUNCH: phase 1 of g consecutive chunks is one grid
        (fewer partly filled last rounds and kernel boundaries per layer), phase 2 follows for their rows; the partial buffer
        then holds g chunks.  Returns a ConvPairs sharing the device arrays."""
        g = max(1, min(int(g), max(self.num_chunks, 1)))
        cp = ConvPairs(self.pair_in, self.pair_pos, self.pair_off, self.tile_start, self.nseg, self.num_pairs, self.nv)
        cp.tile_desc = self.tile_desc
        n = self.num_chunks
        idx = list(range(0, n, g)) + [n]
        ro, po, to = list(self.chunk_row_off), list(self.chunk_pair_off), list(self.chunk_tile_off)
        cp._set_chunks([ro[i] for i in idx], [to[i] for i in idx], [po[i] for i in idx])
        return cp


# phase-1 tiles per chunk launch of the balanced chunking: None = three rounds of one-tile workgroups less 16 = 3 x the CU count - 16
# (752 on MI355X).  Rounds 3-4, fp32 partial rows: two rounds (512; 768 was slower -- 200 MB of partial rows per chunk).  Round 5, 24-bit
# partial rows (150 MB per 768 tiles): 752 / 768 beat 512 by 1.1 % of the scene, 640 (2.5 rounds) loses 4 %, 1024 is 0.5 % behind 768
CONV_TARGET_TILES = None       # (a tuning script may set it: scripts/conv_layer_time.py)


def conv_pairs_build(nbr_map, chunk_rows="balanced", col_tiles=2):
    """nbr_map i32 [27,nv] -> ConvPairs.  Pairs are ordered chunk-major and phase 1 / phase 2 run chunk by chunk, so the partial
    buffer only holds one chunk.  chunk_rows:
      "balanced" (default): chunk heights chosen on the device from the kernel map (gp_conv_chunk_plan) so that every chunk's
          phase-1 launch is at most 3 x CUs - 16 tiles = three rounds of workgroups (col_tiles = cout / 256 column tiles
          per row tile); two host syncs (the plan, the pair bounds);
      an int: equal heights.  Round 4 on the S scene (profiles/r04_conv_launch_groups.log), per 512->512 layer: 8192 rows (17
          launches, 128 MB of partial rows -- inside the Infinity Cache) 1.87 ms, 16384 rows (250 MB) 1.96 ms, 4096 rows 2.65 ms;
          a layer's time follows the rounds of 256 tiles its launches need (profiles/r04_conv_rounds.log); several chunks per
          launch (ConvPairs.regroup) never wins;  None = one chunk.  One host sync."""
    lib = _lib.load()
    kv, nv = nbr_map.shape
    dev = nbr_map.device
    if isinstance(chunk_rows, str):
        if chunk_rows != "balanced":
            raise ValueError(f"conv_pairs_build: chunk_rows={chunk_rows!r}")
    if isinstance(chunk_rows, str):
        granule = 256
        max_chunks = (nv + granule - 1) // granule
        plan = torch.empty(max_chunks + 2, dtype=torch.int32, device=dev)          # [0 .. max_chunks]: row offsets, [-1]: count
        ws = _ws(lib.gp_conv_chunk_plan_workspace_bytes(nv, granule), dev)
        target = CONV_TARGET_TILES or 3 * torch.cuda.get_device_properties(dev).multi_processor_count - 16
        check(lib.gp_conv_chunk_plan(_ptr(nbr_map), nv, kv, granule, int(col_tiles), int(target), max_chunks, _ptr(plan),
                                     _ptr(plan[max_chunks + 1:]), _ptr(ws), ws.numel(), _stream()), "gp_conv_chunk_plan")
        host = readback(plan)                                                        # host sync 1 of 2
        nchunks = host[max_chunks + 1]
        rows = host[:nchunks + 1]
        row_off = plan[:nchunks + 1]
    else:
        ch = nv if chunk_rows is None else max(int(chunk_rows), 256)
        rows = list(range(0, nv, ch)) + [nv]
        nchunks = len(rows) - 1
        row_off = torch.tensor(rows, dtype=torch.int32, device=dev)
    nseg = nchunks * kv
    ws = _ws(lib.gp_conv_pairs_workspace_bytes(nv, kv), dev)
    pair_in = torch.empty(kv * nv, dtype=torch.int32, device=dev)
    pair_pos = torch.empty((kv, nv), dtype=torch.int32, device=dev)
    seg_off = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
    tile_start = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
    tile_desc = torch.empty(((kv * nv) // 256 + nseg + 1, 4), dtype=torch.int32, device=dev)
    check(lib.gp_conv_pairs_build(_ptr(nbr_map), nv, kv, nchunks, _ptr(row_off), _ptr(pair_in), _ptr(pair_pos), _ptr(seg_off),
                                  _ptr(tile_start), _ptr(tile_desc), _ptr(ws), ws.numel(), _stream()), "gp_conv_pairs_build")
    bounds = readback(torch.stack([seg_off[::kv], tile_start[::kv]]))          # the host sync of this call (number of pairs)
    num_pairs = bounds[0][-1]
    cp = ConvPairs(pair_in, pair_pos, seg_off, tile_start, nseg, num_pairs, nv)
    cp.tile_desc = tile_desc
    # host copies of the chunk boundaries (exact row / tile / pair counts per chunk; a single chunk included)
    cp._set_chunks(rows, bounds[1], bounds[0])
    return cp


def conv_weights_split(w, scale_pow2, blocked=None, transpose_flip=False):
    """w fp32 [kv,cin,cout] -> (w_hi, w_lo) f16 of scale_pow2*w.
    blocked (default: whenever the shape allows, cin % 32 == 0 and cout % 256 == 0): the step-blocked layout of the two-phase
    convolution, tensors of shape [kv, cout/256, cin/32, 256, 32] -- sparse_conv_f16x3 recognises it by its five dimensions;
    else [kv,cout,cin] (the dense output layer's operand, gp_embed_head_f16x3; also accepted by sparse_conv_f16x3).
    transpose_flip: the halves of V[k] = w[kv-1-k]^T (the data-gradient operand; V's cin = w's cout and vice versa) straight from w."""
    lib = _lib.load()
    kv, cin, cout = w.shape
    if transpose_flip:
        if cout % 32 == 0 and cin % 256 == 0 and blocked is not False:
            hi = torch.empty((kv, cin // 256, cout // 32, 256, 32), dtype=torch.float16, device=w.device)
            lo = torch.empty_like(hi)
            check(lib.gp_conv_weights_split_blocked(_ptr(w), kv, cout, cin, float(scale_pow2), _ptr(hi), _ptr(lo), 1, _stream()),
                  "gp_conv_weights_split_blocked")
            return hi, lo
        return conv_weights_split(w.flip(0).transpose(1, 2).contiguous(), scale_pow2, blocked)
    if blocked is None:
        blocked = cin % 32 == 0 and cout % 256 == 0
    if blocked:
        hi = torch.empty((kv, cout // 256, cin // 32, 256, 32), dtype=torch.float16, device=w.device)
        lo = torch.empty_like(hi)
        check(lib.gp_conv_weights_split_blocked(_ptr(w), kv, cin, cout, float(scale_pow2), _ptr(hi), _ptr(lo), 0, _stream()),
              "gp_conv_weights_split_blocked")
        return hi, lo
    hi = torch.empty((kv, cout, cin), dtype=torch.float16, device=w.device)
    lo = torch.empty((kv, cout, cin), dtype=torch.float16, device=w.device)
    check(lib.gp_conv_weights_split(_ptr(w), kv, cin, cout, float(scale_pow2), _ptr(hi), _ptr(lo), _stream()),
          "gp_conv_weights_split")
    return hi, lo


def conv_weights_shape(w_hi):
    """(kv, cout, cin) of a split weight tensor in either layout"""
    if w_hi.dim() == 5:
        return int(w_hi.shape[0]), int(w_hi.shape[1]) * 256, int(w_hi.shape[2]) * 32
    return tuple(int(v) for v in w_hi.shape)


def split_f16(x, d=None, scale=None, per_row=False, interleaved=False, dst_row=None, extra_zero_rows=0):
    """fp32 rows -> (hi, lo) f16 rows with x * s = hi + lo.  Unscaled (s = 1): exact to 2^-22 relative only for |x| >= 2^-3
    (below that the lo half is a subnormal f16: absolute error 2^-25).  scale = device scalar from pow2_scale(): one power of
    two for the whole block; per_row=True: a power of two per row, returns (hi, lo, row_inv_scale).
    dst_row (i32 [n], a permutation; scaled forms only): row r is written to row dst_row[r] of the outputs (rcb_order's map).
    extra_zero_rows (plane forms): that many all-zero rows behind the n rows of hi and lo (the weight gradient's padded pairs point there)."""
    lib = _lib.load()
    d = x.shape[1] if d is None else d
    if interleaved:
        # (rows, None, row_inv_scale): per 32-column step [hi 32 | lo 32] in one tensor -- the operand form sparse_conv_f16x3 stages in full lines
        assert per_row and d % 32 == 0, "interleaved rows are the row-scaled operand of the convolution"
        rows = torch.empty((x.shape[0], 2 * d), dtype=torch.float16, device=x.device)
        rinv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        check(lib.gp_split_f16_scaled(_ptr(x), x.stride(0), int(d), x.shape[0], _ptr(rows), None, rows.stride(0), None, _ptr(rinv),
                                      _ptr(dst_row), _stream()), "gp_split_f16_scaled")
        return rows, None, rinv
    hi = torch.empty((x.shape[0] + extra_zero_rows, d), dtype=torch.float16, device=x.device)
    lo = torch.empty((x.shape[0] + extra_zero_rows, d), dtype=torch.float16, device=x.device)
    if extra_zero_rows:
        hi[x.shape[0]:].zero_()
        lo[x.shape[0]:].zero_()
    if scale is None and not per_row:
        if dst_row is not None:
            raise ValueError("split_f16: dst_row needs a scaled form (scale= or per_row=True)")
        check(lib.gp_split_f16(_ptr(x), x.stride(0), int(d), x.shape[0], _ptr(hi), _ptr(lo), hi.stride(0), _stream()),
              "gp_split_f16")
        return hi, lo
    rinv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device) if per_row else None
    check(lib.gp_split_f16_scaled(_ptr(x), x.stride(0), int(d), x.shape[0], _ptr(hi), _ptr(lo), hi.stride(0), _ptr(scale),
                                  _ptr(rinv), _ptr(dst_row), _stream()), "gp_split_f16_scaled")
    return (hi, lo, rinv) if per_row else (hi, lo)


def rcb_order(coords_sorted, chunk_rows=1024, leaf_rows=128):
    """Row order for the pooling operator (gp_rcb_order): recursive coordinate bisection of the Morton-ordered integer coords
    [nv, 3] inside chunks of chunk_rows rows into leaves of leaf_rows rows.  Returns (sigma, rho) i32 [nv]: new position -> row,
    row -> new position."""
    lib = _lib.load()
    _chk(coords_sorted, torch.int32, "coords")
    nv = coords_sorted.shape[0]
    sigma = torch.empty(nv, dtype=torch.int32, device=coords_sorted.device)
    rho = torch.empty(nv, dtype=torch.int32, device=coords_sorted.device)
    check(lib.gp_rcb_order(_ptr(coords_sorted), nv, int(chunk_rows), int(leaf_rows), _ptr(sigma), _ptr(rho), _stream()), "gp_rcb_order")
    return sigma, rho


def rows_renumber(nbr, sigma, rho):
    """out[p, j] = rho[nbr[sigma[p], j]]: neighbour lists i32 [nv, k] in the order / numbering of rcb_order"""
    lib = _lib.load()
    nv, k = nbr.shape
    out = torch.empty_like(nbr)
    check(lib.gp_rows_renumber_i32(_ptr(nbr), nv, int(k), _ptr(sigma), _ptr(rho), _ptr(out), _stream()), "gp_rows_renumber_i32")
    return out


def pow2_scale(x, d=None):
    """Device tensor [s, 1/s]: s = the power of two that puts max |x[:, :d]| into [2^13, 2^14).  No host sync."""
    lib = _lib.load()
    d = x.shape[1] if d is None else d
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = _ws(256, x.device)
    check(lib.gp_pow2_scale(_ptr(x), x.stride(0), int(d), x.shape[0], _ptr(out), _ptr(ws), ws.numel(), _stream()), "gp_pow2_scale")
    return out


def sparse_conv_f16x3(x, pairs, w_hi, w_lo, scale=None, shift=None, residual=None, relu=False, out=None,
                      x_split=None, out_split=None, x_row_inv=None, out_row_inv=None, want_f32=True, fp32_partials=False, dense_single_offset=False):
    """x fp32 [nv, >=cin] and/or x_split=(hi, lo) f16 (pre-split operand -> LDS-DMA path);
    out_split=(hi, lo) f16 buffers to also receive the split output.  x_row_inv fp32 [nv]: the per-row inverse scales of
    a row-scaled x_split; out_row_inv fp32 [nv]: receive the output's (out_split is then row-scaled).
    residual: fp32 rows [nv, >=cout], or a tuple (hi, lo, row_inv | None) of the split planes an earlier layer wrote -- the producer then
    needs no fp32 copy.
    INTERLEAVED rows: any of x_split / out_split / residual may be (t, None[, row_inv]) with t f16 [nv, 2 * channels] holding per 32-channel
    step [hi 32 | lo 32]: the LDS-DMA kernel stages a row and step as one full 128-byte line (interleave_planes / deinterleave_planes
    convert).
    fp32_partials=True: the partial rows between the two phases as fp32 (the format of rounds 1-4, which the 24-bit block-floating rows
    are checked against; also what a call takes by itself when a chunk's 24-bit rows would pass 4 GiB) -- an explicit argument of the
    call (plane_flags bit 3), not a process-wide switch.
    dense_single_offset=True (plane_flags bit 4): kv = 1 and every output row has its pair -- a gather-GEMM; phase 1 writes the fp32 output
    itself, no partial rows, no phase 2."""
    lib = _lib.load()
    res_planes = residual if isinstance(residual, (tuple, list)) else None
    if res_planes is not None:
        residual = None
    rh, rl, ri = (tuple(res_planes) + (None,))[:3] if res_planes is not None else (None, None, None)
    kv, cout, cin = conv_weights_shape(w_hi)
    w_blocked = int(w_hi.dim() == 5)
    nv = pairs.nv
    dev = w_hi.device
    if dense_single_offset:
        partial = torch.empty(64, dtype=torch.float32, device=dev)                    # (not written: phase 1 stores into `out`)
    else:
        if pairs.partial is None or pairs.partial.shape[1] < cout or pairs.partial.shape[0] < max(pairs.max_chunk_pairs, 1):
            pairs.partial = torch.empty((max(pairs.max_chunk_pairs, 1), cout), dtype=torch.float32, device=dev)
        partial = pairs.partial
    if out is None and (want_f32 or out_split is None):
        out = torch.empty((nv, cout), dtype=torch.float32, device=dev)        # want_f32=False: only the split planes are written
    xh, xl = x_split if x_split is not None else (None, None)
    yh, yl = out_split if out_split is not None else (None, None)
    # interleaved rows ([K step][hi 32 | lo 32], ONE tensor of 2 x channels halfs per row): given as (tensor, None)
    plane_flags = (1 if (xh is not None and xl is None) else 0) | (2 if (yh is not None and yl is None) else 0) | \
                  (4 if (rh is not None and rl is None) else 0) | (8 if fp32_partials else 0) | (16 if dense_single_offset else 0)
    check(lib.gp_sparse_conv_f16x3(_ptr(x), x.stride(0) if x is not None else 0, _ptr(xh), _ptr(xl),
                                   xh.stride(0) if xh is not None else 0, _ptr(pairs.pair_in), _ptr(pairs.pair_pos),
                                   _ptr(pairs.pair_off), _ptr(pairs.tile_start), _ptr(pairs.tile_desc), pairs.nseg, pairs.num_pairs, nv, kv, _ptr(w_hi), _ptr(w_lo), cin, cout,
                                   _ptr(partial), _ptr(scale), _ptr(shift), _ptr(residual),
                                   residual.stride(0) if residual is not None else 0, int(bool(relu)), _ptr(out),
                                   out.stride(0) if out is not None else 0, _ptr(yh), _ptr(yl), yh.stride(0) if yh is not None else 0,
                                   int(pairs.num_chunks), pairs.chunk_row_off, pairs.chunk_tile_off, pairs.chunk_pair_off,
                                   _ptr(x_row_inv), _ptr(out_row_inv), _ptr(rh), _ptr(rl), rh.stride(0) if rh is not None else 0, _ptr(ri),
                                   w_blocked, plane_flags, _stream()),
          "gp_sparse_conv_f16x3")
    return out


def interleave_planes(hi, lo):
    """(hi, lo) f16 [n, c] -> one f16 [n, 2 c] of interleaved rows, per 32-channel step [hi 32 | lo 32] (torch ops: tests, tools)"""
    n, c = hi.shape
    return torch.stack([hi.reshape(n, c // 32, 32), lo.reshape(n, c // 32, 32)], dim=2).reshape(n, 2 * c).contiguous()


def deinterleave_planes(t):
    """the inverse of interleave_planes"""
    n, c2 = t.shape
    v = t.reshape(n, c2 // 64, 2, 32)
    return v[:, :, 0].reshape(n, c2 // 2).contiguous(), v[:, :, 1].reshape(n, c2 // 2).contiguous()


def l2norm_rows_(x, d=None):
    lib = _lib.load()
    d = x.shape[1] if d is None else d
    check(lib.gp_l2norm_rows(_ptr(x), x.stride(0), int(d), x.shape[0], _stream()), "gp_l2norm_rows")
    return x


def embed_head_f16x3(x_split, w_hi, w_lo, out_scale, x_row_inv=None, normalize=True, out=None, planes=False, want_f32=True, plane_rows=None):
    """The student's 1x1x1 output layer on pre-split rows, fused with F.normalize (affinity_module.py:66,71,1547).
    x_split = (hi, lo) f16 [nv, >=cin]; w_hi / w_lo f16 [1, cout, cin] or [cout, cin] (conv_weights_split of the [1, cin, cout]
    kernel with a power-of-two pre-scale whose inverse is out_scale).
    planes=True: also returns the rows x 2^10 as f16 (hi, lo) planes -- the operand of affinity_cs_fragments -- written by the same
    epilogue; want_f32=False then skips the fp32 rows.  plane_rows (i32 [nv]): the plane row of input row r (rcb_order's rho).
    Returns out, or (out_or_None, (e_hi, e_lo))."""
    lib = _lib.load()
    hi, lo = x_split
    cout, cin = w_hi.shape[-2:]
    nv = hi.shape[0]
    _chk(hi, torch.float16, "x_hi"), _chk(lo, torch.float16, "x_lo")
    if hi.stride(0) != lo.stride(0) or hi.shape[1] < cin:
        raise ValueError("embed_head_f16x3: the hi / lo planes must share a row stride and hold cin channels")
    if out is None and (want_f32 or not planes):
        out = torch.empty((nv, cout), dtype=torch.float32, device=hi.device)
    eh = el = None
    if planes:
        eh = torch.empty((nv, cout), dtype=torch.float16, device=hi.device)
        el = torch.empty((nv, cout), dtype=torch.float16, device=hi.device)
    check(lib.gp_embed_head_f16x3(_ptr(hi), _ptr(lo), hi.stride(0), _ptr(x_row_inv), _ptr(w_hi), _ptr(w_lo), nv, int(cin), int(cout),
                                  float(out_scale), int(bool(normalize)), _ptr(out), out.stride(0) if out is not None else 0, _ptr(eh), _ptr(el),
                                  AFFINITY_PLANE_SCALE, _ptr(plane_rows) if planes else None, _stream()), "gp_embed_head_f16x3")
    return (out, (eh, el)) if planes else out


def knn_lattice(grid, coords_sorted, ids, k):
    lib = _lib.load()
    nv = coords_sorted.shape[0]
    dev = coords_sorted.device
    ws = _ws(lib.gp_knn_workspace_bytes(nv), dev)
    nbr = torch.empty((nv, k), dtype=torch.int32, device=dev)
    check(lib.gp_knn_lattice(_ptr(grid.buf), _ptr(coords_sorted), _ptr(ids), nv, int(k), _ptr(nbr), _ptr(ws),
                             ws.numel(), _stream()), "gp_knn_lattice")
    return nbr


def affinity_softmax(e, nbr, sharpen=20.0, d=None, into=None):
    """into: a PoolCs whose structure is built (pool_cs_plan(structure=True)): the weights also go straight into its fragment
    arrays -- the operator is complete when this returns (no pool_cs_fill)."""
    lib = _lib.load()
    nv, k = nbr.shape
    d = e.shape[1] if d is None else d
    w = torch.empty((nv, k), dtype=torch.float32, device=e.device)
    if into is not None:
        if into.dst is None or into.nv != nv or tuple(into.dst.shape) != (nv, k):
            raise ValueError("affinity_softmax: `into` needs a PoolCs structure built from these neighbour lists")
        check(lib.gp_affinity_softmax_scatter(_ptr(e), e.stride(0), int(d), _ptr(nbr), int(k), nv, float(sharpen), _ptr(w),
                                              _ptr(into.dst), _ptr(into.wa_hi), _ptr(into.wa_lo), _stream()),
              "gp_affinity_softmax_scatter")
        into.filled = True
        return w
    check(lib.gp_affinity_softmax(_ptr(e), e.stride(0), int(d), _ptr(nbr), int(k), nv, float(sharpen), _ptr(w),
                                  _stream()), "gp_affinity_softmax")
    return w


_E_SCALE = {}
AFFINITY_PLANE_SCALE = 1024.0      # the embedding planes of affinity_cs_fragments carry e x 2^10 (lo halves stay normal numbers)


def affinity_cs_fragments(e, sharpen, op, planes=None):
    """Row 11 + the operator fill in one matrix-core kernel (gp_affinity_cs_fragments): e fp32 [nv, 128] unit rows -- or planes = their
    (hi, lo) f16 planes x 2^10 as embed_head_f16x3(planes=True) writes them -- and op a PoolCs whose structure was built with
    pool_cs_plan(structure="valid").  Completes op (no [nv, k] weight matrix is produced)."""
    lib = _lib.load()
    if op.valid is None:
        raise ValueError("affinity_cs_fragments: the operator needs pool_cs_plan(structure='valid')")
    if planes is not None:
        eh, el = planes
    else:
        key = str(e.device)
        if key not in _E_SCALE:
            _E_SCALE[key] = torch.tensor([AFFINITY_PLANE_SCALE], dtype=torch.float32, device=e.device)
        eh, el = split_f16(e, 128, scale=_E_SCALE[key])
    check(lib.gp_affinity_cs_fragments(_ptr(eh), _ptr(el), op.nv, 128, int(op.k), float(sharpen), _ptr(op.bu_off), _ptr(op.bu_row),
                                       _ptr(op.bu_mask), _ptr(op.valid), int(op.block_rows), _ptr(op.wa_hi), _ptr(op.wa_lo), _stream()),
          "gp_affinity_cs_fragments")
    op.filled = True
    return op


def pool_ell(x, nbr, w, d, out):
    lib = _lib.load()
    nv, k = nbr.shape
    check(lib.gp_pool_ell(_ptr(x), x.stride(0), _ptr(nbr), _ptr(w), int(k), nv, int(d), _ptr(out), out.stride(0),
                          _stream()), "gp_pool_ell")
    return out


class PoolTiles:
    """Affinity operator re-blocked into tiles of r rows (built once per scene, applied many times)."""

    def __init__(self, tile_off, u_row, u_w, r, nv, total):
        self.tile_off, self.u_row, self.u_w, self.r, self.nv, self.total = tile_off, u_row, u_w, r, nv, total


def pool_tiles_build(nbr, w, r=16):
    """One host sync (total union entries, to size the tile arrays)."""
    lib = _lib.load()
    nv, k = nbr.shape
    dev = nbr.device
    nt = (nv + r - 1) // r
    ws = _ws(lib.gp_pool_tiles_workspace_bytes(nv, r), dev)
    off = torch.empty(nt + 1, dtype=torch.int64, device=dev)
    check(lib.gp_pool_tiles_count(_ptr(nbr), nv, int(k), int(r), _ptr(off), _ptr(ws), ws.numel(), _stream()),
          "gp_pool_tiles_count")
    total = int(off[nt].item())
    u_row = torch.empty(max(total, 1), dtype=torch.int32, device=dev)
    u_w = torch.empty((max(total, 1), r), dtype=torch.float32, device=dev)
    check(lib.gp_pool_tiles_fill(_ptr(nbr), _ptr(w), nv, int(k), int(r), _ptr(off), _ptr(u_row), _ptr(u_w), _stream()),
          "gp_pool_tiles_fill")
    return PoolTiles(off, u_row, u_w, r, nv, total)


def pool_tiles_apply(x, tiles, d, out):
    lib = _lib.load()
    check(lib.gp_pool_tiles_apply(_ptr(x), x.stride(0), _ptr(tiles.tile_off), _ptr(tiles.u_row), _ptr(tiles.u_w),
                                  int(tiles.r), tiles.nv, int(d), _ptr(out), out.stride(0), _stream()),
          "gp_pool_tiles_apply")
    return out


class PoolMfma:
    """Affinity operator in matrix-core form: per block of block_rows rows the padded neighbour union and the
    dense [block_rows x union] weight block, pre-split to f16 hi/lo in MFMA A-fragment order."""

    def __init__(self, bu_off, bu_n, bu_row, wa_hi, wa_lo, nv, total, block_rows, min_steps=0):
        self.bu_off, self.bu_n, self.bu_row, self.wa_hi, self.wa_lo = bu_off, bu_n, bu_row, wa_hi, wa_lo
        self.nv, self.total, self.block_rows, self.min_steps = nv, total, block_rows, min_steps
        self._queue = None

    @property
    def queue(self):
        """Tile counters of the persistent kernel (9 x uint32, zero between launches)."""
        if self._queue is None:
            self._queue = torch.zeros(16, dtype=torch.int32, device=self.bu_off.device)
        return self._queue

    @property
    def rows_padded(self):
        """Rows the persistent kernel's output buffers must hold (whole row blocks)."""
        return (self.nv + self.block_rows - 1) // self.block_rows * self.block_rows


def pool_mfma_build(nbr, w, block_rows=64, min_steps=0):
    """One host sync (total padded union rows, to size the arrays).  min_steps: pad every row block to at least this many
    32-row steps (9 for pool_mfma_apply_persistent, else 0)."""
    lib = _lib.load()
    nv, k = nbr.shape
    dev = nbr.device
    nb = (nv + block_rows - 1) // block_rows
    ws = _ws(lib.gp_pool_mfma_workspace_bytes(nv, block_rows), dev)
    bu_off = torch.empty(nb + 1, dtype=torch.int64, device=dev)
    bu_n = torch.empty(nb, dtype=torch.int32, device=dev)
    check(lib.gp_pool_mfma_count(_ptr(nbr), nv, int(k), int(block_rows), int(min_steps), _ptr(bu_off), _ptr(bu_n), _ptr(ws),
                                 ws.numel(), _stream()), "gp_pool_mfma_count")
    total, min_rows = torch.stack([bu_off[nb], torch.diff(bu_off).min()]).cpu().tolist()     # the one host sync
    bu_row = torch.empty(total, dtype=torch.int32, device=dev)
    wa_hi = torch.empty(total // 32 * (block_rows // 16) * 64 * 8, dtype=torch.float16, device=dev)
    wa_lo = torch.empty_like(wa_hi)
    check(lib.gp_pool_mfma_fill(_ptr(nbr), _ptr(w), nv, int(k), int(block_rows), _ptr(bu_off), _ptr(bu_n), total,
                                _ptr(bu_row), _ptr(wa_hi), _ptr(wa_lo), _stream()), "gp_pool_mfma_fill")
    return PoolMfma(bu_off, bu_n, bu_row, wa_hi, wa_lo, nv, total, block_rows, min_steps=min_rows // 32)


def pool_mfma_apply(x_split, op, d, out_split=None, out_f32=None, out_scale=None):
    """x_split / out_split: (hi, lo) f16 [Nv, >=d] pairs; out_f32 fp32 [Nv, >=d]; at least one output.
    out_scale: device scalar multiplied into out_f32 (1/s of a pow2_scale()-scaled x_split)."""
    lib = _lib.load()
    xh, xl = x_split
    assert xh.stride(0) == xl.stride(0)
    yh, yl = out_split if out_split is not None else (None, None)
    check(lib.gp_pool_mfma_apply(_ptr(xh), _ptr(xl), xh.stride(0), _ptr(op.bu_off), _ptr(op.bu_row), _ptr(op.wa_hi),
                                 _ptr(op.wa_lo), op.nv, int(d), int(op.block_rows), _ptr(yh), _ptr(yl),
                                 yh.stride(0) if yh is not None else 0, _ptr(out_f32),
                                 out_f32.stride(0) if out_f32 is not None else 0, _ptr(out_scale), _stream()), "gp_pool_mfma_apply")
    return out_f32 if out_f32 is not None else out_split


class PoolCs:
    """Column-sliced matrix-core pooling operator (gp_pool_cs_*): blocks of up to 128 rows, fragment masks."""

    def __init__(self, bu_off, bu_n, bu_row, bu_mask, wa_hi, wa_lo, nv, total, block_rows=128):
        self.bu_off, self.bu_n, self.bu_row, self.bu_mask = bu_off, bu_n, bu_row, bu_mask
        self.wa_hi, self.wa_lo, self.nv, self.total = wa_hi, wa_lo, nv, total
        self.block_rows = block_rows
        self.dst = None            # i32 [nv, k]: fragment element of (row, neighbour) once the structure is built ahead
        self.max_union = 0         # largest block union (pass 1): sizes the LDS tables of the fill passes (0 = not known)
        self.valid = None          # u32 validity words once the structure is built for affinity_cs_fragments; k: its list length
        self.k = 0
        self.dep = self.flags = None   # the chained launch's dependency lists and flags (pool_cs_deps)
        self.epoch = 0
        self.filled = False        # the weights are in the fragments


def pool_cs_plan(nbr, rows_per_block=128, structure=False):
    """First half of the operator build: needs the neighbour lists only (not the weights), so a scheduler can run it -- and its
    one host sync (total padded union rows, to size the arrays) -- before the affinity weights exist.  Returns a PoolCs without
    weights; pool_cs_fill completes it.  rows_per_block: 16..128 (every height from 96 up measures within 4 % on the S scene,
    profiles/r04_pool_block_height.log; 128 is the default)."""
    lib = _lib.load()
    nv, k = nbr.shape
    dev = nbr.device
    rpb = int(rows_per_block)
    nb = (nv + rpb - 1) // rpb
    ws = _ws(lib.gp_pool_cs_workspace_bytes(nv, rpb), dev)
    bu_off = torch.empty(nb + 2, dtype=torch.int64, device=dev)[:nb + 1]                       # (+ one slot behind it: the largest union)
    bu_n = torch.empty(nb, dtype=torch.int32, device=dev)
    tail = bu_off.as_strided((2,), (1,), bu_off.storage_offset() + nb)                        # [total, largest union]: ONE read-back
    check(lib.gp_pool_cs_count(_ptr(nbr), nv, int(k), rpb, _ptr(bu_off), _ptr(bu_n), tail[1:].data_ptr(), _ptr(ws), ws.numel(), _stream()),
          "gp_pool_cs_count")
    total, max_union = (int(v) for v in readback(tail))                                       # the one host sync
    bu_row = torch.empty(total, dtype=torch.int32, device=dev)
    bu_mask = torch.empty(total // 32, dtype=torch.int32, device=dev)
    wa_hi = torch.empty(total // 32 * 8 * 512, dtype=torch.float16, device=dev)
    wa_lo = torch.empty_like(wa_hi)
    op = PoolCs(bu_off, bu_n, bu_row, bu_mask, wa_hi, wa_lo, nv, total, block_rows=rpb)
    op.max_union = max_union                                   # sizes the LDS tables of the fill passes
    if structure == "valid":
        # the structure for affinity_cs_fragments: validity bits instead of the dst table, no fragment zeroed (that kernel writes
        # every non-empty fragment whole)
        op.valid = torch.empty(total // 32 * 128 + 64, dtype=torch.int32, device=dev)
        check(lib.gp_pool_cs_structure_valid(_ptr(nbr), nv, int(k), rpb, _ptr(bu_off), total, max_union, _ptr(bu_row), _ptr(bu_mask),
                                             _ptr(op.valid), _stream()), "gp_pool_cs_structure_valid")
        op.k = int(k)
        return op
    if structure and total * 128 < 2 ** 31:
        # everything of the fill pass that needs the lists only, + where each (row, neighbour) weight goes: affinity_softmax(into=op)
        # then completes the operator
        op.dst = torch.empty((nv, k), dtype=torch.int32, device=dev)
        check(lib.gp_pool_cs_structure(_ptr(nbr), nv, int(k), rpb, _ptr(bu_off), total, max_union, _ptr(bu_row), _ptr(bu_mask), _ptr(wa_hi),
                                       _ptr(wa_lo), _ptr(op.dst), _stream()), "gp_pool_cs_structure")
    return op


def pool_cs_fill(op, nbr, w):
    """Second half: union rows, fragment masks and the weights in fragment order (no host sync)."""
    lib = _lib.load()
    nv, k = nbr.shape
    check(lib.gp_pool_cs_fill(_ptr(nbr), _ptr(w), nv, int(k), int(op.block_rows), _ptr(op.bu_off), op.total, int(op.max_union), _ptr(op.bu_row),
                              _ptr(op.bu_mask), _ptr(op.wa_hi), _ptr(op.wa_lo), _stream()), "gp_pool_cs_fill")
    op.filled = True
    return op


def pool_cs_build(nbr, w, rows_per_block=128):
    """pool_cs_plan + pool_cs_fill."""
    return pool_cs_fill(pool_cs_plan(nbr, rows_per_block), nbr, w)


def pool_cs_apply(x_split, op, d, out_split=None, out_f32=None, out_scale=None, engine=False):
    """x_split / out_split: (hi, lo) f16 [Nv, >=d] pairs; out_f32 fp32 [Nv, >=d]; at least one output.
    out_scale: device scalar multiplied into out_f32 (1/s of a pow2_scale()-scaled x_split).
    engine: the producer / consumer form of the kernel (gp_pool_cs_apply_engine; same results)."""
    lib = _lib.load()
    xh, xl = x_split
    assert xh.stride(0) == xl.stride(0)
    yh, yl = out_split if out_split is not None else (None, None)
    fn, name = (lib.gp_pool_cs_apply_engine, "gp_pool_cs_apply_engine") if engine else (lib.gp_pool_cs_apply, "gp_pool_cs_apply")
    check(fn(_ptr(xh), _ptr(xl), xh.stride(0), _ptr(op.bu_off), _ptr(op.bu_row), _ptr(op.bu_mask), _ptr(op.wa_hi), _ptr(op.wa_lo),
             op.nv, int(d), int(op.block_rows), _ptr(yh), _ptr(yl), yh.stride(0) if yh is not None else 0,
             _ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0, _ptr(out_scale), _stream()), name)
    return out_f32 if out_f32 is not None else out_split


def pool_cs_apply_half(x_split, op, d, half, out_split=None, out_f32=None, out_scale=None):
    """One 256-column half (0 / 1) of pool_cs_apply: the halves are independent chains."""
    lib = _lib.load()
    xh, xl = x_split
    yh, yl = out_split if out_split is not None else (None, None)
    check(lib.gp_pool_cs_apply_half(_ptr(xh), _ptr(xl), xh.stride(0), _ptr(op.bu_off), _ptr(op.bu_row), _ptr(op.bu_mask), _ptr(op.wa_hi),
                                    _ptr(op.wa_lo), op.nv, int(d), int(op.block_rows), int(half), _ptr(yh), _ptr(yl),
                                    yh.stride(0) if yh is not None else 0, _ptr(out_f32), out_f32.stride(0) if out_f32 is not None else 0,
                                    _ptr(out_scale), _stream()), "gp_pool_cs_apply_half")


def pool_cs_deps(op):
    """Dependency lists + flags of the chained launch (pool_cs_apply_chain); needs the operator's structure (bu_row) only."""
    lib = _lib.load()
    dev = op.bu_row.device
    nb = op.bu_off.numel() - 1
    op.dep = torch.empty(nb * 64, dtype=torch.int32, device=dev)
    scratch = torch.empty(nb, dtype=torch.int32, device=dev)
    check(lib.gp_pool_cs_deps(_ptr(op.bu_off), _ptr(op.bu_row), op.nv, int(op.block_rows), _ptr(op.dep), _ptr(scratch), _stream()),
          "gp_pool_cs_deps")
    op.flags = torch.zeros(lib.gp_pool_cs_chain_flag_words(op.nv, int(op.block_rows)), dtype=torch.int32, device=dev)
    op.epoch = 0
    return op


def pool_cs_apply_chain(x_split, pong, op, d, applications, out_f32, out_scale=None):
    """All `applications` (>= 2) of the operator in ONE launch (gp_pool_cs_apply_chain): the same planes and bits as that many
    pool_cs_apply calls ping-ponging between x_split and pong; x_split is rewritten.  op.flags[0] != 0 afterwards (read at the
    caller's next synchronisation point: pool_cs_chain_check) means the launch gave up and the outputs are invalid."""
    lib = _lib.load()
    if getattr(op, "dep", None) is None:
        pool_cs_deps(op)
    xh, xl = x_split
    ph, pl_ = pong
    assert xh.stride(0) == xl.stride(0) == ph.stride(0) == pl_.stride(0)
    check(lib.gp_pool_cs_apply_chain(_ptr(xh), _ptr(xl), _ptr(ph), _ptr(pl_), xh.stride(0), _ptr(op.bu_off), _ptr(op.bu_row),
                                     _ptr(op.bu_mask), _ptr(op.wa_hi), _ptr(op.wa_lo), op.nv, int(d), int(op.block_rows),
                                     int(applications), _ptr(out_f32), out_f32.stride(0), _ptr(out_scale), _ptr(op.dep), _ptr(op.flags),
                                     op.epoch & 0xFFFFFFFF, _stream()), "gp_pool_cs_apply_chain")
    op.epoch += int(applications)
    op.chain_done = torch.cuda.Event()                    # recorded on the LAUNCHING stream: pool_cs_chain_check waits for it
    op.chain_done.record(torch.cuda.current_stream(xh.device))
    return out_f32


def pool_cs_chain_check(op):
    """Host side of the chained launch's contract: wait for the op's last chained launch (the event recorded on the stream that
    launched it -- the caller may be on another stream) and raise if the abort word is set."""
    if getattr(op, "flags", None) is None:
        return
    ev = getattr(op, "chain_done", None)
    if ev is not None:
        ev.synchronize()
    if int(op.flags[0].item()) != 0:
        raise _lib.GeoPurifyHipError("gp_pool_cs_apply_chain: a workgroup waited 2 s for a dependency; the pooled features are invalid")


def pool_mfma_apply_persistent(x_split, op, d, out_split=None, out_f32=None, out_scale=None, dynamic=True):
    """Persistent matrix-core pooling (one workgroup per CU).  Outputs need op.rows_padded rows; exactly one output form.
    dynamic: workgroups claim tiles from op.queue (False: static tile lists, the slower tuning reference)."""
    lib = _lib.load()
    xh, xl = x_split
    assert xh.stride(0) == xl.stride(0)
    yh, yl = out_split if out_split is not None else (None, None)
    rows = (yh if yh is not None else out_f32).shape[0]
    check(lib.gp_pool_mfma_apply_persistent(_ptr(xh), _ptr(xl), xh.stride(0), _ptr(op.bu_off), _ptr(op.bu_row), _ptr(op.wa_hi),
                                            _ptr(op.wa_lo), op.nv, int(d), int(op.block_rows), int(op.min_steps), _ptr(yh), _ptr(yl),
                                            yh.stride(0) if yh is not None else 0, _ptr(out_f32),
                                            out_f32.stride(0) if out_f32 is not None else 0, int(rows), _ptr(out_scale),
                                            _ptr(op.queue) if dynamic else None, _stream()),
          "gp_pool_mfma_apply_persistent")
    return out_f32 if out_f32 is not None else out_split


# ------------------------------------------------------------------------------------------ rows 5-7
def lift_dense_accum(feat2d, pt, x, y, sum_, cnt):
    lib = _lib.load()
    d, H, W = feat2d.shape
    check(lib.gp_lift_dense_accum(_ptr(feat2d), d, H, W, _ptr(pt), _ptr(x), _ptr(y), pt.shape[0], _ptr(sum_),
                                  sum_.stride(0), _ptr(cnt), _stream()), "gp_lift_dense_accum")


def lift_dense_bilinear_accum(feat_lo, out_h, out_w, pt, x, y, sum_, cnt):
    """LSeg path: feat_lo fp32 [D,h,w]; (x, y) index the [out_h, out_w] image the reference resizes the map to."""
    lib = _lib.load()
    d, h, w = feat_lo.shape
    check(lib.gp_lift_dense_bilinear_accum(_ptr(feat_lo), d, h, w, int(out_h), int(out_w), _ptr(pt), _ptr(x), _ptr(y),
                                           pt.shape[0], _ptr(sum_), sum_.stride(0), _ptr(cnt), _stream()),
          "gp_lift_dense_bilinear_accum")


def lift_dense_finish(sum_, d, cnt):
    lib = _lib.load()
    n = sum_.shape[0]
    seen = torch.empty(n, dtype=torch.uint8, device=sum_.device)
    check(lib.gp_lift_dense_finish(_ptr(sum_), sum_.stride(0), int(d), _ptr(cnt), n, _ptr(seen), _stream()),
          "gp_lift_dense_finish")
    return seen


def lift_masks_view(pred_masks, scores, taps, out_hw, x, y, want_logit=False, workspace=None):
    """pred_masks fp32 [Q,h,w]; scores fp32 [Q]; taps = (x0 i32 [W], wx f32 [W,4], y0 i32 [H], wy f32 [H,4])."""
    lib = _lib.load()
    Q, h, w = pred_masks.shape
    n_v = x.shape[0]
    dev = pred_masks.device
    need = lib.gp_lift_masks_workspace_bytes(Q, h, w)
    if workspace is None or workspace.numel() < need:
        workspace = _ws(need, dev)
    seg = torch.empty(n_v, dtype=torch.int32, device=dev)
    lg = torch.empty(n_v, dtype=torch.float32, device=dev) if want_logit else None
    tx0, twx, ty0, twy = taps
    check(lib.gp_lift_masks_view(_ptr(pred_masks), Q, h, w, _ptr(scores), _ptr(tx0), _ptr(twx), _ptr(ty0), _ptr(twy),
                                 int(out_hw[0]), int(out_hw[1]), _ptr(x), _ptr(y), n_v, _ptr(seg), _ptr(lg),
                                 _ptr(workspace), workspace.numel(), _stream()), "gp_lift_masks_view")
    return (seg, lg) if want_logit else seg


def segment_tables(mask_embed, text_norm, logit_scale, f_seg, logit_seg):
    lib = _lib.load()
    Q, d = mask_embed.shape
    C = text_norm.shape[0]
    check(lib.gp_segment_tables(_ptr(mask_embed), Q, d, _ptr(text_norm), C, float(logit_scale), _ptr(f_seg),
                                _ptr(logit_seg), _stream()), "gp_segment_tables")


def pv_count(pt, cnt):
    lib = _lib.load()
    check(lib.gp_pv_count(_ptr(pt), pt.shape[0], _ptr(cnt), _stream()), "gp_pv_count")


def exclusive_scan_i64(x):
    lib = _lib.load()
    n = x.shape[0]
    out = torch.empty_like(x)
    ws = _ws(lib.gp_scan_workspace_bytes(n), x.device)
    check(lib.gp_exclusive_scan_i64(_ptr(x), n, _ptr(out), _ptr(ws), ws.numel(), _stream()), "gp_exclusive_scan_i64")
    return out


def pv_fill(pt, seg, view, pv_start, cursor, pv_view, pv_seg):
    lib = _lib.load()
    check(lib.gp_pv_fill(_ptr(pt), _ptr(seg), pt.shape[0], int(view), _ptr(pv_start), _ptr(cursor), _ptr(pv_view),
                         _ptr(pv_seg), _stream()), "gp_pv_fill")


def fuse_views_top3(pv_start, pv_view, pv_seg, n, f_seg, logit_seg, out):
    lib = _lib.load()
    V, Q, d = f_seg.shape
    C = logit_seg.shape[2]
    seen = torch.empty(n, dtype=torch.uint8, device=out.device)
    check(lib.gp_fuse_views_top3(_ptr(pv_start), _ptr(pv_view), _ptr(pv_seg), int(n), _ptr(f_seg), _ptr(logit_seg),
                                 Q, d, C, _ptr(out), out.stride(0), _ptr(seen), _stream()), "gp_fuse_views_top3")
    return seen


def nn1(ref_xyz, query_xyz):
    """Exact nearest reference index per query (fp32 coords, fp64 distances)."""
    lib = _lib.load()
    _chk(ref_xyz, torch.float32, "ref_xyz")
    _chk(query_xyz, torch.float32, "query_xyz")
    nr, nq = ref_xyz.shape[0], query_xyz.shape[0]
    nn = torch.empty(nq, dtype=torch.int64, device=ref_xyz.device)
    if nq == 0:
        return nn
    ws = _ws(lib.gp_nn1_workspace_bytes(nr, nq), ref_xyz.device)
    check(lib.gp_nn1_f64(_ptr(ref_xyz), nr, _ptr(query_xyz), nq, _ptr(nn), _ptr(ws), ws.numel(), _stream()), "gp_nn1_f64")
    return nn


def nn1_masked(xyz, ref_mask, query_mask, workspace=None):
    """nn[i] = nearest point with ref_mask among xyz for every i with query_mask, else -1.  No sync."""
    lib = _lib.load()
    _chk(xyz, torch.float32, "xyz")
    n = xyz.shape[0]
    nn = torch.empty(n, dtype=torch.int64, device=xyz.device)
    need = lib.gp_nn1_masked_workspace_bytes(n)
    if workspace is None or workspace.numel() < need:
        workspace = _ws(need, xyz.device)
    check(lib.gp_nn1_masked_f64(_ptr(xyz), n, _ptr(ref_mask), _ptr(query_mask), _ptr(nn), _ptr(workspace),
                                workspace.numel(), _stream()), "gp_nn1_masked_f64")
    return nn


def visible_lists(mapping, pt, x, y, count_dev, workspace=None):
    lib = _lib.load()
    n = mapping.shape[0]
    need = lib.gp_visible_lists_workspace_bytes(n)
    if workspace is None or workspace.numel() < need:
        workspace = _ws(need, mapping.device)
    check(lib.gp_visible_lists(_ptr(mapping), n, _ptr(pt), _ptr(x), _ptr(y), _ptr(count_dev), _ptr(workspace),
                               workspace.numel(), _stream()), "gp_visible_lists")


def views_visible_lists(coords, params, depth_all, width, height, cut_bound, vis_thres, min_visible, val_keep):
    """All views of a scene at once: coords f64 [N,3]; params f64 [V,20] (device) = world->camera (16) | fx fy cx cy;
    depth_all f64 [V,H,W] or None.  Returns the view-major entry arrays (sized V*N; view_off[V] = entries used)."""
    lib = _lib.load()
    _chk(coords, torch.float64, "coords")
    _chk(params, torch.float64, "params")
    n, nv = coords.shape[0], params.shape[0]
    dev = coords.device
    if depth_all is not None:
        _chk(depth_all, torch.float64, "depth_all")
        assert depth_all.shape == (nv, height, width)
    ent = {"pt": torch.empty(nv * n, dtype=torch.int64, device=dev), "x": torch.empty(nv * n, dtype=torch.int64, device=dev),
           "y": torch.empty(nv * n, dtype=torch.int64, device=dev), "view": torch.empty(nv * n, dtype=torch.int32, device=dev),
           "view_off": torch.empty(nv + 1, dtype=torch.int64, device=dev), "keep": torch.empty(nv, dtype=torch.uint8, device=dev)}
    ws = _ws(lib.gp_views_visible_lists_workspace_bytes(n, nv), dev)
    check(lib.gp_views_visible_lists(_ptr(coords), n, _ptr(params), _ptr(depth_all), nv, int(width), int(height), int(cut_bound),
                                     float(vis_thres), int(min_visible), int(val_keep), _ptr(ent["pt"]), _ptr(ent["x"]), _ptr(ent["y"]),
                                     _ptr(ent["view"]), _ptr(ent["view_off"]), _ptr(ent["keep"]), _ptr(ws), ws.numel(), _stream()),
          "gp_views_visible_lists")
    return ent


def lift_masks_views(pred_masks, scores, taps, out_hw, xyz, ent, total, nviews, fill_cap=None):
    """Rows 6-7 up to the point -> (view, segment) lists for all views at once.  pred_masks f32 [Vsrc,Q,h,w], scores f32
    [Vsrc,Q], xyz f32 [N,3], ent = views_visible_lists(...), total = entries used.  Returns (seg, pv_start, pv_view, pv_seg).
    fill_cap: capacity of the in-view fill's partial arrays in fill queries (None: a quarter of the entries, at least 65 536 --
    the uncovered share of the visible pixels is a few per cent on every scene measured; more queries than that are still
    answered, by the kernel's overflow path)."""
    lib = _lib.load()
    _chk(pred_masks, torch.float32, "pred_masks")
    _chk(scores, torch.float32, "scores")
    _chk(xyz, torch.float32, "xyz")
    nsrc, Q, h, w = pred_masks.shape
    n = xyz.shape[0]
    dev = xyz.device
    seg = torch.empty(total, dtype=torch.int32, device=dev)
    start = torch.empty(n + 1, dtype=torch.int64, device=dev)
    pvv = torch.empty(total, dtype=torch.int32, device=dev)
    pvs = torch.empty(total, dtype=torch.int32, device=dev)
    if fill_cap is None:
        fill_cap = min(int(total), max(int(total) // 4, 65536))
    ws = _ws(lib.gp_lift_masks_views_workspace_bytes(nsrc, Q, h, w, total, n, int(fill_cap)), dev)
    tx0, twx, ty0, twy = taps
    check(lib.gp_lift_masks_views(_ptr(pred_masks), nsrc, Q, h, w, _ptr(scores), _ptr(tx0), _ptr(twx), _ptr(ty0), _ptr(twy),
                                  int(out_hw[0]), int(out_hw[1]), _ptr(xyz), n, _ptr(ent["pt"]), _ptr(ent["x"]), _ptr(ent["y"]),
                                  _ptr(ent["view"]), _ptr(ent["view_off"]), _ptr(ent["keep"]), int(nviews), int(total), int(fill_cap), _ptr(seg),
                                  _ptr(start), _ptr(pvv), _ptr(pvs), _ptr(ws), ws.numel(), _stream()), "gp_lift_masks_views")
    return seg, start, pvv, pvs


# ------------------------------------------------------------------------------------------ row 13
def classify_argmax(feat, text_norm, logit_scale, d=None):
    lib = _lib.load()
    n = feat.shape[0]
    d = feat.shape[1] if d is None else d
    C = text_norm.shape[0]
    pred = torch.empty(n, dtype=torch.int64, device=feat.device)
    zero = torch.empty(n, dtype=torch.uint8, device=feat.device)
    check(lib.gp_classify_argmax(_ptr(feat), feat.stride(0), int(d), n, _ptr(text_norm), C, float(logit_scale),
                                 _ptr(pred), _ptr(zero), _stream()), "gp_classify_argmax")
    return pred, zero


def classify_argmax_gemm(feat, text_norm, d=None):
    """Same decision as classify_argmax (arg-max_c <f/|f|, t_c> = arg-max_c <f, t_c>) with the logits from
    the exact-fp32 MFMA GEMM kernel -- for large class counts (Matterport-160 / ScanNet200)."""
    lib = _lib.load()
    n = feat.shape[0]
    d = feat.shape[1] if d is None else d
    C = text_norm.shape[0]
    cpad = (C + 127) // 128 * 128
    w = torch.zeros((d, cpad), dtype=torch.float32, device=feat.device)
    w[:, :C] = text_norm.t()
    logits = sparse_conv(feat, None, w)
    pred = torch.empty(n, dtype=torch.int64, device=feat.device)
    zero = torch.empty(n, dtype=torch.uint8, device=feat.device)
    check(lib.gp_rows_argmax(_ptr(logits), logits.stride(0), C, n, _ptr(feat), feat.stride(0), int(d), _ptr(pred),
                             _ptr(zero), _stream()), "gp_rows_argmax")
    return pred, zero


def iou_hist(pred, target, num_classes, ignore_ids, counts):
    lib = _lib.load()
    ig = (ctypes.c_int64 * max(len(ignore_ids), 1))(*[int(v) for v in ignore_ids])
    check(lib.gp_iou_hist_i64(_ptr(pred), _ptr(target), pred.shape[0], int(num_classes), ig, len(ignore_ids),
                              _ptr(counts), _stream()), "gp_iou_hist_i64")
    return counts


# ------------------------------------------------------------------------------------------ training step
def col_stats(y, c=None):
    """mean, biased variance of the rows of y fp32 [nv, >=c] (BatchNorm1d training statistics)."""
    lib = _lib.load()
    nv = y.shape[0]
    c = y.shape[1] if c is None else c
    ws = _ws(lib.gp_col_stats_workspace_bytes(nv, c), y.device)
    mean = torch.empty(c, dtype=torch.float32, device=y.device)
    var = torch.empty(c, dtype=torch.float32, device=y.device)
    check(lib.gp_col_stats(_ptr(y), y.stride(0), nv, int(c), _ptr(mean), _ptr(var), _ptr(ws), ws.numel(), _stream()), "gp_col_stats")
    return mean, var


def col_sums_f64(y, c=None, mean=None):
    """fp64 column sums of y fp32 [nv, >=c] (mean None) or of (y - mean)^2: the SyncBatchNorm reduction vectors."""
    lib = _lib.load()
    nv = y.shape[0]
    c = y.shape[1] if c is None else c
    ws = _ws(lib.gp_col_stats_workspace_bytes(nv, c), y.device)
    out = torch.empty(c, dtype=torch.float64, device=y.device)
    check(lib.gp_col_sums_f64(_ptr(y), y.stride(0), nv, int(c), _ptr(mean), _ptr(out), _ptr(ws), ws.numel(), _stream()), "gp_col_sums_f64")
    return out


def bn_bwd_sums_f64(dout, act, y, mean, var, eps, mask_affine=None):
    """fp64 [2c]: sum dz | sum dz * xhat over this rank's rows (dz = dout masked by act > 0, or -- act None, mask_affine = (gamma, beta) --
    by the sign of the normalised, affine y: see bn_train_backward)."""
    lib = _lib.load()
    nv, c = y.shape[0], mean.shape[0]
    ws = _ws(lib.gp_col_stats_workspace_bytes(nv, c), y.device)
    sums = torch.empty(2 * c, dtype=torch.float64, device=y.device)
    check(lib.gp_bn_bwd_sums_f64(_ptr(dout), dout.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(y), y.stride(0),
                                 _ptr(mean), _ptr(var), float(eps), _ptr(mask_affine[0]) if mask_affine else None,
                                 _ptr(mask_affine[1]) if mask_affine else None, nv, int(c), _ptr(sums), _ptr(ws), ws.numel(), _stream()),
          "gp_bn_bwd_sums_f64")
    return sums


def bn_bwd_apply(dout, act, y, mean, var, eps, gamma, sums_f32, n_total, want_dz=False, dy_scale2=None, beta_mask=None):
    """dy (and dz) of the BatchNorm backward pass from reduction vectors taken over n_total rows (all ranks).
    dy_scale2 (fp32 [2] device tensor): receives pow2_scale(dy) from the same sweep."""
    lib = _lib.load()
    nv, c = y.shape[0], mean.shape[0]
    dy = torch.empty((nv, c), dtype=torch.float32, device=y.device)
    dz = torch.empty((nv, c), dtype=torch.float32, device=y.device) if want_dz else None
    check(lib.gp_bn_bwd_apply(_ptr(dout), dout.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(y), y.stride(0),
                              _ptr(mean), _ptr(var), float(eps), _ptr(gamma), _ptr(beta_mask), _ptr(sums_f32), int(n_total), nv, int(c), _ptr(dy), dy.stride(0),
                              _ptr(dz), dz.stride(0) if dz is not None else 0, _ptr(dy_scale2), _stream()), "gp_bn_bwd_apply")
    return (dy, dz) if want_dz else dy


def bn_train_apply(y, mean, var, gamma, beta, eps, residual=None, relu=True, want_split=False, momentum=0.1,
                   running_mean=None, running_var=None, want_f32=True):
    """(out fp32 | None, (hi, lo) | None).  want_f32=False (with want_split): only the split planes are written -- for a layer whose
    backward pass takes its ReLU mask from y (bn_train_backward(beta_mask=))."""
    lib = _lib.load()
    nv, c = y.shape[0], mean.shape[0]
    assert want_f32 or want_split
    out = torch.empty((nv, c), dtype=torch.float32, device=y.device) if want_f32 else None
    hi = lo = None
    if want_split:
        hi = torch.empty((nv, c), dtype=torch.float16, device=y.device)
        lo = torch.empty((nv, c), dtype=torch.float16, device=y.device)
    check(lib.gp_bn_train_apply(_ptr(y), y.stride(0), nv, int(c), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps),
                                _ptr(residual), residual.stride(0) if residual is not None else 0, int(bool(relu)), _ptr(out),
                                out.stride(0) if out is not None else 0, _ptr(hi), _ptr(lo), hi.stride(0) if hi is not None else 0, float(momentum),
                                _ptr(running_mean), _ptr(running_var), _stream()), "gp_bn_train_apply")
    return out, ((hi, lo) if want_split else None)


def bn_train_backward(dout, act, y, mean, var, eps, gamma, want_dz=False, dy_scale2=None, beta_mask=None, split=False):
    """Returns dy, dgamma, dbeta (, dz).  act: post-ReLU activation (mask) or None; act None and beta_mask = the layer's beta: the mask of a
    layer WITHOUT a residual recomputed from y ((y - mean) * invstd * gamma + beta > 0: the float the forward pass evaluated).
    dy_scale2 (fp32 [2] device tensor): receives pow2_scale(dy) from the sweep that writes dy.
    split=True (needs dy_scale2): dy comes back as ((hi, lo) f16 [nv + 1, c] with a zero last row) holding dy * dy_scale2[0] -- written by
    the sweep itself, the scale from a bound of max |dy| taken in the reduction pass (no fp32 dy, no split pass)."""
    lib = _lib.load()
    nv, c = y.shape[0], mean.shape[0]
    dev = y.device
    hi = lo = dy = None
    if split:
        assert dy_scale2 is not None
        hi = torch.empty((nv + 1, c), dtype=torch.float16, device=dev)
        lo = torch.empty((nv + 1, c), dtype=torch.float16, device=dev)
    else:
        dy = torch.empty((nv, c), dtype=torch.float32, device=dev)
    dz = torch.empty((nv, c), dtype=torch.float32, device=dev) if want_dz else None
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    ws = _ws(lib.gp_bn_train_backward_workspace_bytes(nv, int(c)), dev)
    check(lib.gp_bn_train_backward(_ptr(dout), dout.stride(0), _ptr(act), act.stride(0) if act is not None else 0, _ptr(y),
                                   y.stride(0), _ptr(mean), _ptr(var), float(eps), _ptr(gamma), _ptr(beta_mask), nv, int(c), _ptr(dy),
                                   dy.stride(0) if dy is not None else 0, _ptr(dz), dz.stride(0) if dz is not None else 0, _ptr(dgamma),
                                   _ptr(dbeta), _ptr(dy_scale2), _ptr(hi), _ptr(lo), hi.stride(0) if hi is not None else 0, _ptr(ws),
                                   ws.numel(), _stream()), "gp_bn_train_backward")
    dy = (hi, lo) if split else dy
    return (dy, dgamma, dbeta, dz) if want_dz else (dy, dgamma, dbeta)


def infonce_fwd_bwd(e, sample_to_voxel, point_to_batch, num_anchors, num_negatives, temperature):
    """loss (0-d device tensor) and d loss / d e of the InfoNCE over (anchor | positive | negatives) samples."""
    lib = _lib.load()
    nv, d = e.shape
    ns = sample_to_voxel.shape[0]
    _chk(sample_to_voxel, torch.int64, "sample_to_voxel")
    _chk(point_to_batch, torch.int64, "point_to_batch")
    assert point_to_batch.shape[0] == num_anchors * (2 + num_negatives)
    loss = torch.empty((), dtype=torch.float32, device=e.device)
    de = torch.empty((nv, d), dtype=torch.float32, device=e.device)
    ws = _ws(lib.gp_infonce_workspace_bytes(ns, d), e.device)
    check(lib.gp_infonce_fwd_bwd(_ptr(e), e.stride(0), nv, int(d), _ptr(sample_to_voxel), ns, _ptr(point_to_batch),
                                 int(num_anchors), int(num_negatives), float(temperature), _ptr(loss), _ptr(de), de.stride(0),
                                 _ptr(ws), ws.numel(), _stream()), "gp_infonce_fwd_bwd")
    return loss, de


def adamw_step_(param, grad, exp_avg, exp_avg_sq, lr, step, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8):
    lib = _lib.load()
    for t in (param, grad, exp_avg, exp_avg_sq):
        _chk(t, torch.float32, "adamw tensor")
        assert t.is_contiguous() and t.numel() == param.numel()
    check(lib.gp_adamw_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
                            float(betas[1]), float(eps), float(weight_decay), int(step), _stream()), "gp_adamw_step")
    return param


def knn_points(xyz, queries, k):
    """xyz fp32 [N,3]; queries i64 [A] row ids -> i64 [A,k] nearest other rows, (d^2, id) order."""
    lib = _lib.load()
    _chk(xyz, torch.float32, "xyz")
    _chk(queries, torch.int64, "queries")
    out = torch.empty((queries.shape[0], k), dtype=torch.int64, device=xyz.device)
    flag = torch.zeros(1, dtype=torch.int32, device=xyz.device)
    check(lib.gp_knn_points_f32(_ptr(xyz), xyz.shape[0], _ptr(queries), queries.shape[0], int(k), _ptr(out), _ptr(flag),
                                _stream()), "gp_knn_points_f32")
    return out, flag


def sampler_select(sim, anchor_indices, k, n=None):
    """positive i64 [A], macro i64 [A, k] of sim fp32 [A, >= n] (gp_sampler_select): the arg-max over the points other than the anchor
    and the k lowest other than anchor and positive, ascending by (value, index).  sim is not written."""
    lib = _lib.load()
    A = sim.shape[0]
    n = sim.shape[1] if n is None else n
    assert sim.stride(1) == 1 and anchor_indices.dtype == torch.int64 and anchor_indices.is_contiguous()
    positive = torch.empty(A, dtype=torch.int64, device=sim.device)
    macro = torch.empty((A, k), dtype=torch.int64, device=sim.device)
    check(lib.gp_sampler_select(_ptr(sim), sim.stride(0), A, int(n), _ptr(anchor_indices), int(k), _ptr(positive), _ptr(macro), _stream()),
          "gp_sampler_select")
    return positive, macro


def normalize_split_f16(x, n_pad=None, eps=1e-12):
    """(hi, lo) f16 [n_pad, d] of F.normalize(x, dim=1); rows beyond x's are zero (gp_normalize_split_f16)."""
    lib = _lib.load()
    n, d = x.shape
    n_pad = n if n_pad is None else n_pad
    hi = torch.empty((n_pad, d), dtype=torch.float16, device=x.device)
    lo = torch.empty((n_pad, d), dtype=torch.float16, device=x.device)
    check(lib.gp_normalize_split_f16(_ptr(x), x.stride(0), int(d), n, int(n_pad), float(eps), _ptr(hi), _ptr(lo), hi.stride(0), _stream()),
          "gp_normalize_split_f16")
    return hi, lo


class WgradPlan:
    """Pair lists of one voxel set in the layout of gp_conv_wgrad_f16x3 (shared by every 3x3x3 layer of a step)."""

    def __init__(self, pair_in, pair_out, segs, seg_off, num_segments, nv):
        self.pair_in, self.pair_out, self.segs, self.seg_off, self.num_segments, self.nv = pair_in, pair_out, segs, seg_off, num_segments, nv
        self.workspace = None


def wgrad_plan_build(offset_pairs, nv, steps_per_segment=128):
    """offset_pairs: list over the kernel offsets of (out_rows i64, in_rows i64) device tensors (host knows the sizes)."""
    dev = offset_pairs[0][0].device
    pin, pout, segs, seg_off = [], [], [], [0]
    step0 = 0
    for k, (out_rows, in_rows) in enumerate(offset_pairs):
        n = int(out_rows.numel())
        pad = (-n) % 32
        pin.append(in_rows.to(torch.int32))
        pout.append(out_rows.to(torch.int32))
        if pad:
            pin.append(torch.zeros(pad, dtype=torch.int32, device=dev))
            pout.append(torch.full((pad,), nv, dtype=torch.int32, device=dev))
        steps = (n + pad) // 32
        for s in range(0, steps, steps_per_segment):
            segs.append((k, step0 + s, min(steps_per_segment, steps - s), 0))
        seg_off.append(len(segs))
        step0 += steps
    return WgradPlan(torch.cat(pin).contiguous(), torch.cat(pout).contiguous(),
                     torch.tensor(segs, dtype=torch.int32, device=dev).contiguous(),
                     torch.tensor(seg_off, dtype=torch.int32, device=dev), len(segs), nv)


def kernel_map_pairs(nbr_map):
    """The (output row, input row) pairs of every kernel offset of nbr_map i32 [kv, nv] in offset-major, row-ascending order:
    (offset i64 [P], out_rows i64 [P], in_rows i64 [P], counts: host list of kv ints).  Two host syncs for the whole map (a per-offset
    torch.nonzero costs one each: 27 syncs with the GPU idle at the head of every training step)."""
    kv, nv = nbr_map.shape
    kk, rr = torch.nonzero(nbr_map >= 0, as_tuple=True)
    counts = readback(torch.bincount(kk, minlength=kv))
    in_rows = nbr_map.reshape(-1)[kk * nv + rr].long()
    return kk, rr, in_rows, [int(c) for c in counts]


def wgrad_plan_from_pairs(kk, out_rows, in_rows, counts, nv, steps_per_segment=128):
    """wgrad_plan_build for the pair arrays of kernel_map_pairs: same layout (every offset padded to a multiple of 32 pairs with
    (in 0, out nv)), built with a handful of device operations instead of four per offset."""
    dev = out_rows.device
    shift, segs, seg_off = [], [], [0]
    base = cum = step0 = 0
    for k, n in enumerate(counts):
        pad = (-n) % 32
        shift.append(base - cum)
        steps = (n + pad) // 32
        for s in range(0, steps, steps_per_segment):
            segs.append((k, step0 + s, min(steps_per_segment, steps - s), 0))
        seg_off.append(len(segs))
        step0 += steps
        base += n + pad
        cum += n
    pin = torch.zeros(max(base, 1), dtype=torch.int32, device=dev)
    pout = torch.full((max(base, 1),), nv, dtype=torch.int32, device=dev)
    if cum:
        dest = torch.arange(cum, device=dev) + torch.tensor(shift, dtype=torch.int64, device=dev)[kk]
        pin[dest] = in_rows.to(torch.int32)
        pout[dest] = out_rows.to(torch.int32)
    return WgradPlan(pin, pout, torch.tensor(segs, dtype=torch.int32, device=dev).reshape(-1, 4).contiguous(),
                     torch.tensor(seg_off, dtype=torch.int32, device=dev), len(segs), nv)


def conv_wgrad_f16x3(x_split, y_split, plan, cin_pad, cin_out, cout, inv_scale=None, out=None):
    """x_split (hi, lo) f16 [nv, >=cin_pad]; y_split (hi, lo) f16 [nv+1, >=cout] whose last row is zero.
    out: contiguous fp32 [kv, cin_out, cout] to receive dW (e.g. a slice of a gradient bucket: sharding.GradientBuckets.view)."""
    lib = _lib.load()
    xh, xl = x_split
    yh, yl = y_split
    assert yh.shape[0] == plan.nv + 1 and xh.shape[0] >= plan.nv
    need = lib.gp_conv_wgrad_workspace_bytes(plan.num_segments, int(cin_pad), int(cout))
    if plan.workspace is None or plan.workspace.numel() < need:
        plan.workspace = _ws(need, xh.device)
    kv = plan.seg_off.shape[0] - 1
    dw = out if out is not None else torch.empty((kv, cin_out, cout), dtype=torch.float32, device=xh.device)
    assert dw.shape == (kv, cin_out, cout) and dw.is_contiguous() and dw.dtype == torch.float32
    check(lib.gp_conv_wgrad_f16x3(_ptr(xh), _ptr(xl), xh.stride(0), _ptr(yh), _ptr(yl), yh.stride(0), _ptr(plan.pair_in),
                                  _ptr(plan.pair_out), _ptr(plan.segs), plan.num_segments, _ptr(plan.seg_off), kv, int(cin_pad),
                                  int(cin_out), int(cout), _ptr(inv_scale), _ptr(dw), _ptr(plan.workspace), plan.workspace.numel(),
                                  _stream()), "gp_conv_wgrad_f16x3")
    return dw


# ------------------------------------------------------------------------------------------ SURVEY 8(f)-3
def fused_decode(mask_chunk, feat, vox_ind, mode, row_keep=None):
    """Rows of a fused-feature file for the voxel representatives (gp_fused_decode; dataset/feature_loader.py:113-192).
    mask_chunk bool/u8 [N], feat [rows, D] (fp16 or fp32), vox_ind i64 [Nv], row_keep bool/u8 [rows] or None.
    mode 0: (feat rows of the kept voxels, compact [n_sel, D]; mask bool [Nv]) -- one host sync for n_sel;
    mode 1: (one row per voxel [Nv, D], zeros outside the chunk; mask bool [Nv])."""
    lib = _lib.load()
    dev = feat.device
    mc = mask_chunk.to(torch.uint8).contiguous()
    rk = row_keep.to(torch.uint8).contiguous() if row_keep is not None else None
    feat = feat.contiguous()
    vox_ind = vox_ind.to(torch.int64).contiguous()
    n, nv = mc.shape[0], vox_ind.shape[0]
    out = torch.empty((nv,) + tuple(feat.shape[1:]), dtype=feat.dtype, device=dev)
    mask_out = torch.empty(nv, dtype=torch.uint8, device=dev)
    n_sel = torch.zeros(2, dtype=torch.int64, device=dev)
    ws = _ws(lib.gp_fused_decode_workspace_bytes(n, nv), dev)
    row_bytes = feat[0].numel() * feat.element_size()
    check(lib.gp_fused_decode(_ptr(mc), n, _ptr(rk), _ptr(feat), feat.shape[0], row_bytes, _ptr(vox_ind), nv, int(mode), _ptr(out),
                              _ptr(mask_out), _ptr(n_sel), _ptr(ws), ws.numel(), _stream()), "gp_fused_decode")
    kept, in_chunk = (int(v) for v in n_sel.tolist())                # the one host sync (a loader item, off the scene's hot path)
    if in_chunk != feat.shape[0]:
        # the host formulation (reference: dataset/feature_loader.py:141-190) fails on such a file; so does the device path
        raise ValueError(f"fused-feature file: mask_full selects {in_chunk} points but feat holds {feat.shape[0]} rows")
    if mode == 0:
        out = out[:kept]
    return out, mask_out.bool()
