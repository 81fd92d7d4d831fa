"""Tensor-level wrappers over the C ABI (include/rcb.h) + autograd glue.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic of the
hot path runs in librcb_hip.so.  Every wrapper raises if the library is missing or a tensor is
not on the GPU -- there is no CPU fallback.
"""
import collections
import ctypes as C
import os as _os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import AdamCfg, AdamTensor, Level, LevelBwd, RcbError, SirenDesc, addr, check, ptr, stream_ptr

f32 = torch.float32
f64 = torch.float64
i32 = torch.int32


# ----------------------------------------------------------------------------------------------
# SIREN MLP (K3 + K4)
# ----------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class SirenMeta:
    """Static geometry of one batched-MLP launch."""
    samples: int
    n_pix: int
    fourier_dim: int
    pe_dim: int
    n_hidden: int
    hidden: int
    out_dim: int
    w0: float = 30.0
    precision: int = 0
    hidden_dims: Optional[tuple] = None      # per-layer hidden widths when they differ (fp32 mode only); hidden = their maximum

    @property
    def dims(self):
        hid = list(self.hidden_dims) if self.hidden_dims else [self.hidden] * self.n_hidden
        return [self.fourier_dim + self.pe_dim] + hid + [self.out_dim]

    @property
    def d_net(self):
        dims = self.dims
        return sum(dims[i + 1] * (dims[i] + 1) for i in range(len(dims) - 1))


def _xf_stride(xf, meta, n_inr):
    """xf is [P,F] (shared), or [N,P,F] possibly an expanded (stride-0) view."""
    P, F = meta.n_pix, meta.fourier_dim
    if xf.dim() == 2:
        if tuple(xf.shape) != (P, F) or not xf.is_contiguous():
            raise RcbError(f"xf must be contiguous [{P},{F}], got {tuple(xf.shape)}")
        return 0
    if tuple(xf.shape) != (n_inr, P, F):
        raise RcbError(f"xf must be [{n_inr},{P},{F}], got {tuple(xf.shape)}")
    if xf.stride(1) != F or xf.stride(2) != 1:
        raise RcbError("xf rows must be contiguous")
    if xf.stride(0) not in (0, P * F):
        raise RcbError("unsupported xf stride")
    return int(xf.stride(0))


# Rows of many pixels are cut into pieces of 32 .. 63 of the 32-pixel tiles, one workgroup each (rcb_siren_desc.pixel_chunks;
# partial weight gradients summed in a fixed order by rcb_siren_reduce_chunks).  The rule looks at the geometry only, never at
# the number of rows in the launch: an INR's gradient stays bit-identical whether it is trained alone or in a large batch
# (test_determinism_and_batch_invariance), and the patched presets (Kodak 4096, video 6144 pixels per patch: a few hundred rows
# for 256 CUs x 2 workgroups) fill the chip.
LONG_ROW_TILES = 64

# Opt-in on top of that: also split SHORTER rows when a launch has few of them.  Off by default because that rule does depend on
# the batch size (tools/siren_chunks.py on MI355X, 4096 pixels: 96 rows 85 -> 44 us with 4 chunks, 192 rows 90 -> 68 us).
PIXEL_CHUNKS_AUTO = False


def siren_pixel_chunks(G, meta: SirenMeta):
    """workgroups per row of wvec chosen for a launch of G rows (16-bit kernels): rows of at least LONG_ROW_TILES tiles are
    always cut into ntiles // 32 pieces; with PIXEL_CHUNKS_AUTO a width-32 launch of fewer than 256 shorter rows is split as
    well, every chunk keeping at least 8 tiles (2 per wave)"""
    ntiles = (meta.n_pix + 31) // 32
    if meta.precision == 0:
        return 1
    if ntiles >= LONG_ROW_TILES:
        return ntiles // 32
    if not PIXEL_CHUNKS_AUTO or meta.hidden != 32 or G >= 256 or ntiles < 16:
        return 1
    return max(1, min(4 if G <= 128 else 2, ntiles // 8))


_XF16_CACHE = collections.OrderedDict()


def xf_bf16(xf, precision=1):
    """16-bit copy of the coordinate features in the operand format of `precision` (1: bf16, 2: f16), rows zero-padded to a
    multiple of 8 features (rcb_siren_desc.xf_bf16); None for grids that are not shared ([P, F] or a stride-0 expansion of
    one).  A caller that records kernel launches (HIP graph) must OWN the copy it passes on (`xf16=` of the siren_* wrappers;
    the models keep it in their graph workspace): the cache below only serves eager calls and may drop its entries."""
    if xf.dim() == 3 and xf.stride(0) != 0:
        return None
    base = xf if xf.dim() == 2 else xf[0]
    F = base.shape[-1]
    out = torch.zeros(base.shape[0], (F + 7) // 8 * 8, dtype=bf16 if precision == 1 else torch.float16, device=base.device)
    out[:, :F] = base
    return out


def _xf_bf16_cached(xf, precision=1):
    """xf_bf16 through a small LRU cache for eager calls: the grid is constant for a whole run.  An entry keeps a
    reference to its source tensor (while it is held the allocator cannot hand the address to other data, so (address,
    shape, version counter) identifies the contents); the least recently used entry is dropped beyond 16 -- a copy is only
    ever referenced by the call that fetched it (`d._keep`), never by a recorded graph."""
    if xf.dim() == 3 and xf.stride(0) != 0:
        return None
    base = xf if xf.dim() == 2 else xf[0]
    key = (base.data_ptr(), tuple(base.shape), base._version, str(base.device), precision)
    hit = _XF16_CACHE.get(key)
    if hit is not None:
        _XF16_CACHE.move_to_end(key)
        return hit[1]
    copy = xf_bf16(base, precision)
    if not torch.cuda.is_current_stream_capturing():      # (a copy made inside a capture lives in that graph's memory pool)
        _XF16_CACHE[key] = (base, copy)
        while len(_XF16_CACHE) > 16:
            _XF16_CACHE.popitem(last=False)
    return copy


class PeLayout:
    """pe / dpe given as the upsampling net's own channel-last output on the STITCHED grids of the patched presets
    (rcb_siren_desc.pe_grid_dims): [S * n_datapoints, *(patch_nums[i] * patch_size[i]), E] instead of [G, P, E]."""

    def __init__(self, patch_nums, patch_size):
        self.patch_nums, self.patch_size = [int(v) for v in patch_nums], [int(v) for v in patch_size]
        assert 1 <= len(self.patch_nums) == len(self.patch_size) <= 3

    def shape(self, G, meta):
        n_inr = G // meta.samples
        per = int(np.prod(self.patch_nums))
        return (meta.samples * (n_inr // per),) + tuple(a * b for a, b in zip(self.patch_nums, self.patch_size)) + (meta.pe_dim,)


def _siren_desc(meta: SirenMeta, wvec, xf, pe=None, dw_bf16=None, chunks=1, pe_layout=None, xf16=None, dw_planes=None):
    if wvec.dim() != 2 or wvec.stride(1) != 1:
        raise RcbError("wvec must be 2-D with unit column stride")
    G = wvec.shape[0]
    if wvec.shape[1] != meta.d_net:
        raise RcbError(f"wvec has {wvec.shape[1]} columns, geometry needs {meta.d_net}")
    if G % meta.samples:
        raise RcbError("rows of wvec must be a multiple of samples")
    d = SirenDesc(G, meta.samples, meta.n_pix, meta.fourier_dim, meta.pe_dim, meta.n_hidden, meta.hidden,
                  meta.out_dim, _xf_stride(xf, meta, G // meta.samples), int(wvec.stride(0)), meta.w0,
                  meta.precision, int(pe is not None and pe.dtype == bf16), None if dw_bf16 is None else dw_bf16.data_ptr(), int(chunks), None)
    d.dw_bf16_stride = 0 if dw_bf16 is None else int(dw_bf16.stride(0))
    if dw_planes is not None:              # the gradient as (hi, lo) planes (ops.Planes)
        if (dw_planes.rows, dw_planes.cols) != (G, meta.d_net):
            raise RcbError("siren: gradient planes do not match the rows of wvec")
        d.dw_bf16, d.dw_lo, d.dw_bf16_stride = dw_planes.buf[0].data_ptr(), dw_planes.buf[1].data_ptr(), int(dw_planes.ld)
    if meta.hidden_dims and len(set(meta.hidden_dims)) > 1:
        if meta.precision != 0 or len(meta.hidden_dims) != meta.n_hidden:
            raise RcbError("per-layer hidden widths: fp32 mode only, one width per hidden layer")
        for i, w_ in enumerate(meta.hidden_dims):
            d.hidden_dims[i] = int(w_)
    if pe_layout is not None:
        nd = len(pe_layout.patch_nums)
        d.pe_grid_dims = nd
        for i in range(nd):
            d.pe_patch_nums[i] = pe_layout.patch_nums[i]
            d.pe_patch_size[i] = pe_layout.patch_size[i]
    if meta.precision in (1, 2) and pe is not None and pe.dtype == bf16 and not _os.environ.get("RCB_SIREN_NO_XF16"):   # (A/B switch)
        x16 = xf16 if xf16 is not None else _xf_bf16_cached(xf, meta.precision)
        if x16 is not None:
            want = (xf.shape[-2], (meta.fourier_dim + 7) // 8 * 8)
            if tuple(x16.shape) != want or x16.dtype != (bf16 if meta.precision == 1 else torch.float16) or not x16.is_contiguous():
                raise RcbError(f"xf16: expected a contiguous {want} copy of xf in the operand format (ops.xf_bf16(xf, precision)), "
                               f"got {tuple(x16.shape)} {x16.dtype}")
            d.xf_bf16 = x16.data_ptr()
            d._keep = x16
    return d, G


def _dev_ptr_strided(t):
    if not t.is_cuda or t.dtype != f32:
        raise RcbError("expected an fp32 GPU tensor")
    return C.c_void_p(t.data_ptr())


def _check_pe(pe, G, meta, pe_layout=None):
    if meta.pe_dim == 0:
        return
    want = (G, meta.n_pix, meta.pe_dim) if pe_layout is None else pe_layout.shape(G, meta)
    if pe is None or tuple(pe.shape) != tuple(want) or not pe.is_contiguous():
        raise RcbError(f"pe must be contiguous {list(want)}")
    if pe.dtype not in (f32, bf16) or (pe.dtype == bf16 and meta.precision == 0):
        raise RcbError("pe must be fp32 (any precision mode) or bf16 (16-bit modes only)")


def siren_fwd(xf, pe, wvec, meta: SirenMeta, pixel_chunks=None, pe_layout=None):
    lib = _lib.load()
    d, G = _siren_desc(meta, wvec, xf, pe, None, pixel_chunks or siren_pixel_chunks(wvec.shape[0], meta), pe_layout)
    _check_pe(pe, G, meta, pe_layout)
    y = torch.empty(G, meta.n_pix, meta.out_dim, device=wvec.device, dtype=f32)
    check(lib.rcb_siren_fwd(C.byref(d), _dev_ptr_strided(xf), ptr(pe, None, True), _dev_ptr_strided(wvec),
                            ptr(y), stream_ptr()), "rcb_siren_fwd")
    return y


def siren_bwd(xf, pe, wvec, dy, meta: SirenMeta, want_dpe=True, pixel_chunks=None):
    lib = _lib.load()
    chunks = pixel_chunks or siren_pixel_chunks(wvec.shape[0], meta)
    d, G = _siren_desc(meta, wvec, xf, pe, None, chunks)
    _check_pe(pe, G, meta)
    if tuple(dy.shape) != (G, meta.n_pix, meta.out_dim):
        raise RcbError("dy shape mismatch")
    dw = torch.empty(G, wvec.stride(0), device=wvec.device, dtype=f32)[:, :meta.d_net]
    part = torch.empty(chunks, G, wvec.stride(0), device=wvec.device, dtype=f32) if chunks > 1 else None
    dpe = torch.empty_like(pe) if (want_dpe and meta.pe_dim) else None
    check(lib.rcb_siren_bwd(C.byref(d), _dev_ptr_strided(xf), ptr(pe, None, True), _dev_ptr_strided(wvec),
                            ptr(dy.contiguous(), f32), _dev_ptr_strided(dw if part is None else part), ptr(dpe, None, True),
                            stream_ptr()), "rcb_siren_bwd")
    if part is not None:
        check(lib.rcb_siren_reduce_chunks(C.byref(d), ptr(part), C.c_void_p(0), _dev_ptr_strided(dw), C.c_void_p(0),
                                          stream_ptr()), "rcb_siren_reduce_chunks")
    return dw, dpe


def siren_wide_layers(meta: SirenMeta):
    """(number of layer vectors of maximal length, that length)"""
    dims = meta.dims
    sizes = [dims[i + 1] * (dims[i] + 1) for i in range(len(dims) - 1)]
    return sizes.count(max(sizes)), max(sizes)


def siren_loss_bwd(xf, pe, wvec, target, dy_scale: float, meta: SirenMeta, want_dpe=True, want_bf16=False,
                   pixel_chunks=None, pe_layout=None, xf16=None, want_planes=False, clock_probe=None):
    """-> (sse [G], dwvec [G, d_net] (row stride = wvec's), dpe [G,P,E] or None); with want_bf16 (16-bit modes) also a
    bf16 copy of dwvec, [G, d_net] with a row stride that is a multiple of 8 (rcb_siren_desc.dw_bf16: the operand of the
    A transform's batched weight-gradient GEMM, written by the kernel's epilogue).
    want_planes (16-bit modes): -> (sse, None, dpe, Planes): the gradient as its (hi, lo) bf16 planes INSTEAD of fp32
    (rcb_siren_desc.dw_bf16 + dw_lo): the operand form of ATransform.dgrad / .wgrad."""
    lib = _lib.load()
    G = wvec.shape[0]
    if want_planes:
        if meta.precision == 0:
            raise RcbError("gradient planes need a 16-bit precision mode")
        planes = Planes(G, meta.d_net, wvec.device)
        chunks = pixel_chunks or siren_pixel_chunks(G, meta)
        d, G = _siren_desc(meta, wvec, xf, pe, None, chunks, pe_layout, xf16, dw_planes=planes if chunks == 1 else None)
        _check_pe(pe, G, meta, pe_layout)
        N = G // meta.samples
        if tuple(target.shape) != (N, meta.n_pix, meta.out_dim):
            raise RcbError(f"target must be [{N},{meta.n_pix},{meta.out_dim}], got {tuple(target.shape)}")
        sse = torch.empty(G, device=wvec.device, dtype=f32)
        dpe = torch.empty_like(pe) if (want_dpe and meta.pe_dim) else None
        part = sse_part = None
        if chunks > 1:
            part = torch.empty(chunks, G, wvec.stride(0), device=wvec.device, dtype=f32)
            sse_part = torch.empty(chunks, G, device=wvec.device, dtype=f32)
        check(lib.rcb_siren_loss_bwd(C.byref(d), _dev_ptr_strided(xf), ptr(pe, None, True), _dev_ptr_strided(wvec),
                                     ptr(target, f32), C.c_float(dy_scale), ptr(sse if part is None else sse_part),
                                     ptr(part, f32, True), ptr(dpe, None, True), stream_ptr()), "rcb_siren_loss_bwd")
        if part is not None:
            d2, _ = _siren_desc(meta, wvec, xf, pe, None, chunks, None, xf16, dw_planes=planes)
            check(lib.rcb_siren_reduce_chunks(C.byref(d2), ptr(part), ptr(sse_part), C.c_void_p(0), ptr(sse), stream_ptr()),
                  "rcb_siren_reduce_chunks")
        return sse, None, dpe, planes
    dw16 = None
    if want_bf16:
        if meta.precision == 0:
            raise RcbError("the bf16 copy of the gradient needs a 16-bit precision mode")
        dw16 = torch.empty(G, (meta.d_net + 7) // 8 * 8, device=wvec.device, dtype=bf16)[:, :meta.d_net]
    chunks = pixel_chunks or siren_pixel_chunks(G, meta)
    d, G = _siren_desc(meta, wvec, xf, pe, None if chunks > 1 else dw16, chunks, pe_layout, xf16)
    if clock_probe is not None:          # int64 [256, 4]: per-workgroup clock stamps (rcb_siren_desc.clock_probe; shader_clock_ghz)
        if clock_probe.dtype != torch.int64 or clock_probe.numel() < 1024 or not clock_probe.is_contiguous():
            raise RcbError("clock_probe: contiguous int64 [256, 4] expected")
        d.clock_probe = clock_probe.data_ptr()
    _check_pe(pe, G, meta, pe_layout)
    N = G // meta.samples
    if tuple(target.shape) != (N, meta.n_pix, meta.out_dim):
        raise RcbError(f"target must be [{N},{meta.n_pix},{meta.out_dim}], got {tuple(target.shape)}")
    sse = torch.empty(G, device=wvec.device, dtype=f32)
    dw = torch.empty(G, wvec.stride(0), device=wvec.device, dtype=f32)[:, :meta.d_net]
    dpe = torch.empty_like(pe) if (want_dpe and meta.pe_dim) else None
    part = sse_part = None
    if chunks > 1:          # few rows: pixel tiles split over several workgroups per row, partials summed in chunk order
        part = torch.empty(chunks, G, wvec.stride(0), device=wvec.device, dtype=f32)
        sse_part = torch.empty(chunks, G, device=wvec.device, dtype=f32)
    check(lib.rcb_siren_loss_bwd(C.byref(d), _dev_ptr_strided(xf), ptr(pe, None, True), _dev_ptr_strided(wvec),
                                 ptr(target, f32), C.c_float(dy_scale), ptr(sse if part is None else sse_part),
                                 _dev_ptr_strided(dw if part is None else part), ptr(dpe, None, True), stream_ptr()),
          "rcb_siren_loss_bwd")
    if part is not None:
        d2, _ = _siren_desc(meta, wvec, xf, pe, dw16, chunks, None, xf16)
        check(lib.rcb_siren_reduce_chunks(C.byref(d2), ptr(part), ptr(sse_part), _dev_ptr_strided(dw), ptr(sse), stream_ptr()),
              "rcb_siren_reduce_chunks")
    return (sse, dw, dpe, dw16) if want_bf16 else (sse, dw, dpe)


class SirenFn(torch.autograd.Function):
    """y[G,P,C] = SIREN(xf | pe ; wvec) with gradients for pe and wvec."""

    @staticmethod
    def forward(ctx, xf, pe, wvec, meta):
        ctx.meta = meta
        ctx.save_for_backward(xf, pe, wvec)
        return siren_fwd(xf, pe.contiguous() if pe is not None else None, wvec, meta)

    @staticmethod
    def backward(ctx, dy):
        xf, pe, wvec = ctx.saved_tensors
        dw, dpe = siren_bwd(xf, pe.contiguous() if pe is not None else None, wvec, dy, ctx.meta,
                            want_dpe=ctx.needs_input_grad[1])
        return None, dpe, dw, None


# ----------------------------------------------------------------------------------------------
# posterior levels (K1 / K10 and their backward)
# ----------------------------------------------------------------------------------------------
class LevelSpec:
    """Device-side description of one level of the factorised Gaussian posterior together with the
    index maps that scatter it onto INRs (hierarchy) and un-permute it (test-time grouping)."""

    def __init__(self, loc, log_scale, cols_out, n_inr, row_map=None, row_perm=None, col_map=None,
                 enc_sample=None, enc_mask=None, scale_is_sigma=False):
        self.loc, self.log_scale = loc, log_scale
        self.scale_is_sigma = bool(scale_is_sigma)     # `log_scale` holds sigma itself (rcb_level.scale_is_sigma)
        self.rows, self.cols = loc.shape[0], loc.shape[1]
        self.cols_out, self.n_inr = int(cols_out), int(n_inr)
        dev = loc.device
        self.enc_sample, self.enc_mask = enc_sample, enc_mask

        def dev_i32(a):
            if a is None:
                return None
            return torch.as_tensor(np.ascontiguousarray(a), dtype=i32).to(dev).contiguous()
        self.row_map = dev_i32(row_map)
        self.row_perm = dev_i32(row_perm)
        self.col_map = dev_i32(col_map)
        # inverses for the backward gather
        self.member_ptr = self.member_idx = self.row_perm_inv = self.col_inv = None
        if row_map is not None:
            rm = np.asarray(row_map)
            order = np.argsort(rm, kind="stable")
            counts = np.bincount(rm, minlength=self.rows)
            self.member_ptr = dev_i32(np.r_[0, np.cumsum(counts)])
            self.member_idx = dev_i32(order)
        if row_perm is not None:
            rp = np.asarray(row_perm)
            inv = np.empty_like(rp)
            cols = np.arange(rp.shape[1])[None, :].repeat(rp.shape[0], 0)
            inv[rp, cols] = np.arange(rp.shape[0])[:, None]
            self.row_perm_inv = dev_i32(inv)
        if col_map is not None:
            cm = np.asarray(col_map)
            inv = np.full(self.cols, 2 ** 30, dtype=np.int64)   # columns not produced: d >= cols_out
            inv[cm] = np.arange(cm.shape[0])
            self.col_inv = dev_i32(inv)

    def c_fwd(self, eps, ws=None):
        """ws: optional scratch [2 * rows * cols] for the packed (mu, sigma) records of a gathered level (rcb_level.mu_sigma_ws)"""
        if tuple(eps.shape[-1:]) != (self.cols_out,):
            raise RcbError("eps last dim must equal cols_out")
        return Level(addr(self.loc.detach(), f32), addr(self.log_scale.detach(), f32), addr(self.enc_sample, f32),
                     addr(self.enc_mask, f32), addr(self.row_map, i32), addr(self.row_perm, i32),
                     addr(self.col_map, i32), addr(eps, f32), self.rows, self.cols, self.cols_out, int(self.scale_is_sigma),
                     addr(ws, f32))


def reparam_fwd(levels: Sequence[LevelSpec], eps: Sequence[torch.Tensor], samples: int):
    """out[N, S, cols_out(level 0)] (rcb_reparam_fwd)."""
    lib = _lib.load()
    n = levels[0].n_inr
    cols = levels[0].cols_out
    # gathered levels (test-time layout) of several INRs per launch: scratch for their packed (mu, sigma) records (rcb.h)
    staged = (len(levels) == 1 and levels[0].col_map is not None and levels[0].row_map is None and levels[0].row_perm is None
              and levels[0].cols * 16 <= 150 * 1024)                 # (the one-level LDS-staged kernel needs none)
    scratch = [torch.empty(2 * lv.rows * lv.cols, device=lv.loc.device, dtype=f32)
               if ((lv.col_map is not None or lv.row_perm is not None) and n * samples >= 8 and not staged) else None for lv in levels]
    arr = (Level * len(levels))(*[lv.c_fwd(e, w_) for lv, e, w_ in zip(levels, eps, scratch)])
    for e, lv in zip(eps, levels):
        if tuple(e.shape) != (n, samples, lv.cols_out) or not e.is_contiguous():
            raise RcbError(f"eps must be contiguous [{n},{samples},{lv.cols_out}], got {tuple(e.shape)}")
    out = torch.empty(n, samples, cols, device=levels[0].loc.device, dtype=f32)
    check(lib.rcb_reparam_fwd(arr, len(levels), n, samples, cols, ptr(out), stream_ptr()), "rcb_reparam_fwd")
    return out


def hier_rng_eligible(levels: Sequence[LevelSpec]):
    """the levels of a patched preset in training: level 0 one plain row per INR, the coarser ones behind row maps only"""
    if len(levels) < 2 or not rng_eligible(levels[0]):
        return False
    return all(lv.row_perm is None and lv.col_map is None and lv.enc_mask is None and not lv.scale_is_sigma
               and lv.cols_out == lv.cols == levels[0].cols for lv in levels[1:])


def reparam_hier_rng(levels: Sequence[LevelSpec], eps_out: Sequence[torch.Tensor], seed: int, rng_streams: Sequence[int], step,
                     group_offset=0):
    """-> out [N, 1, cols]: the multi-level training sample with every level's noise drawn in the kernel
    (rcb_reparam_hier_rng_fwd); eps_out[l] ([N, 1, cols], fp32) receives level l's noise for the posterior update.  `step` is the
    device-resident int64 step counter; level l draws from Philox stream rng_streams[l] at element index n * cols + d."""
    lib = _lib.load()
    n, cols = levels[0].n_inr, levels[0].cols_out
    for e in eps_out:
        if tuple(e.shape) != (n, 1, cols) or e.dtype != f32 or not e.is_contiguous():
            raise RcbError(f"reparam_hier_rng: eps buffers must be contiguous fp32 [{n}, 1, {cols}]")
    arr = (Level * len(levels))(*[lv.c_fwd(e) for lv, e in zip(levels, eps_out)])
    outs = (C.c_void_p * len(levels))(*[e.data_ptr() for e in eps_out])
    streams = (C.c_uint32 * len(levels))(*[int(v) for v in rng_streams])
    out = torch.empty(n, 1, cols, device=levels[0].loc.device, dtype=f32)
    check(lib.rcb_reparam_hier_rng_fwd(arr, len(levels), n, cols, outs, ptr(out), C.c_uint64(seed & (2 ** 64 - 1)), streams,
                                       ptr(step, torch.int64), C.c_uint64(group_offset), stream_ptr()), "rcb_reparam_hier_rng_fwd")
    return out


class Planes:
    """A [rows, cols] fp32 matrix held as its two bf16 PLANES, x = hi + lo: hi = bf16(x), lo = bf16(x - hi) -- what the
    A-transform kernels split a row into, written by the PRODUCER of the row instead (rcb_reparam_rng_fwd, the fused next
    sample of rcb_posterior_bwd, the SIREN kernels' gradient epilogue): the same 4 bytes per element as fp32, no conversion
    work in the consumer, and `hi` alone is the weight-gradient GEMM's operand.  One buffer [2, rows, ld], ld a multiple of
    32 elements (64-byte rows: the LDS-DMA streams of rcb_atrans_apply)."""

    def __init__(self, rows, cols, device):
        self.rows, self.cols = int(rows), int(cols)
        self.ld = (self.cols + 31) // 32 * 32
        self.buf = torch.zeros(2, self.rows, self.ld, device=device, dtype=bf16)

    @property
    def hi(self):
        return self.buf[0, :, :self.cols]

    @property
    def lo(self):
        return self.buf[1, :, :self.cols]

    def float(self):
        """the fp32 values the planes stand for (tests, fallbacks)"""
        return self.hi.float() + self.lo.float()

    @classmethod
    def from_float(cls, x):
        p = cls(x.shape[0], x.shape[1], x.device)
        h = x.to(bf16)
        p.hi.copy_(h)
        p.lo.copy_((x - h.float()).to(bf16))
        return p


def rng_eligible(lv: LevelSpec):
    """plain level (no maps, no masks, every column produced): the in-kernel noise path applies"""
    return (lv.row_map is None and lv.row_perm is None and lv.col_map is None and lv.enc_mask is None
            and not lv.scale_is_sigma and lv.cols_out == lv.cols and lv.n_inr == lv.rows and lv.loc.is_contiguous() and lv.log_scale.is_contiguous())


def sample_buffers(lv: LevelSpec, want_bf16=False, planes=False, want_eps=True):
    """(out [n, 1, cols], eps [n, 1, cols], bf16 copy [n, ld16] or None): the buffers of one level's sample, for callers
    that keep them across steps (PriorBNNmodel.train: the posterior update of step t writes the sample of step t + 1).
    planes: the sample as a Planes pair INSTEAD of fp32 (out = None, third entry = the Planes).  want_eps=False: no copy of
    the noise (the posterior update re-draws it, rcb_level_bwd.eps_from_rng)."""
    n, cols = lv.rows, lv.cols
    eps = torch.empty(n, 1, cols, device=lv.loc.device, dtype=f32) if want_eps else None
    if planes:
        return None, eps, Planes(n, cols, lv.loc.device)
    out = torch.empty(n, 1, cols, device=lv.loc.device, dtype=f32)
    o16 = torch.empty(n, (cols + 7) // 8 * 8, device=lv.loc.device, dtype=bf16) if want_bf16 else None
    return out, eps, o16


class NextSample:
    """argument of posterior_bwd(next_sample=): draw the next step's sample in the same pass (rcb_level_bwd.next_*) into
    the buffers of sample_buffers(); step = the device step counter, the sample is that of counter + step_add"""

    def __init__(self, buffers, seed: int, rng_stream: int, step, step_add=1, redraw_eps=False, group_offset=0):
        self.out, self.eps, self.o16 = buffers
        self.group_offset = int(group_offset)      # Philox group of element 0 (see reparam_rng)
        self.seed, self.rng_stream, self.step, self.step_add = seed, rng_stream, step, step_add
        # redraw_eps: THIS step's noise is re-drawn from the counter it came from (rcb_level_bwd.eps_from_rng) instead of
        # being read back from memory -- posterior_bwd is then called with eps = None
        self.redraw_eps = bool(redraw_eps)


def rng_group_offset(row0: int, cols: int):
    """Philox group offset with which a launch over rows [row0, ...) of a [rows, cols] level draws the noise those rows have
    in a launch over all rows (rcb_reparam_rng_fwd group_offset); row0 * cols must be a multiple of 4"""
    if (int(row0) * int(cols)) % 4:
        raise RcbError(f"rng_group_offset: row offset {row0} x {cols} columns is not a multiple of 4 elements")
    return int(row0) * int(cols) // 4


def reparam_rng(lv: LevelSpec, seed: int, rng_stream: int, step, want_bf16=False, buffers=None, group_offset=0):
    """-> (out [n, 1, cols], eps [n, 1, cols]): reparameterised sample with the noise drawn inside the kernel
    (rcb_reparam_rng_fwd).  `step` is the device-resident int64 step counter.  want_bf16: third result, a bf16 copy of
    out as [n, cols] with a row stride that is a multiple of 8 (operand of the A transform's weight-gradient GEMM).
    buffers: write into the tensors of sample_buffers() instead of new ones."""
    lib = _lib.load()
    if not rng_eligible(lv):
        raise RcbError("reparam_rng: plain levels only")
    n, cols = lv.rows, lv.cols
    out, eps, o16 = buffers if buffers is not None else sample_buffers(lv, want_bf16)
    if isinstance(o16, Planes):            # the sample as (hi, lo) planes (sample_buffers(planes=True)): -> (None, eps, Planes)
        check(lib.rcb_reparam_rng_fwd(ptr(lv.loc.detach(), f32), ptr(lv.log_scale.detach(), f32), C.c_int64(n * cols),
                                      C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint32(rng_stream), ptr(step, torch.int64),
                                      ptr(eps, f32, True), ptr(out, f32, True), C.c_void_p(o16.buf[0].data_ptr()),
                                      C.c_void_p(o16.buf[1].data_ptr()), int(cols), C.c_int64(o16.ld), C.c_uint64(group_offset),
                                      stream_ptr()), "rcb_reparam_rng_fwd")
        return out, eps, o16
    if want_bf16 and o16 is None:
        raise RcbError("reparam_rng: the buffers have no bf16 copy")
    if not want_bf16:
        o16 = None
    check(lib.rcb_reparam_rng_fwd(ptr(lv.loc.detach(), f32), ptr(lv.log_scale.detach(), f32), C.c_int64(n * cols),
                                  C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint32(rng_stream), ptr(step, torch.int64),
                                  ptr(eps, f32, True), ptr(out), ptr(o16, bf16, True), C.c_void_p(0), int(cols),
                                  C.c_int64(0 if o16 is None else o16.stride(0)), C.c_uint64(group_offset), stream_ptr()),
          "rcb_reparam_rng_fwd")
    return (out, eps, o16[:, :cols]) if want_bf16 else (out, eps)


def philox_normal(n, seed: int, rng_stream: int, step, device="cuda", group_offset=0):
    """the noise stream of reparam_rng as a tensor (step: python int or device int64 tensor)"""
    lib = _lib.load()
    out = torch.empty(n, device=device, dtype=f32)
    dev_step = step if torch.is_tensor(step) else None
    check(lib.rcb_philox_normal(ptr(out), C.c_int64(n), C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint32(rng_stream),
                                ptr(dev_step, torch.int64, True), C.c_int64(0 if dev_step is not None else int(step)),
                                C.c_uint64(group_offset), stream_ptr()), "rcb_philox_normal")
    return out


def posterior_bwd(lv: LevelSpec, p_loc, p_scale, p_is_log: bool, kl_scalar: float, d_out, eps, samples: int,
                  beta=None, group_idx=None, n_groups=0, adam: Optional[AdamCfg] = None, state=None,
                  want_grads=False, kl_accum=None, kl_scalar_dev=None, next_sample: Optional["NextSample"] = None):
    """Fused gradient gather + KL gradient (+ Adam).  Returns (g_loc, g_log_scale) when requested.
    next_sample (plain levels with Adam): also draw the next step's sample from the updated parameters."""
    lib = _lib.load()
    g_loc = g_ls = None
    if want_grads:
        g_loc = torch.empty_like(lv.loc)
        g_ls = torch.empty_like(lv.log_scale)
    if adam is not None and state is None:
        raise RcbError("adam step needs state")
    s = state or {}
    b = LevelBwd(addr(lv.loc.detach(), f32), addr(lv.log_scale.detach(), f32), addr(lv.enc_mask, f32),
                 addr(p_loc, f32), addr(p_scale, f32), int(bool(p_is_log)), addr(beta, f32), addr(group_idx, i32),
                 int(n_groups), float(kl_scalar), addr(d_out, f32), addr(eps, f32), addr(lv.member_ptr, i32),
                 addr(lv.member_idx, i32), addr(lv.row_perm_inv, i32), addr(lv.col_inv, i32), lv.rows, lv.cols,
                 lv.cols_out, int(samples), addr(g_loc, f32), addr(g_ls, f32), addr(s.get("m_loc"), f32),
                 addr(s.get("v_loc"), f32), addr(s.get("m_ls"), f32), addr(s.get("v_ls"), f32), addr(kl_accum, torch.int64),
                 addr(kl_scalar_dev, f32))
    b.col_map = addr(lv.col_map, i32)           # (a hint: threads indexed by the produced column where that pays, rcb.h)
    sums_ws = None
    if (d_out is not None and eps is not None and samples > 1 and lv.col_inv is not None and lv.member_ptr is None
            and lv.rows == lv.n_inr):
        # test-time level 1 behind the per-column row permutation: sums over the samples in a contiguous pass first (rcb.h)
        sums_ws = torch.empty(2 * lv.rows * lv.cols_out, device=lv.loc.device, dtype=f32)
        b.sample_sum_ws = sums_ws.data_ptr()
    if next_sample is not None:
        ns = next_sample
        for t_ in (ns.out, ns.eps):
            if t_ is not None and tuple(t_.shape) != (lv.rows, 1, lv.cols):
                raise RcbError("posterior_bwd: next-sample buffers do not match the level")
        b.next_out, b.next_eps = addr(ns.out, f32), addr(ns.eps, f32)
        if isinstance(ns.o16, Planes):
            if (ns.o16.rows, ns.o16.cols) != (lv.rows, lv.cols):
                raise RcbError("posterior_bwd: next-sample planes do not match the level")
            b.next_out_bf16, b.next_out_lo = ns.o16.buf[0].data_ptr(), ns.o16.buf[1].data_ptr()
            b.next_ld_bf16 = int(ns.o16.ld)
        else:
            if ns.out is None:
                raise RcbError("posterior_bwd: the next sample needs an fp32 buffer or planes")
            b.next_out_bf16 = addr(ns.o16, bf16)
            b.next_ld_bf16 = 0 if ns.o16 is None else int(ns.o16.stride(0))
        b.eps_from_rng = int(ns.redraw_eps)
        b.rng_group_offset = ns.group_offset
        if ns.redraw_eps and eps is not None:
            raise RcbError("posterior_bwd: redraw_eps takes eps = None")
        b.rng_seed = ns.seed & (2 ** 64 - 1)
        b.rng_step_dev = addr(ns.step, torch.int64)
        b.rng_step_add = int(ns.step_add)
        b.rng_stream = int(ns.rng_stream)
    if d_out is not None and tuple(d_out.shape) != (lv.n_inr, samples, lv.cols_out):
        raise RcbError(f"d_out must be [{lv.n_inr},{samples},{lv.cols_out}], got {tuple(d_out.shape)}")
    check(lib.rcb_posterior_bwd(C.byref(b), C.byref(adam) if adam is not None else None, stream_ptr()),
          "rcb_posterior_bwd")
    return g_loc, g_ls


class ReparamFn(torch.autograd.Function):
    """Differentiable sampling: inputs (loc_0, log_scale_0, loc_1, ...) -> out[N,S,D]."""

    @staticmethod
    def forward(ctx, levels, eps, samples, *params):
        ctx.levels, ctx.eps, ctx.samples = levels, eps, samples
        return reparam_fwd(levels, eps, samples)

    @staticmethod
    def backward(ctx, d_out):
        d_out = d_out.contiguous()
        grads = []
        for lv, e in zip(ctx.levels, ctx.eps):
            d = d_out if lv.cols_out == d_out.shape[-1] else d_out[..., :lv.cols_out].contiguous()
            zeros = torch.zeros(lv.cols, device=d.device, dtype=f32)
            ones = torch.ones(lv.cols, device=d.device, dtype=f32)
            g = posterior_bwd(lv, zeros, ones, False, 0.0, d, e, ctx.samples, want_grads=True)
            grads += list(g)
        return (None, None, None, *grads)


def sample_levels(levels: Sequence[LevelSpec], eps, samples):
    params = []
    for lv in levels:
        params += [lv.loc, lv.log_scale]
    return ReparamFn.apply(tuple(levels), tuple(eps), samples, *params)


# ----------------------------------------------------------------------------------------------
# KL (K5 - K8)
# ----------------------------------------------------------------------------------------------
def gauss_kl(loc, log_scale, p_loc, p_scale, p_is_log=False, beta=None, group_idx=None, seg_start=None,
             seg_end=None, want_rows=True, want_groups=False):
    """-> (kl_row fp64 [rows] or None, kl_group fp64 [rows, G] or None)."""
    lib = _lib.load()
    loc2 = loc.detach().reshape(loc.shape[0], -1)
    ls2 = log_scale.detach().reshape(loc.shape[0], -1)
    rows, cols = loc2.shape
    n_groups = int(seg_start.shape[0]) if seg_start is not None else (int(beta.shape[1]) if beta is not None else 0)
    kl_row = torch.empty(rows, device=loc.device, dtype=f64) if want_rows else None
    kl_group = torch.empty(rows, n_groups, device=loc.device, dtype=f64) if want_groups else None
    pl, ps = p_loc.reshape(-1), p_scale.reshape(-1)
    if pl.numel() != cols or ps.numel() != cols:
        raise RcbError("prior shape mismatch")
    check(lib.rcb_gauss_kl(ptr(loc2, f32), ptr(ls2, f32), ptr(pl, f32), ptr(ps, f32), int(bool(p_is_log)), rows,
                           cols, ptr(beta, f32, True), ptr(group_idx, i32, True), n_groups,
                           ptr(seg_start, i32, True), ptr(seg_end, i32, True), ptr(kl_row, f64, True),
                           ptr(kl_group, f64, True), stream_ptr()), "rcb_gauss_kl")
    return kl_row, kl_group


class GaussKLFn(torch.autograd.Function):
    """sum_{r,j} w(r,j) KL(N(loc, st(log_scale)) || N(p_loc, p_scale)) as a 0-d fp32 tensor."""

    @staticmethod
    def forward(ctx, loc, log_scale, p_loc, p_scale, p_is_log, beta, group_idx, seg_start, seg_end):
        ctx.save_for_backward(loc, log_scale, p_loc, p_scale)
        # snapshot beta: the annealing step may overwrite it in place between forward and backward,
        # and the gradient must use the weights the loss was formed with (test_model.py:629-635)
        ctx.aux = (p_is_log, None if beta is None else beta.clone(), group_idx, seg_start, seg_end)
        rows, _ = gauss_kl(loc, log_scale, p_loc, p_scale, p_is_log, beta, group_idx, seg_start, seg_end)
        return rows.sum().to(f32)

    @staticmethod
    def backward(ctx, g):
        loc, log_scale, p_loc, p_scale = ctx.saved_tensors
        p_is_log, beta, group_idx, seg_start, seg_end = ctx.aux
        shape = loc.shape
        l2 = loc.detach().reshape(shape[0], -1)
        s2 = log_scale.detach().reshape(shape[0], -1)
        lv = LevelSpec(l2, s2, l2.shape[1], l2.shape[0])
        n_groups = int(beta.shape[1]) if beta is not None else 0
        gl, gs = posterior_bwd(lv, p_loc.reshape(-1).contiguous(), p_scale.reshape(-1).contiguous(), p_is_log, 1.0,
                               None, None, 1, beta=beta, group_idx=group_idx, n_groups=n_groups, want_grads=True)
        return gl.reshape(shape) * g, gs.reshape(shape) * g, None, None, None, None, None, None, None


COLSUM_FX = 2.0 ** 30          # RCB_COLSUM_FX_SCALE
KL_FX = 2.0 ** 24              # RCB_KL_FX_SCALE: the KL slots of rcb_posterior_bwd / rcb_step_* count 2^-24 nats
MOM_FX, MOM_FX_LO = 2.0 ** 30, 2.0 ** 32      # RCB_MOM_FX_SCALE, RCB_MOM_FX_LO_SCALE


def gauss_kl_colsum_fx(loc, q_scale, p_loc, p_scale, q_is_log=False):
    """-> int64 [cols + 1]: sum over rows of the elementwise KL in units of 1 / COLSUM_FX nats (every element rounded to the
    integer grid on its own, exact integer accumulation: bitwise independent of the order of the workgroups and of how the
    rows are sharded; summed over ranks with an integer all-reduce) + the count of not-representable elements."""
    lib = _lib.load()
    l2 = loc.detach().reshape(loc.shape[0], -1).contiguous()
    s2 = q_scale.detach().reshape(loc.shape[0], -1).contiguous()
    rows, cols = l2.shape
    out = torch.empty(cols + 1, device=loc.device, dtype=torch.int64)     # [cols] sums + the not-representable counter
    check(lib.rcb_gauss_kl_colsum(ptr(l2, f32), ptr(s2, f32), int(bool(q_is_log)), ptr(p_loc.reshape(-1).contiguous(), f32),
                                  ptr(p_scale.reshape(-1).contiguous(), f32), rows, cols, ptr(out), stream_ptr()),
          "rcb_gauss_kl_colsum")
    return out


def gauss_kl_colsum(loc, q_scale, p_loc, p_scale, q_is_log=False):
    """-> fp64 [cols]: sum over rows of the elementwise KL (NaN everywhere if any element's KL was not finite)."""
    return colsum_from_fx(gauss_kl_colsum_fx(loc, q_scale, p_loc, p_scale, q_is_log))


def colsum_from_fx(fx):
    """fp64 column sums from the [cols + 1] fixed-point result (last element: count of non-finite / out-of-range elements,
    which the reference would have propagated as NaN / Inf)"""
    v = fx[:-1].to(f64) / COLSUM_FX
    return torch.where(fx[-1] != 0, torch.full_like(v, float("nan")), v)


def beta_update(kl_group, beta, done_u8, bits=16.0, upper=0.0, lower=0.4, step=0.05):
    lib = _lib.load()
    rows, G = beta.shape
    check(lib.rcb_beta_update(ptr(kl_group, f64), ptr(beta, f32), ptr(done_u8, torch.uint8, True), rows, G,
                              C.c_double(bits), C.c_double(upper), C.c_double(lower), C.c_double(step),
                              stream_ptr()), "rcb_beta_update")


# ----------------------------------------------------------------------------------------------
# Adam, moments, REC, softplus
# ----------------------------------------------------------------------------------------------
def adam_cfg(lr, step, beta1=0.9, beta2=0.999, eps=1e-8, dyn=None):
    """dyn: optional fp32 GPU tensor [2] = {lr/(1-beta1^t), sqrt(1-beta2^t)} read by the kernels at run time."""
    return AdamCfg(float(lr), float(beta1), float(beta2), float(eps), int(step), addr(dyn, f32))


def adam_table(lr, n_steps, beta1=0.9, beta2=0.999):
    """[n_steps, 2] host table of the per-step scalars (step t = row t-1), computed in fp64 like torch."""
    t = np.arange(1, n_steps + 1, dtype=np.float64)
    return torch.from_numpy(np.stack([lr / (1.0 - beta1 ** t), np.sqrt(1.0 - beta2 ** t)], 1).astype(np.float32))


def adam_flat(p, g, m, v, cfg: AdamCfg):
    lib = _lib.load()
    if not (p.is_contiguous() and g.is_contiguous()):
        raise RcbError("adam_flat needs contiguous tensors")
    check(lib.rcb_adam_flat(ptr(p.detach(), f32), ptr(g, f32), ptr(m, f32), ptr(v, f32), C.c_int64(p.numel()),
                            C.byref(cfg), stream_ptr()), "rcb_adam_flat")


def step_begin(table, step, dyn, kl_slots=None):
    """dyn <- table[step] (per-step Adam scalars), kl_slots <- 0 (rcb_step_begin); everything stays on the device."""
    lib = _lib.load()
    if table.dim() != 2 or table.shape[1] != 2 or (kl_slots is not None and kl_slots.numel() != 1024):
        raise RcbError("step_begin: table must be [n_steps, 2], kl_slots [1024]")
    check(lib.rcb_step_begin(ptr(table, f32), C.c_int64(table.shape[0]), ptr(step, torch.int64), ptr(dyn, f32),
                             ptr(kl_slots, torch.int64, True), stream_ptr()), "rcb_step_begin")


def step_end(step, sse=None, mse_scale=1.0, kl_slots=None, mse_log=None, kl_log=None, aux_counter=None):
    """mse_log[step] = mse_scale * sum(sse), kl_log[step] = sum(kl_slots), step += 1, aux_counter += 1 (rcb_step_end)."""
    lib = _lib.load()
    n_log = min(t.numel() for t in (mse_log, kl_log) if t is not None) if (mse_log is not None or kl_log is not None) else 0
    check(lib.rcb_step_end(ptr(sse, f32, True), 0 if sse is None else sse.numel(), C.c_double(mse_scale),
                           ptr(kl_slots, torch.int64, True), ptr(mse_log, f64, True), ptr(kl_log, f64, True), C.c_int64(n_log),
                           ptr(step, torch.int64), ptr(aux_counter, torch.int64, True), stream_ptr()), "rcb_step_end")


ADAM_MAX_TENSORS = 16


def adam_multi(params, grads, ms, vs, cfg: AdamCfg):
    """Adam step over a list of fp32 tensors in one launch (rcb_adam_multi); lists longer than 16 are chunked."""
    lib = _lib.load()
    items = list(zip(params, grads, ms, vs))
    for k in range(0, len(items), ADAM_MAX_TENSORS):
        chunk = items[k:k + ADAM_MAX_TENSORS]
        arr = (AdamTensor * len(chunk))()
        for i, (p, g, m, v) in enumerate(chunk):
            if not (p.is_contiguous() and g.is_contiguous()) or g.numel() != p.numel():
                raise RcbError("adam_multi needs contiguous tensors of matching size")
            arr[i] = AdamTensor(ptr(p.detach(), f32).value, ptr(g, f32).value, ptr(m, f32).value, ptr(v, f32).value,
                                p.numel())
        check(lib.rcb_adam_multi(arr, len(chunk), C.byref(cfg), stream_ptr()), "rcb_adam_multi")


def col_moments_fx(loc, log_scale):
    """-> int64 [6 * cols + 1]: exact fixed-point sums over the rows of loc (rcb_col_moments): as [3, 2, cols] the (hi, lo)
    parts of sum x, sum x^2, sum sigma^2, then the count of terms that were NaN / Inf / out of range.  Integer sums: add them
    over ranks with an integer all-reduce, then moments_from_fx."""
    lib = _lib.load()
    l2 = loc.detach().reshape(loc.shape[0], -1)
    s2 = log_scale.detach().reshape(loc.shape[0], -1)
    rows, cols = l2.shape
    out = torch.empty(6 * cols + 1, device=loc.device, dtype=torch.int64)
    check(lib.rcb_col_moments(ptr(l2, f32), ptr(s2, f32), rows, cols, ptr(out), stream_ptr()), "rcb_col_moments")
    return out


def moments_from_fx(fx, n_rows):
    """(sum, M2 = sum (x - mean)^2, sum sigma^2) in fp64 from the fixed-point sums of n_rows rows (col_moments_fx layout);
    all NaN when the kernel counted a term it could not represent (a diverged posterior must not refit a finite prior)"""
    bad, fx = fx[-1], fx[:-1].view(3, 2, -1)
    v = (fx[:, 0].to(f64) + fx[:, 1].to(f64) / MOM_FX_LO) / MOM_FX
    v = torch.where(bad != 0, torch.full_like(v, float("nan")), v)
    return v[0], v[1] - v[0] * v[0] / float(n_rows), v[2]


def col_moments(loc, log_scale):
    """-> (sum, m2, sum sigma^2) fp64 [cols] over the rows of loc."""
    return moments_from_fx(col_moments_fx(loc, log_scale), loc.shape[0])


def shader_clock_ghz(probe):
    """mean shader clock of the workgroups that filled a rcb_siren_desc.clock_probe buffer (int64 [256, 4]; call after a
    synchronisation): (memtime end - start) / ((memrealtime end - start) / 100 MHz) per workgroup, averaged"""
    p = probe.cpu().double()
    p = p[(p[:, 3] > p[:, 1])]
    if p.shape[0] == 0:
        return float("nan")
    return float(((p[:, 2] - p[:, 0]) / ((p[:, 3] - p[:, 1]) / 1e8)).mean() / 1e9)


def softplus_scale(log_scale):
    lib = _lib.load()
    out = torch.empty_like(log_scale)
    check(lib.rcb_softplus_scale(ptr(log_scale.detach(), f32), ptr(out), C.c_int64(out.numel()), stream_ptr()),
          "rcb_softplus_scale")
    return out


class RecTables:
    """Candidate tables of the A* coder on the device, one per group length: the reference's fp64 [K, g] table
    (test_model.py:493-498) held as fp32 TRANSPOSED [g, K] -- its values are fp32-precision ndtri widened to fp64, so the
    narrowing is exact (checked on insertion) -- plus the device-side pointer / max |xi| arrays the kernels index by group
    length (include/rcb.h, rcb_rec_desc)."""

    def __init__(self, device, n_candidates):
        self.device, self.K = torch.device(device), int(n_candidates)
        self.t = {}            # g -> fp32 [g, K]
        self.absmax = {}
        self._dev = None       # (max_glen, pointer array int64 [max_glen + 1], absmax fp64 [max_glen + 1])

    def __contains__(self, g):
        return int(g) in self.t

    def add(self, g, table):
        """table: fp64 (or fp32) [K, g] tensor, any device."""
        g = int(g)
        if tuple(table.shape) != (self.K, g):
            raise RcbError(f"candidate table for group length {g} must be [{self.K}, {g}], got {tuple(table.shape)}")
        t32 = table.to(f32)
        if not torch.equal(t32.to(table.dtype), table):
            raise RcbError("candidate table is not exactly representable in fp32 (the reference's tables are fp32-precision "
                           "data, SURVEY A16): refusing to round it")
        self.t[g] = t32.t().contiguous().to(self.device)
        self.absmax[g] = float(table.abs().max())
        self._dev = None

    @classmethod
    def from_dict(cls, tables: dict, device, n_candidates):
        rt = cls(device, n_candidates)
        for g, t in tables.items():
            rt.add(g, t)
        return rt

    def device_arrays(self):
        if self._dev is None:
            mg = max(self.t)
            ptrs = np.zeros(mg + 1, dtype=np.int64)
            amax = np.zeros(mg + 1, dtype=np.float64)
            for g, t in self.t.items():
                ptrs[g] = t.data_ptr()
                amax[g] = self.absmax[g]
            self._dev = (mg, torch.from_numpy(ptrs).to(self.device), torch.from_numpy(amax).to(self.device))
        return self._dev


class RecJobs:
    """(row, start, glen[, group]) arrays of a batch of encodes on the device (int32)."""

    def __init__(self, row, start, glen, group=None):
        self.row, self.start, self.glen, self.group = row, start, glen, group
        self.n = int(row.shape[0])

    @classmethod
    def from_host(cls, device, row, start, glen, group=None, rows=None, cols=None, sort=True):
        """host arrays, validated here (device-built job lists are validated inside the kernels); sorted by group
        length so that the fast scorer's eight-job workgroups share table loads.  Returns (jobs, order) with
        order[i] = position of sorted job i in the caller's arrays."""
        row, start, glen = (np.asarray(v, dtype=np.int64).reshape(-1) for v in (row, start, glen))
        if row.shape[0] == 0:
            raise RcbError("rec: no jobs")
        if glen.min() < 1:
            raise RcbError("rec: group length < 1")
        if rows is not None and ((row < 0).any() or (row >= rows).any() or (start < 0).any() or (start + glen > cols).any()):
            raise RcbError("job outside the parameter matrix")
        order = np.argsort(glen, kind="stable") if sort else np.arange(row.shape[0])

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a[order]).astype(np.int32)).to(device)
        grp = None if group is None else dev(np.asarray(group, dtype=np.int64).reshape(-1))
        return cls(dev(row), dev(start), dev(glen), grp), order


def _rec_desc(loc, scale, p_loc, p_scale, tables: RecTables, gumbel, gumbel_absmax, jobs: RecJobs):
    mg, ptrs, amax = tables.device_arrays()
    rows, cols = (loc.shape if loc is not None else (int(1 << 30), p_loc.shape[0]))
    d = _lib.RecDesc(addr(None if loc is None else loc.detach(), f32), addr(scale, f32), addr(p_loc, f32), addr(p_scale, f32),
                     int(rows), int(cols), addr(ptrs, torch.int64), addr(amax, f64), int(mg),
                     addr(gumbel, f64), float(gumbel_absmax), int(tables.K), addr(jobs.row, i32), addr(jobs.start, i32),
                     addr(jobs.glen, i32), jobs.n)
    return d, (ptrs, amax)


REC_EXACT, REC_FAST = 0, 1


def rec_score(loc, scale, p_loc, p_scale, tables: RecTables, gumbel, jobs: RecJobs, mode=REC_FAST, want_logw0=False,
              gumbel_absmax=None):
    """Batched A* scoring, everything on the device (rcb_rec_score_argmax).
    -> (idx int32 [B], best fp64 [B, 2], uncertified uint8 [B] (jobs the exact scorer had to decide), logw0 or None)."""
    lib = _lib.load()
    if loc.dim() != 2 or tuple(scale.shape) != tuple(loc.shape) or p_loc.numel() != loc.shape[1] or p_scale.numel() != loc.shape[1]:
        raise RcbError("rec_score: loc / scale [rows, cols] and priors [cols] expected")
    K = tables.K
    if gumbel.numel() != K:
        raise RcbError(f"rec_score: {gumbel.numel()} Gumbel values for {K} candidates")
    if gumbel_absmax is None:
        gumbel_absmax = float(gumbel.abs().max())
    d, keep = _rec_desc(loc, scale, p_loc, p_scale, tables, gumbel, gumbel_absmax, jobs)
    dev = loc.device
    B = jobs.n
    n_ws = int(lib.rcb_rec_workspace_bytes(B, d.max_glen, K))
    if n_ws < 0:
        raise RcbError("rcb_rec_workspace_bytes rejected the shape")
    ws = torch.empty(n_ws, device=dev, dtype=torch.uint8)
    idx = torch.empty(B, device=dev, dtype=i32)
    best = torch.empty(B, 2, device=dev, dtype=f64)
    unc = torch.empty(B, device=dev, dtype=torch.uint8)
    logw0 = torch.empty(K, device=dev, dtype=f64) if want_logw0 else None
    check(lib.rcb_rec_score_argmax(C.byref(d), int(mode), ptr(ws), C.c_int64(n_ws), ptr(idx), ptr(best), ptr(unc),
                                   ptr(logw0, f64, True), stream_ptr()), "rcb_rec_score_argmax")
    return idx, best, unc, logw0


def rec_commit(p_loc, p_scale, tables: RecTables, jobs: RecJobs, idx, n_groups=0, want_z=False, enc_sample=None,
               enc_mask=None, done=None, beta=None, idx_groupwise=None):
    """z = p_loc + p_scale * xi[idx] per job (fp64 mul, add) and the state update of compress_group (rcb_rec_commit).
    enc_sample / enc_mask: fp32 [rows, cols]; done uint8 / beta fp32 / idx_groupwise int32: [rows, n_groups]."""
    lib = _lib.load()
    rows = enc_sample.shape[0] if enc_sample is not None else (done.shape[0] if done is not None else 1 << 30)
    mg, ptrs, amax = tables.device_arrays()
    d = _lib.RecDesc(None, None, addr(p_loc, f32), addr(p_scale, f32), int(rows), int(p_loc.numel()), addr(ptrs, torch.int64),
                     addr(amax, f64), int(mg), None, 0.0, int(tables.K), addr(jobs.row, i32), addr(jobs.start, i32),
                     addr(jobs.glen, i32), jobs.n)
    z = torch.zeros(jobs.n, mg, device=p_loc.device, dtype=f64) if want_z else None
    for t_, shape in ((enc_sample, (rows, p_loc.numel())), (enc_mask, (rows, p_loc.numel())), (done, (rows, n_groups)),
                      (beta, (rows, n_groups)), (idx_groupwise, (rows, n_groups))):
        if t_ is not None and tuple(t_.shape) != tuple(shape):
            raise RcbError(f"rec_commit: state tensor of shape {tuple(t_.shape)}, expected {tuple(shape)}")
    check(lib.rcb_rec_commit(C.byref(d), ptr(idx, i32), ptr(jobs.group, i32, True), int(n_groups), ptr(z, f64, True),
                             ptr(enc_sample, f32, True), ptr(enc_mask, f32, True), ptr(done, torch.uint8, True),
                             ptr(beta, f32, True), ptr(idx_groupwise, i32, True), stream_ptr()), "rcb_rec_commit")
    return z


def rec_score_argmax(loc, scale, p_loc, p_scale, tables, gumbel, job_row, job_start, job_glen, want_logw0=False,
                     mode=REC_FAST):
    """Convenience form with host job arrays (validated here).  tables: RecTables or {group_len: fp64 [K, g] tensor}.
    -> (idx int32 [B] (GPU), z fp64 [B, max_g] (GPU), best fp64 [B,2] (GPU), logw0 or None), in the caller's job order."""
    dev = loc.device
    K = int(gumbel.shape[0])
    rows, cols = loc.shape
    if not isinstance(tables, RecTables):
        need = set(int(g) for g in np.unique(np.asarray(job_glen)))
        for g in need:
            t = tables.get(g)
            if t is None or tuple(t.shape) != (K, g):
                raise RcbError(f"missing / malformed candidate table for group length {g}")
        tables = RecTables.from_dict({g: tables[g] for g in need}, dev, K)
    else:
        for g in np.unique(np.asarray(job_glen)):
            if int(g) not in tables:
                raise RcbError(f"missing / malformed candidate table for group length {int(g)}")
    jobs, order = RecJobs.from_host(dev, job_row, job_start, job_glen, rows=rows, cols=cols, sort=not want_logw0)
    idx, best, _, logw0 = rec_score(loc, scale, p_loc, p_scale, tables, gumbel, jobs, mode, want_logw0)
    z = rec_commit(p_loc, p_scale, tables, jobs, idx, want_z=True)
    inv = torch.from_numpy(np.argsort(order)).to(dev)
    max_g = int(np.max(job_glen))
    return idx[inv], z[inv][:, :max_g].contiguous(), best[inv], logw0


# ----------------------------------------------------------------------------------------------
# N1: phase-form upsampling stages (rcb_upconv_*)
# ----------------------------------------------------------------------------------------------
bf16 = torch.bfloat16


def _xmode(x, preact):
    """0: bf16 activation, 1: fp32 pre-activation, 2: bf16 pre-activation"""
    if x.dtype == f32:
        return 1
    return 2 if preact else 0


def upconv_fwd(x, weff, bias, grid, cout, out_f32, preact=False, linear_bf16=False, pack=None):
    """x [B, g, g, 64] (fp32/bf16 pre-activation or bf16 activation) -> y [B, 2g, 2g, cout]: bf16 with LeakyReLU,
    fp32 linear (out_f32) or bf16 linear (linear_bf16)."""
    lib = _lib.load()
    B = x.shape[0]
    y = torch.empty(B, 2 * grid, 2 * grid, cout, device=x.device, dtype=f32 if out_f32 else bf16)
    omode = 1 if out_f32 else (2 if linear_bf16 else 0)
    check(lib.rcb_upconv_fwd(ptr(x), _xmode(x, preact), ptr(weff, f32), ptr(bias, f32), ptr(y), omode, B, grid,
                             cout, ptr(pack, None, True), stream_ptr()), "rcb_upconv_fwd")
    return y


def upconv_dgrad(dy, weff, x, grid, cout, preact=False, want_dbias=False, pack=None):
    """-> dx (dtype of x); with want_dbias (stage-2 geometry) also the [workgroups, 64] partial channel sums of dx."""
    lib = _lib.load()
    B = x.shape[0]
    dx = torch.empty_like(x)
    part = None
    if want_dbias:
        part = torch.empty(int(lib.rcb_upconv_dgrad_partial_rows(B)), 64, device=x.device, dtype=f32)
    check(lib.rcb_upconv_dgrad(ptr(dy), int(dy.dtype == f32), ptr(weff, f32), ptr(x), _xmode(x, preact), ptr(dx),
                               ptr(part, f32, True), B, grid, cout, ptr(pack, None, True), stream_ptr()), "rcb_upconv_dgrad")
    return (dx, part) if want_dbias else dx


def upconv_wgrad(x, dy, grid, cout, preact=False):
    """-> (dWeff [2,2,64,2,2,cout], dbias [cout]) in one pass over x and dy (deterministic: no atomics)."""
    lib = _lib.load()
    B = x.shape[0]
    buf = torch.empty(2 * 2 * 64 * 2 * 2 * cout + cout, device=x.device, dtype=f32)
    dw, db = buf[:-cout].view(2, 2, 64, 2, 2, cout), buf[-cout:]
    n_ws = int(lib.rcb_upconv_wgrad_workspace(B, cout))
    ws = torch.empty(n_ws, device=x.device, dtype=f32)
    check(lib.rcb_upconv_wgrad(ptr(x), _xmode(x, preact), ptr(dy), int(dy.dtype == f32), ptr(dw), ptr(db), B, grid,
                               cout, ptr(ws), C.c_int64(n_ws), stream_ptr()), "rcb_upconv_wgrad")
    return dw, db


def upconv_bwd_fused(dy, weff, x, grid, cout, pack=None):
    """stage-3 geometry, bf16 tensors: -> (dx, dWeff [2,2,64,2,2,cout], dbias [cout]) in one pass over dy and x"""
    lib = _lib.load()
    if dy.dtype != bf16 or x.dtype != bf16:
        raise RcbError("upconv_bwd_fused: bf16 tensors expected")
    B = x.shape[0]
    dx = torch.empty_like(x)
    buf = torch.empty(2 * 2 * 64 * 2 * 2 * cout + cout, device=x.device, dtype=f32)
    dw, db = buf[:-cout].view(2, 2, 64, 2, 2, cout), buf[-cout:]
    n_ws = int(lib.rcb_upconv_wgrad_workspace(B, cout))
    ws = torch.empty(n_ws, device=x.device, dtype=f32)
    check(lib.rcb_upconv_bwd_fused(ptr(dy), ptr(weff, f32), ptr(x), ptr(dx), ptr(dw), ptr(db), B, grid, cout, ptr(ws),
                                   C.c_int64(n_ws), ptr(pack, None, True), stream_ptr()), "rcb_upconv_bwd_fused")
    return dx, dw, db


# ---------------------------------------------------------------------------------------------------
# overlapping tiles of a stitched grid (rcb_tile_*): bf16 channel-last images
# ---------------------------------------------------------------------------------------------------
def tile_count(size, grid):
    """tiles per axis of an upconv stage with kernel grid `grid` over `size` source pixels (tiles overlap by one pixel)"""
    return -(-(2 * size + 1) // (2 * grid - 2))


def _tile_img(img):
    if img.dtype != bf16 or not img.is_cuda or img.dim() != 4 or not img.is_contiguous() or img.shape[-1] % 8:
        raise RcbError("tiles: contiguous bf16 GPU image [n, H, W, C] with C % 8 == 0 expected")
    return img.shape


def tile_gather(img, Ty, Tx, T, step, off, ring):
    """img [n,H,W,C] -> tiles [n*Ty*Tx, T, T, C] (rcb_tile_gather)"""
    n, H, W, Cc = _tile_img(img)
    tiles = torch.empty(n * Ty * Tx, T, T, Cc, device=img.device, dtype=bf16)
    check(_lib.load().rcb_tile_gather(ptr(img), ptr(tiles), n, H, W, Cc, Ty, Tx, T, step, off, ring, stream_ptr()), "rcb_tile_gather")
    return tiles


def _tile_out(tiles, n, H, W, Ty, Tx):
    if tiles.dtype != bf16 or not tiles.is_cuda or tiles.dim() != 4 or not tiles.is_contiguous() or tiles.shape[0] != n * Ty * Tx \
            or tiles.shape[1] != tiles.shape[2] or tiles.shape[-1] % 8:
        raise RcbError("tiles: contiguous bf16 GPU tiles [n*Ty*Tx, T, T, C] expected")
    return torch.empty(n, H, W, tiles.shape[-1], device=tiles.device, dtype=bf16)


def tile_crop(tiles, n, H, W, Ty, Tx, off):
    """tiles [n*Ty*Tx, T, T, C] -> img [n,H,W,C] from the inner T-2 rows / columns of every tile (rcb_tile_crop)"""
    img = _tile_out(tiles, n, H, W, Ty, Tx)
    check(_lib.load().rcb_tile_crop(ptr(tiles), ptr(img), n, H, W, tiles.shape[-1], Ty, Tx, tiles.shape[1], off, stream_ptr()),
          "rcb_tile_crop")
    return img


def tile_fold(tiles, n, H, W, Ty, Tx, off):
    """tiles [n*Ty*Tx, T, T, C] -> img [n,H,W,C], overlapping tile borders summed (rcb_tile_fold)"""
    img = _tile_out(tiles, n, H, W, Ty, Tx)
    check(_lib.load().rcb_tile_fold(ptr(tiles), ptr(img), n, H, W, tiles.shape[-1], Ty, Tx, tiles.shape[1], off, stream_ptr()),
          "rcb_tile_fold")
    return img


def _win_geo(shape):
    nd = len(shape) - 2
    if nd < 1 or nd > 3 or shape[-1] % 8:
        raise RcbError("windows: [B, *grid (1-3 axes), C] with C % 8 == 0 expected")
    g = list(shape[1:-1]) + [1] * (3 - nd)
    return nd, g


def window_gather(x):
    """x [B, *g, C] (bf16, contiguous) -> cols [B*prod(g), 3^d * C]: the 3^d-pixel window around every grid position"""
    if x.dtype != bf16 or not x.is_cuda or not x.is_contiguous():
        raise RcbError("window_gather: contiguous bf16 GPU tensor expected")
    nd, g = _win_geo(x.shape)
    cols = torch.empty(x.shape[0] * g[0] * g[1] * g[2], 3 ** nd * x.shape[-1], device=x.device, dtype=bf16)
    check(_lib.load().rcb_window_gather(ptr(x), ptr(cols), x.shape[0], g[0], g[1], g[2], x.shape[-1], nd, stream_ptr()),
          "rcb_window_gather")
    return cols


def window_fold(dcols, shape):
    """dcols [B*prod(g), 3^d * C] (bf16) -> dx of `shape` = [B, *g, C]: the adjoint of window_gather"""
    nd, g = _win_geo(shape)
    if dcols.dtype != bf16 or not dcols.is_cuda or not dcols.is_contiguous() or \
            tuple(dcols.shape) != (shape[0] * g[0] * g[1] * g[2], 3 ** nd * shape[-1]):
        raise RcbError("window_fold: contiguous bf16 GPU matrix [B*prod(g), 3^d*C] expected")
    dx = torch.empty(tuple(shape), device=dcols.device, dtype=bf16)
    check(_lib.load().rcb_window_fold(ptr(dcols), ptr(dx), shape[0], g[0], g[1], g[2], shape[-1], nd, stream_ptr()),
          "rcb_window_fold")
    return dx


def phase_bigweight(W, factors, k, pad, dtype=bf16):
    """conv weight W [Cout, Cin, *k] (fp32) -> the window-GEMM weight [3^d * Cin, prod(f) * Cout] of the stage
    nearest-upsample(f) -> conv(k, pad) (rcb_phase_bigweight), in `dtype` (bf16 / fp32)"""
    if W.dtype != f32 or not W.is_cuda or dtype not in (bf16, f32):
        raise RcbError("phase_bigweight: fp32 GPU weight, bf16 or fp32 result")
    nd, cout, cin = W.dim() - 2, W.shape[0], W.shape[1]
    wt = W.permute(*range(2, 2 + nd), 1, 0).contiguous()                       # [k^d, Cin, Cout]
    f = (C.c_int32 * 3)(*[int(v) for v in factors], *([0] * (3 - nd)))
    big = torch.empty(3 ** nd * cin, int(np.prod(factors)) * cout, device=W.device, dtype=dtype)
    check(_lib.load().rcb_phase_bigweight(ptr(wt), ptr(big, None, True), int(dtype == bf16), nd, f, int(k), int(pad), cin, cout,
                                          stream_ptr()), "rcb_phase_bigweight")
    return big


def phase_bigweight_grad(dbig, w_shape, factors, k, pad):
    """the adjoint of phase_bigweight: dbig (bf16 / fp32) -> dW of shape `w_shape` = [Cout, Cin, *k] (fp32)"""
    nd, cout, cin = len(w_shape) - 2, w_shape[0], w_shape[1]
    if dbig.dtype not in (bf16, f32) or not dbig.is_cuda or not dbig.is_contiguous() or \
            tuple(dbig.shape) != (3 ** nd * cin, int(np.prod(factors)) * cout):
        raise RcbError("phase_bigweight_grad: contiguous [3^d * Cin, prod(f) * Cout] GPU matrix expected")
    f = (C.c_int32 * 3)(*[int(v) for v in factors], *([0] * (3 - nd)))
    dwt = torch.empty(*([int(k)] * nd), cin, cout, device=dbig.device, dtype=f32)
    check(_lib.load().rcb_phase_bigweight_grad(ptr(dbig, None, True), int(dbig.dtype == bf16), ptr(dwt), nd, f, int(k), int(pad),
                                               cin, cout, stream_ptr()), "rcb_phase_bigweight_grad")
    return dwt.permute(nd + 1, nd, *range(nd)).contiguous()


# ---------------------------------------------------------------------------------------------------
# stage 1 of the 1-D upsampling net, direct (rcb_stage1_1d_*)
# ---------------------------------------------------------------------------------------------------
def _s1_check(x, who):
    if x.dim() != 3 or x.shape[-1] != 128 or x.dtype != f32 or not x.is_cuda or not x.is_contiguous():
        raise RcbError(f"{who}: contiguous fp32 GPU latent grid [B, g, 128] expected")


def stage1_1d_fwd(x, wbig, bias):
    """x [B, g, 128] (fp32), wbig = phase_bigweight(conv1.weight, [4], 5, 2) (bf16 [384, 256]), bias [64] (fp32)
    -> x1 [B, 4 g, 64] bf16 = LeakyReLU(bf16(conv1(up1(x)) + bias)) (rcb_stage1_1d_fwd)"""
    _s1_check(x, "stage1_1d_fwd")
    if tuple(wbig.shape) != (384, 256) or wbig.dtype != bf16 or not wbig.is_contiguous() or bias.numel() != 64:
        raise RcbError("stage1_1d_fwd: wbig bf16 [384, 256] and bias [64] expected")
    x1 = torch.empty(x.shape[0], 4 * x.shape[1], 64, device=x.device, dtype=bf16)
    check(_lib.load().rcb_stage1_1d_fwd(ptr(x, f32), ptr(wbig), ptr(bias.detach().float().contiguous(), f32), ptr(x1), x.shape[0], x.shape[1],
                                        stream_ptr()), "rcb_stage1_1d_fwd")
    return x1


_SCRATCH = {}


def _scratch(tag, n_floats, device):
    """Per-(device, stream) fp32 scratch of a kernel family, allocated once: the slab workspaces are sized for the largest
    grid (100 MB for the stage-1 weight gradient) and were allocated on every backward call -- inside a captured step that
    pinned one such buffer in the graph's private pool per model.  Work on one stream is ordered, so one buffer per stream is
    enough; a buffer first needed during a capture is taken from that capture's pool and not cached."""
    key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = torch.empty(n_floats, device=device, dtype=f32)
        if not torch.cuda.is_current_stream_capturing():
            _SCRATCH[key] = buf
    return buf


def stage1_1d_dgrad(dz, wbig):
    """dz [B, 4 g, 64] bf16 (gradient of the stage's PRE-activation) -> dx [B, g, 128] fp32 (rcb_stage1_1d_dgrad)"""
    if dz.dim() != 3 or dz.shape[-1] != 64 or dz.shape[1] % 4 or dz.dtype != bf16 or not dz.is_cuda or not dz.is_contiguous():
        raise RcbError("stage1_1d_dgrad: contiguous bf16 GPU tensor [B, 4 g, 64] expected")
    dx = torch.empty(dz.shape[0], dz.shape[1] // 4, 128, device=dz.device, dtype=f32)
    check(_lib.load().rcb_stage1_1d_dgrad(ptr(dz), ptr(wbig), ptr(dx, f32), dz.shape[0], dz.shape[1] // 4, stream_ptr()),
          "rcb_stage1_1d_dgrad")
    return dx


def stage1_1d_wgrad(x, dz):
    """-> (dwbig [384, 256] fp32: the gradient of phase_bigweight's result, zero where no phase reads; dbias [64] fp32)"""
    _s1_check(x, "stage1_1d_wgrad")
    if tuple(dz.shape) != (x.shape[0], 4 * x.shape[1], 64) or dz.dtype != bf16 or not dz.is_contiguous():
        raise RcbError("stage1_1d_wgrad: dz must be contiguous bf16 [B, 4 g, 64]")
    lib = _lib.load()
    n_ws = int(lib.rcb_stage1_1d_wgrad_workspace())
    ws = _scratch("stage1_1d_wgrad", n_ws, x.device)
    dwbig = torch.empty(384, 256, device=x.device, dtype=f32)
    db = torch.empty(64, device=x.device, dtype=f32)
    check(lib.rcb_stage1_1d_wgrad(ptr(x, f32), ptr(dz), ptr(dwbig, f32), ptr(db, f32), ptr(ws, f32), C.c_int64(n_ws), x.shape[0],
                                  x.shape[1], stream_ptr()), "rcb_stage1_1d_wgrad")
    return dwbig, db


# ---------------------------------------------------------------------------------------------------
# direct sub-pixel convolutions for grids of any dimension (rcb_phaseconv_*)
# ---------------------------------------------------------------------------------------------------
def _pc_geo(shape):
    nd = len(shape) - 2
    if nd < 1 or nd > 3 or shape[-1] != 64:
        raise RcbError("phaseconv: [B, *grid (1-3 axes), 64] expected")
    return nd, list(shape[1:-1]) + [1] * (3 - nd)


def phaseconv_pack(W, want_fwd=True, want_dgrad=True):
    """conv weight [Cout, 64, 3, ...] (fp32) -> (fwd fragments, dgrad fragments) as uint8 tensors (rcb_phaseconv_pack)"""
    lib = _lib.load()
    nd, cout = W.dim() - 2, W.shape[0]
    if W.shape[1] != 64 or any(k != 3 for k in W.shape[2:]) or cout not in (16, 64):
        raise RcbError("phaseconv_pack: conv weight [16 | 64, 64, 3, ...] expected")
    out = []
    for which, want in ((0, want_fwd), (1, want_dgrad)):
        n = int(lib.rcb_phaseconv_pack_uint4(nd, cout, which))
        out.append(torch.empty(n * 16, device=W.device, dtype=torch.uint8) if want else None)
    check(lib.rcb_phaseconv_pack(ptr(W.detach().contiguous(), f32), nd, cout, ptr(out[0], None, True), ptr(out[1], None, True),
                                 stream_ptr()), "rcb_phaseconv_pack")
    return out[0], out[1]


def phaseconv_fwd(x, frags, bias, cout, leaky_out):
    """x [B, *g, 64] bf16 activations -> y [B, *(2 g), cout] bf16 (rcb_phaseconv_fwd)"""
    if x.dtype != bf16 or not x.is_contiguous():
        raise RcbError("phaseconv_fwd: contiguous bf16 input expected")
    nd, g = _pc_geo(x.shape)
    y = torch.empty([x.shape[0]] + [2 * v for v in x.shape[1:-1]] + [cout], device=x.device, dtype=bf16)
    check(_lib.load().rcb_phaseconv_fwd(ptr(x), ptr(frags), ptr(bias.detach().contiguous(), f32), ptr(y), x.shape[0], g[0], g[1], g[2],
                                        nd, cout, int(bool(leaky_out)), stream_ptr()), "rcb_phaseconv_fwd")
    return y


def phaseconv_dgrad(dy, frags, x_act):
    """dy [B, *(2 g), cout] bf16, x_act [B, *g, 64] bf16 (stage input activations) -> dx [B, *g, 64] bf16, times LeakyReLU'(x_act)"""
    if dy.dtype != bf16 or not dy.is_contiguous() or x_act.dtype != bf16 or not x_act.is_contiguous():
        raise RcbError("phaseconv_dgrad: contiguous bf16 tensors expected")
    nd, g = _pc_geo(x_act.shape)
    if list(dy.shape[1:-1]) != [2 * v for v in x_act.shape[1:-1]] or dy.shape[0] != x_act.shape[0]:
        raise RcbError("phaseconv_dgrad: dy grid must be twice the input grid")
    dx = torch.empty_like(x_act)
    check(_lib.load().rcb_phaseconv_dgrad(ptr(dy), ptr(frags), ptr(x_act), ptr(dx), x_act.shape[0], g[0], g[1], g[2], nd, dy.shape[-1],
                                          stream_ptr()), "rcb_phaseconv_dgrad")
    return dx


def phaseconv_wgrad(x_act, dy):
    """x_act [B, *g, 64], dy [B, *(2 g), cout] (bf16) -> (dW [cout, 64, 3, ...] fp32, dbias [cout] fp32) (rcb_phaseconv_wgrad)"""
    lib = _lib.load()
    if dy.dtype != bf16 or not dy.is_contiguous() or x_act.dtype != bf16 or not x_act.is_contiguous():
        raise RcbError("phaseconv_wgrad: contiguous bf16 tensors expected")
    nd, g = _pc_geo(x_act.shape)
    cout = dy.shape[-1]
    if list(dy.shape[1:-1]) != [2 * v for v in x_act.shape[1:-1]] or dy.shape[0] != x_act.shape[0]:
        raise RcbError("phaseconv_wgrad: dy grid must be twice the input grid")
    n_ws = int(lib.rcb_phaseconv_wgrad_workspace(nd, cout))
    ws = _scratch("phaseconv_wgrad_%d_%d" % (nd, cout), n_ws, dy.device)
    dW = torch.empty([cout, 64] + [3] * nd, device=dy.device, dtype=f32)
    db = torch.empty(cout, device=dy.device, dtype=f32)
    check(lib.rcb_phaseconv_wgrad(ptr(x_act), ptr(dy), ptr(dW), ptr(db), ptr(ws), C.c_int64(n_ws), x_act.shape[0], g[0], g[1], g[2],
                                  nd, cout, stream_ptr()), "rcb_phaseconv_wgrad")
    return dW, db


UPCONV_PACK_UINT4 = 22528


def upconv_weff_build(W1, b1, W2, W3, bf16_out):
    """conv weights of the CIFAR-geometry upsampling net -> (Weff1 [512,4096], b1rep [4096], Weff2, Weff3, pack) where
    pack holds the stage-2/3 effective weights as pre-ordered bf16 MFMA fragments for upconv_fwd / upconv_dgrad."""
    lib = _lib.load()
    dev = W1.device
    if (tuple(W1.shape), tuple(W2.shape), tuple(W3.shape)) != ((64, 128, 5, 5), (64, 64, 3, 3), (16, 64, 3, 3)):
        raise RcbError("upconv_weff_build: conv weight shapes of the CIFAR upsampling net expected")
    dt = bf16 if bf16_out else f32
    weff1 = torch.empty(512, 4096, device=dev, dtype=dt)
    b1rep = torch.empty(4096, device=dev, dtype=dt)
    weff2 = torch.empty(2, 2, 64, 2, 2, 64, device=dev, dtype=f32)
    weff3 = torch.empty(2, 2, 64, 2, 2, 16, device=dev, dtype=f32)
    pack = torch.empty(UPCONV_PACK_UINT4 * 16, device=dev, dtype=torch.uint8)      # bf16 MFMA fragments (include/rcb.h)
    check(lib.rcb_upconv_weff_build(ptr(W1.detach(), f32), ptr(b1.detach(), f32), ptr(W2.detach(), f32),
                                    ptr(W3.detach(), f32), ptr(weff1), ptr(b1rep), int(bool(bf16_out)), ptr(weff2),
                                    ptr(weff3), ptr(pack), stream_ptr()), "rcb_upconv_weff_build")
    return weff1, b1rep, weff2, weff3, pack


def upconv_weff_grad(dweff1, dweff2, dweff3, db1_partial=None):
    """gradients of the effective weights -> (dW1 [64,128,5,5], dW2 [64,64,3,3], dW3 [16,64,3,3]) and, given the
    per-workgroup partials of upconv_dgrad(want_dbias=True), the stage-1 bias gradient db1 [64]."""
    lib = _lib.load()
    dev = dweff1.device
    if dweff1.numel() != 512 * 4096 or dweff2.numel() != 65536 or dweff3.numel() != 16384:
        raise RcbError("upconv_weff_grad: shape mismatch")
    dW1 = torch.empty(64, 128, 5, 5, device=dev, dtype=f32)
    dW2 = torch.empty(64, 64, 3, 3, device=dev, dtype=f32)
    dW3 = torch.empty(16, 64, 3, 3, device=dev, dtype=f32)
    db1 = torch.empty(64, device=dev, dtype=f32) if db1_partial is not None else None
    check(lib.rcb_upconv_weff_grad(ptr(dweff1), int(dweff1.dtype == bf16), ptr(dweff2, f32), ptr(dweff3, f32), ptr(dW1),
                                   ptr(dW2), ptr(dW3), ptr(db1_partial, f32, True),
                                   0 if db1_partial is None else db1_partial.shape[0], ptr(db1, f32, True), stream_ptr()),
          "rcb_upconv_weff_grad")
    return (dW1, dW2, dW3) if db1_partial is None else (dW1, dW2, dW3, db1)


class ATransform:
    """The A transform `wvec[:, lo:hi] = h_w[:, lo:hi] @ A[l]` (prior_model.py:173-174, test_model.py:348-349), its data
    gradient `dh[:, lo:hi] = dw[:, lo:hi] @ A[l]^T` and its weight gradient on the hand-written kernels of atrans.hip:
    all layers in one launch per direction, the per-row operand split hi + lo inside the kernel, the mappings as bf16
    (terms = 2) or hi + lo (terms = 3).  prepare(A) converts the mappings (once per step when they are trained)."""

    def __init__(self, slices, device, terms=2, dgrad_terms=None):
        if terms not in (1, 2, 3):
            raise RcbError("ATransform: terms must be 1, 2 or 3")
        self.terms = terms
        self.dgrad_terms = terms if dgrad_terms is None else dgrad_terms
        if self.dgrad_terms not in (1, 2, 3) or self.dgrad_terms > terms:
            raise RcbError("ATransform: dgrad_terms must be 1 .. terms")
        self.slices = list(slices)
        if self.slices[0][0] != 0 or any(a[1] != b[0] for a, b in zip(self.slices, self.slices[1:])):
            raise RcbError("ATransform: the layer vectors must tile the row without gaps")
        self.sizes = [hi - lo for lo, hi in self.slices]
        self.cols = self.slices[-1][1]
        self.device = torch.device(device)
        self._sizes_c = (C.c_int32 * len(self.sizes))(*self.sizes)
        n = _lib.load().rcb_atrans_pack_elems(len(self.sizes), self._sizes_c)
        if n <= 0:
            raise RcbError("ATransform: bad layer sizes")
        self.packed = torch.zeros(int(n), device=self.device, dtype=bf16)
        self._plans = {}
        self.n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count

    def prepare(self, A):
        mats = [a.detach() for a in A]
        for a, n in zip(mats, self.sizes):
            if tuple(a.shape) != (n, n) or a.dtype != f32 or not a.is_cuda or not a.is_contiguous():
                raise RcbError("ATransform.prepare: contiguous fp32 GPU matrices [L_l, L_l] expected")
        arr = (C.c_void_p * len(mats))(*[a.data_ptr() for a in mats])
        check(_lib.load().rcb_atrans_pack(arr, len(mats), self._sizes_c, ptr(self.packed), int(self.terms == 3), stream_ptr()),
              "rcb_atrans_pack")

    def _plan(self, rows):
        pl = self._plans.get(rows)
        if pl is None:
            lib = _lib.load()
            need = lib.rcb_atrans_plan(C.c_int64(rows), len(self.sizes), self._sizes_c, int(self.n_cu), None, 0)   # size query
            if need <= 0:
                check(need or -1, "rcb_atrans_plan")
            buf = (C.c_int32 * need)()
            n = lib.rcb_atrans_plan(C.c_int64(rows), len(self.sizes), self._sizes_c, int(self.n_cu), buf, need)
            if n != need:
                check(n if n < 0 else -1, "rcb_atrans_plan")
            host = torch.frombuffer(buf, dtype=torch.int32, count=n).clone()
            head = (C.c_int32 * 12)(*[int(v) for v in host[:12]])
            nws = int(lib.rcb_atrans_workspace_floats(C.c_int64(rows), len(self.sizes), self._sizes_c, head))
            pl = (host.to(self.device), head, nws)
            self._plans[rows] = pl
        return pl

    def _apply(self, x, out, transpose, terms):
        """x: fp32 rows, or a Planes pair (the producer's hi / lo split of the same rows: bit-identical result)"""
        planes = isinstance(x, Planes)
        rows = x.rows if planes else x.shape[0]
        for t_ in ((out,) if planes else (x, out)):
            if t_.dtype != f32 or not t_.is_cuda or t_.dim() != 2 or t_.stride(1) != 1 or t_.shape[1] != self.cols:
                raise RcbError(f"ATransform: fp32 GPU rows of {self.cols} columns expected")
        if planes and x.cols != self.cols:
            raise RcbError(f"ATransform: planes of {self.cols} columns expected")
        if out.shape[0] != rows:
            raise RcbError("ATransform: row counts differ")
        plan, head, nws = self._plan(rows)
        ws = torch.empty(nws, device=self.device, dtype=f32) if nws else None      # (few rows: slabs of the K-split launch)
        if planes:
            xa = (C.c_void_p(0), C.c_int64(0), C.c_void_p(x.buf[0].data_ptr()), C.c_void_p(x.buf[1].data_ptr()), C.c_int64(x.ld))
        else:
            xa = (C.c_void_p(x.data_ptr()), C.c_int64(x.stride(0)), C.c_void_p(0), C.c_void_p(0), C.c_int64(0))
        check(_lib.load().rcb_atrans_apply(*xa, C.c_void_p(out.data_ptr()), C.c_int64(out.stride(0)), C.c_int64(rows),
                                          len(self.sizes), self._sizes_c, ptr(self.packed), int(transpose), int(terms),
                                          ptr(plan), head, ptr(ws, f32, True), stream_ptr()), "rcb_atrans_apply")
        return out

    def forward(self, h_w, out):
        """out[:, lo:hi] = h_w[:, lo:hi] @ A[l]"""
        return self._apply(h_w, out, 0, self.terms)

    def dgrad(self, dw, out):
        """out[:, lo:hi] = dw[:, lo:hi] @ A[l]^T"""
        return self._apply(dw, out, 1, self.dgrad_terms)

    def new_rows(self, rows):
        """fp32 [rows, cols] view whose rows start on 128-byte lines (row stride a multiple of 32 floats): the layout the
        kernels stream fastest.  The SIREN kernels take the stride (rcb_siren_desc.w_row_stride) and give it to dwvec."""
        ld = (self.cols + 31) // 32 * 32
        return torch.empty(rows, ld, device=self.device, dtype=f32)[:, :self.cols]

    def wgrad(self, h_w, dw, h16=None, dw16=None, bf16_hi=True, part="all"):
        """dA[l] = h_w[:, lo:hi]^T @ dw[:, lo:hi], summed over the rows (INRs x samples).  With bf16_hi the widest layers
        (adjacent, one size, a multiple of 8) take bf16 operands in one batched GEMM with fp32 accumulation: the sum over
        thousands of rows averages the unbiased operand rounding down, unlike the per-row products of forward / dgrad.
        h16 / dw16: bf16 copies of h_w / dw ([rows, cols], row stride a multiple of 8) written by their producers
        (reparam_rng(want_bf16), siren_loss_bwd(want_bf16)); cast here when absent.  Narrow layers (the output layer):
        rcb_atrans_wgrad_narrow, exact fp32 products in a fixed order; anything else: fp32 GEMMs."""
        hp, dp = isinstance(h_w, Planes), isinstance(dw, Planes)
        if hp:                      # operands as planes: hi is the bf16 copy, the narrow layers read float(hi) + float(lo)
            h16 = h_w.hi
        if dp:
            dw16 = dw.hi
        if (hp or dp) and not bf16_hi:
            h_w, dw, hp, dp = (h_w.float() if hp else h_w), (dw.float() if dp else dw), False, False
        # part: "all", or -- for callers that run the two halves on different streams -- "wide" (the batched GEMM of the widest
        # layers: reads only the bf16 copies) / "rest" (everything else: reads the fp32 rows); entries of the other half are None
        out = [None] * len(self.sizes)
        big = max(self.sizes)
        wide = [i for i, n in enumerate(self.sizes) if n == big]
        adjacent = all(b == a + 1 for a, b in zip(wide, wide[1:]))
        rows = h_w.rows if hp else h_w.shape[0]
        wide_ok = bool(bf16_hi and big >= 256 and big % 8 == 0 and adjacent and self.slices[wide[0]][0] % 8 == 0)
        if part == "wide" and not wide_ok:
            return out
        if wide_ok and part != "rest":
            k, lo0 = len(wide), self.slices[wide[0]][0]

            def view(t16, t32):
                if t16 is None or t16.stride(1) != 1 or t16.stride(0) % 8:
                    t16 = torch.empty(rows, (self.cols + 7) // 8 * 8, device=self.device, dtype=bf16)[:, :self.cols]
                    t16.copy_(t32)
                return t16.as_strided((k, rows, big), (big, t16.stride(0), 1), t16.storage_offset() + lo0)

            g = torch.bmm(view(h16, h_w).transpose(1, 2), view(dw16, dw), out_dtype=f32)
            for j, i in enumerate(wide):
                out[i] = g[j]
        if part == "wide":
            return out
        lib = _lib.load()
        for i, (lo, hi) in enumerate(self.slices):
            if out[i] is not None or (part == "rest" and wide_ok and i in wide):
                continue
            n = hi - lo
            if n <= 256:
                slabs = max(1, min(64, (rows + 63) // 64))
                ws = torch.empty(int(lib.rcb_atrans_wgrad_narrow_workspace(n, slabs)), device=self.device, dtype=f32)
                g = torch.empty(n, n, device=self.device, dtype=f32)
                def operand(t, is_planes):
                    if is_planes:
                        return (C.c_void_p(0), C.c_void_p(t.buf[0, :, lo:hi].data_ptr()), C.c_void_p(t.buf[1, :, lo:hi].data_ptr()),
                                C.c_int64(t.ld))
                    return (C.c_void_p(t[:, lo:hi].data_ptr()), C.c_void_p(0), C.c_void_p(0), C.c_int64(t.stride(0)))
                check(lib.rcb_atrans_wgrad_narrow(*operand(h_w, hp), *operand(dw, dp), C.c_int64(rows), n, ptr(g), ptr(ws), slabs,
                                                  stream_ptr()), "rcb_atrans_wgrad_narrow")
                out[i] = g
            else:
                hf = h_w.float() if hp else h_w
                df = dw.float() if dp else dw
                out[i] = torch.mm(hf[:, lo:hi].t(), df[:, lo:hi])
        return out
