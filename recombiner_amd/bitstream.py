"""N4: an auditable bitstream and a standalone decoder.

The reference writes the A* indices as a CSV of float-formatted numbers (main_compression.py:169-178) and has no
decoder: "decoding" is `predict` on the encoder's own object with every group masked.  Here the indices of all levels
are packed at ceil(log2 K) = 16 bits per (row, group) -- exactly the rate the reference's bpp formula charges
(test_model.py:245-250) -- and `decode()` rebuilds the signal from the prior checkpoint and the bitstream alone:

    z[row, start_g:end_g] = p_loc + p_scale * sobol_normal_table_g[idx[row, g]]      (test_model.py:501-533)

evaluated like the scoring kernel (fp64 multiply, then add, rounded to fp32 when committed), so the decoder's
parameters equal the encoder's bit for bit.

Container (little endian):  b"RCB1" | u16 version | u8 n_levels | u8 bits | per level: u32 rows, u32 n_groups |
payload: for each level (1, then 2 and 3 for patched presets) rows x n_groups uint16 indices, row-major | u32 CRC-32 of
the payload."""
import struct
import zlib

import numpy as np
import torch

MAGIC = b"RCB1"
VERSION = 1


def pack_indices(levels, bits=16):
    """levels: list of integer arrays [rows, n_groups] with values < 2**bits  ->  bytes."""
    if bits != 16:
        raise ValueError("only 16 bits per group are defined (K = 65536 candidates, test_model.py:441-444)")
    if not 1 <= len(levels) <= 3:
        raise ValueError("1 to 3 levels expected")
    head = MAGIC + struct.pack("<HBB", VERSION, len(levels), bits)
    body = b""
    for a in levels:
        a = np.asarray(a)
        if a.ndim != 2:
            raise ValueError("index arrays must be [rows, n_groups]")
        if a.size and (a.min() < 0 or a.max() >= (1 << bits) or np.any(a != np.floor(a))):
            raise ValueError("index out of range for %d bits" % bits)
        head += struct.pack("<II", a.shape[0], a.shape[1])
        body += np.ascontiguousarray(a.astype("<u2")).tobytes()
    return head + body + struct.pack("<I", zlib.crc32(body) & 0xFFFFFFFF)


def unpack_indices(blob):
    """bytes -> list of int64 arrays [rows, n_groups]; raises ValueError on a malformed or corrupted stream."""
    if len(blob) < 12 or blob[:4] != MAGIC:
        raise ValueError("not an RCB1 bitstream")
    version, n_levels, bits = struct.unpack_from("<HBB", blob, 4)
    if version != VERSION or bits != 16 or not 1 <= n_levels <= 3:
        raise ValueError("unsupported bitstream header (version %d, %d levels, %d bits)" % (version, n_levels, bits))
    off = 8
    shapes = []
    for _ in range(n_levels):
        if off + 8 > len(blob):
            raise ValueError("truncated header")
        shapes.append(struct.unpack_from("<II", blob, off))
        off += 8
    n_bytes = sum(r * g for r, g in shapes) * 2
    if len(blob) != off + n_bytes + 4:
        raise ValueError("bitstream length %d does not match its header (%d expected)" % (len(blob), off + n_bytes + 4))
    body = blob[off:off + n_bytes]
    (crc,) = struct.unpack_from("<I", blob, off + n_bytes)
    if zlib.crc32(body) & 0xFFFFFFFF != crc:
        raise ValueError("bitstream checksum mismatch")
    out, p = [], 0
    for r, g in shapes:
        out.append(np.frombuffer(body, dtype="<u2", count=r * g, offset=p).reshape(r, g).astype(np.int64))
        p += r * g * 2
    return out


def payload_bits(blob):
    """bits spent on indices (what bpp is computed from; header and checksum are per-file constants)."""
    return sum(a.size for a in unpack_indices(blob)) * 16


def _levels_of(model):
    return [model._l1] + ([model._l2, model._l3] if model.patch else [])


def encode(model):
    """bitstream of a fully compressed TestBNNmodel (every group of every level encoded)."""
    lvls = _levels_of(model)
    for lv in lvls:
        if not lv.mask_groupwise.all():
            raise ValueError("level %r has %d groups that are not encoded yet" % (lv.pre, int((~lv.mask_groupwise).sum())))
    return pack_indices([lv.idx_groupwise for lv in lvls])


def apply_indices(model, levels):
    """Rebuild the encoded samples of every level of `model` from index arrays (decoder side): afterwards every group
    is marked encoded and `model.predict` reconstructs the signal."""
    from . import ops
    lvls = _levels_of(model)
    if len(levels) != len(lvls):
        raise ValueError("bitstream has %d levels, the model %d" % (len(levels), len(lvls)))
    K = int(np.ceil(2 ** model.bit_per_group))
    for lv, idx in zip(lvls, levels):
        if tuple(idx.shape) != (lv.rows, lv.n_groups):
            raise ValueError("level %r: index array %s, model needs %s" % (lv.pre, idx.shape, (lv.rows, lv.n_groups)))
        if idx.min() < 0 or idx.max() >= K:
            raise ValueError("level %r: index outside [0, %d)" % (lv.pre, K))
        dev = lv.loc.device
        tables = model._rec_tables(lv, lv.end - lv.start, K)
        # every (row, group) pair is one job of the encoder's own commit kernel (rcb_rec_commit): fp64 multiply, then add,
        # rounded to fp32 -- the decoder's parameters are the encoder's bit for bit by construction
        rows = np.repeat(np.arange(lv.rows), lv.n_groups)
        groups = np.tile(np.arange(lv.n_groups), lv.rows)
        jobs, order = ops.RecJobs.from_host(dev, rows, lv.start[groups], (lv.end - lv.start)[groups], group=groups,
                                            rows=lv.rows, cols=lv.D, sort=False)
        idx_t = torch.from_numpy(np.ascontiguousarray(idx.reshape(-1)).astype(np.int32)).to(dev)
        ops.rec_commit(lv.p_loc, ops.softplus_scale(lv.p_log_scale), tables, jobs, idx_t, n_groups=lv.n_groups,
                       enc_sample=lv.sample, enc_mask=lv.mask, done=lv.d_done, beta=lv.kl_beta, idx_groupwise=lv.d_idx)


def decode(config, dataset, checkpoint, blob, x, n_datapoints, device="cuda", seed=42, precision=0):
    """Standalone decoder: prior checkpoint (drivers.load_checkpoint) + bitstream -> reconstruction [N, P, C].
    x: the coordinate features the encoder used ([P, F] or [N, P, F]); n_datapoints: INRs (patches) in the stream."""
    from .drivers import build_test_model
    levels = unpack_indices(blob)
    if levels[0].shape[0] != n_datapoints:
        raise ValueError("bitstream holds %d rows, %d datapoints requested" % (levels[0].shape[0], n_datapoints))
    model = build_test_model(config, dataset, checkpoint, n_datapoints, device, seed)
    model.precision = precision
    apply_indices(model, levels)
    with torch.no_grad():
        return model.predict(x.to(device))
