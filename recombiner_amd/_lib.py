"""ctypes binding of librcb_hip.so (include/rcb.h).  The product has no CPU fallback: if the
library is missing or a call fails, an exception is raised."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RCB_LIB: alternative build of the same ABI (same-box A/B timing of kernel changes); never a different backend
LIB_PATH = os.environ.get("RCB_LIB") or os.path.join(_HERE, "lib", "librcb_hip.so")

EXPORTS = ["rcb_version", "rcb_last_error_string", "rcb_struct_bytes", "rcb_siren_fwd", "rcb_siren_bwd", "rcb_siren_loss_bwd",
           "rcb_reparam_fwd", "rcb_gauss_kl", "rcb_beta_update", "rcb_posterior_bwd", "rcb_adam_flat",
           "rcb_col_moments", "rcb_rec_score_argmax", "rcb_rec_commit", "rcb_rec_workspace_bytes", "rcb_softplus_scale", "rcb_gauss_kl_colsum", "rcb_upconv_fwd",
           "rcb_upconv_dgrad", "rcb_upconv_wgrad", "rcb_upconv_wgrad_workspace", "rcb_adam_multi", "rcb_step_begin",
           "rcb_step_end", "rcb_upconv_weff_build", "rcb_upconv_weff_grad",
           "rcb_upconv_dgrad_partial_rows", "rcb_debug_generic_kernels_only", "rcb_philox_normal",
           "rcb_reparam_rng_fwd", "rcb_upconv_bwd_fused", "rcb_tile_gather", "rcb_tile_crop", "rcb_tile_fold",
           "rcb_window_gather", "rcb_window_fold", "rcb_siren_reduce_chunks", "rcb_phaseconv_pack", "rcb_phaseconv_pack_uint4",
           "rcb_phaseconv_fwd", "rcb_phaseconv_dgrad", "rcb_phaseconv_wgrad", "rcb_phaseconv_wgrad_workspace",
           "rcb_phase_bigweight", "rcb_phase_bigweight_grad", "rcb_atrans_pack_elems", "rcb_atrans_pack", "rcb_atrans_plan",
           "rcb_atrans_apply", "rcb_atrans_workspace_floats", "rcb_atrans_wgrad_narrow_workspace", "rcb_atrans_wgrad_narrow",
           "rcb_stage1_1d_fwd", "rcb_stage1_1d_dgrad", "rcb_stage1_1d_wgrad", "rcb_stage1_1d_wgrad_workspace",
           "rcb_reparam_hier_rng_fwd", "rcb_debug_siren_wave_tiles"]


class RcbError(RuntimeError):
    pass


class SirenDesc(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("samples", C.c_int32), ("n_pix", C.c_int32), ("fourier_dim", C.c_int32),
                ("pe_dim", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32),
                ("xf_inr_stride", C.c_int64), ("w_row_stride", C.c_int64), ("w0", C.c_float),
                ("precision", C.c_int32), ("pe_bf16", C.c_int32), ("dw_bf16", C.c_void_p), ("pixel_chunks", C.c_int32),
                ("xf_bf16", C.c_void_p), ("pe_grid_dims", C.c_int32), ("pe_patch_nums", C.c_int32 * 3),
                ("pe_patch_size", C.c_int32 * 3), ("dw_bf16_stride", C.c_int64), ("hidden_dims", C.c_int32 * 4),
                ("dw_lo", C.c_void_p), ("clock_probe", C.c_void_p)]


class Level(C.Structure):
    _fields_ = [("loc", C.c_void_p), ("log_scale", C.c_void_p), ("enc_sample", C.c_void_p),
                ("enc_mask", C.c_void_p), ("row_map", C.c_void_p), ("row_perm", C.c_void_p),
                ("col_map", C.c_void_p), ("eps", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("cols_out", C.c_int32), ("scale_is_sigma", C.c_int32), ("mu_sigma_ws", C.c_void_p)]


class RecDesc(C.Structure):
    _fields_ = [("loc", C.c_void_p), ("scale", C.c_void_p), ("p_loc", C.c_void_p), ("p_scale", C.c_void_p),
                ("rows", C.c_int32), ("cols", C.c_int32), ("tables_t", C.c_void_p), ("table_absmax", C.c_void_p),
                ("max_glen", C.c_int32), ("gumbel", C.c_void_p), ("gumbel_absmax", C.c_double),
                ("n_candidates", C.c_int32), ("job_row", C.c_void_p), ("job_start", C.c_void_p),
                ("job_glen", C.c_void_p), ("n_jobs", C.c_int32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("n", C.c_int64)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int32), ("dyn_scalars", C.c_void_p)]


class LevelBwd(C.Structure):
    _fields_ = [("loc", C.c_void_p), ("log_scale", C.c_void_p), ("enc_mask", C.c_void_p), ("p_loc", C.c_void_p),
                ("p_scale", C.c_void_p), ("p_scale_is_log", C.c_int32), ("beta", C.c_void_p),
                ("group_idx", C.c_void_p), ("n_groups", C.c_int32), ("kl_scalar", C.c_float),
                ("d_out", C.c_void_p), ("eps", C.c_void_p), ("member_ptr", C.c_void_p),
                ("member_idx", C.c_void_p), ("row_perm_inv", C.c_void_p), ("col_inv", C.c_void_p),
                ("rows", C.c_int32), ("cols", C.c_int32), ("cols_out", C.c_int32), ("samples", C.c_int32),
                ("g_loc", C.c_void_p), ("g_log_scale", C.c_void_p), ("m_loc", C.c_void_p), ("v_loc", C.c_void_p),
                ("m_ls", C.c_void_p), ("v_ls", C.c_void_p), ("kl_accum", C.c_void_p), ("kl_scalar_dev", C.c_void_p),
                ("next_out", C.c_void_p), ("next_eps", C.c_void_p), ("next_out_bf16", C.c_void_p),
                ("next_ld_bf16", C.c_int64), ("rng_seed", C.c_uint64), ("rng_step_dev", C.c_void_p),
                ("rng_step_add", C.c_int64), ("rng_stream", C.c_uint32), ("eps_from_rng", C.c_uint32),
                ("rng_group_offset", C.c_uint64), ("next_out_lo", C.c_void_p), ("sample_sum_ws", C.c_void_p), ("col_map", C.c_void_p)]


_lib = None
# the header these mirrors were written against (include/rcb.h: RCB_VERSION) and the structures load() verifies by size
ABI_VERSION = 406
_MIRRORS = {0: SirenDesc, 1: Level, 2: LevelBwd, 3: AdamCfg, 4: AdamTensor, 5: RecDesc}


def load():
    """dlopen the in-tree library (raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RcbError(f"{LIB_PATH} not found: run `python -m recombiner_amd.build` (no CPU fallback exists)")
        lib = C.CDLL(LIB_PATH)
        lib.rcb_last_error_string.restype = C.c_char_p
        lib.rcb_upconv_wgrad_workspace.restype = C.c_int64
        lib.rcb_rec_workspace_bytes.restype = C.c_int64
        lib.rcb_phaseconv_pack_uint4.restype = C.c_int64
        lib.rcb_phaseconv_wgrad_workspace.restype = C.c_int64
        lib.rcb_stage1_1d_wgrad_workspace.restype = C.c_int64
        lib.rcb_atrans_pack_elems.restype = C.c_int64
        lib.rcb_atrans_wgrad_narrow_workspace.restype = C.c_int64
        lib.rcb_atrans_workspace_floats.restype = C.c_int64
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RcbError(f"{LIB_PATH} does not export {name}")
        lib.rcb_struct_bytes.restype = C.c_int64
        if lib.rcb_version() != ABI_VERSION:
            raise RcbError(f"{LIB_PATH} is ABI version {lib.rcb_version()}, this binding was written against {ABI_VERSION}: "
                           "rebuild with `python -m recombiner_amd.build --force`")
        for which, cls in _MIRRORS.items():
            if lib.rcb_struct_bytes(which) != C.sizeof(cls):
                raise RcbError(f"{cls.__name__}: the library's structure has {lib.rcb_struct_bytes(which)} bytes, the ctypes "
                               f"mirror {C.sizeof(cls)} -- _lib.py has drifted from include/rcb.h")
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().rcb_last_error_string().decode(errors="replace")
        raise RcbError(f"{what} failed with code {rc}: {msg}")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, allow_none=False):
    """device pointer of a contiguous CUDA tensor (or NULL when allowed)."""
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise RcbError("null tensor")
    if not t.is_cuda:
        raise RcbError("tensor must live on the GPU (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise RcbError("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RcbError(f"expected {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def addr(t, dtype=None):
    """integer address for structure fields (None -> NULL)."""
    if t is None:
        return None
    return ptr(t, dtype).value
