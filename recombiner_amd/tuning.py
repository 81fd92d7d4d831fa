"""Selection of library-GEMM solutions for the dense fp32 GEMMs that stay on hipBLASLt/rocBLAS (the A transform
and stage 1 of the upsampling net).  `tools/tune_gemms.py` runs PyTorch's TunableOp once on an MI355X and stores
the chosen solutions in `recombiner_amd/tuned/gemm_mi355x.csv`; `enable_tuned_gemms()` only *looks them up*
(no tuning at run time).  A missing or non-matching file leaves the library defaults in place."""
import os

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_mi355x.csv")
_tuning_to = None      # set while tools/tune_gemms.py is recording: later lookup requests must not switch it off


def enable_tuned_gemms(path=TUNED_FILE, tune=False):
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tun
    global _tuning_to
    if _tuning_to is not None:
        return True
    if tune:
        _tuning_to = path
        tun.enable(True)
        tun.tuning_enable(True)
        tun.set_max_tuning_duration(300)
        tun.set_max_tuning_iterations(50)
        tun.set_filename(path)
        return True
    if not os.path.exists(path):
        return False
    try:
        tun.enable(True)
        tun.tuning_enable(False)
        ok = tun.read_file(path)
        if not ok:
            tun.enable(False)
        return bool(ok)
    except Exception:
        try:
            tun.enable(False)
        except Exception:
            pass
        return False
