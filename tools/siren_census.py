"""Instruction census of the tile loop of the production SIREN kernel (width 32, bf16 operands, fused loss + backward,
16-bit inputs) from the compiler's own assembly, priced with the issue costs of MI355X_MICROARCH.md (per wave-instruction on
one SIMD: transcendental 8 cycles, other VALU 4, v_cvt_pk_bf16_f32 4.5, an MFMA holds the vector issue for 8 of its 32, LDS /
VMEM / scalar 4 each as an upper bound on their issue slots).  Runs in the build container (hipcc cross-compiles):
    python tools/siren_census.py            -> profiles/r04_siren_isa_census.json, stamped with the hash of the kernel sources
bench.py reads that file for the `valu` entry of its roofline object (the VALU-issue floor of the kernel at the shader clock
measured in the same run)."""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recombiner_amd", "csrc")
SYMBOL = "siren_bf16_kernelIDF16bLi3ELi16ELi16ELi3ELi2ELb1EEE"        # <__bf16, NH 3, F 16, E 16, C 3, MODE_LOSS, IN16>


def source_sha16():
    h = hashlib.sha256()
    for f in ("siren_mlp_bf16.hip", "siren_common.h", "siren_op16.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "siren16.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S",
                               "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "siren_mlp_bf16.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL, cwd=td)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and SYMBOL in l and l.rstrip().split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the tile loop = the innermost loop that holds the transcendentals
    head = None
    for i, l in enumerate(body):
        m = re.match(r"^\.L(BB\d+_\d+):.*Loop Header", l)
        if m:
            cand = m.group(1)
            j = i + 1
            while j < len(body) and not (body[j].startswith(".LBB") and ("Header=" + cand) not in body[j]):
                j += 1
            if any("v_sin_f32" in b for b in body[i:j]):
                head = (i, j)
    assert head, "tile loop not found"
    ops = collections.Counter()
    for l in body[head[0]:head[1]]:
        t = l.strip().split()
        if t and not t[0].startswith((";", ".")):
            ops[t[0]] += 1
    cat = collections.Counter()
    for k, v in ops.items():
        if "mfma" in k:
            cat["mfma"] += v
        elif k.startswith(("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
            cat["transcendental"] += v
        elif k.startswith("v_cvt_pk"):
            cat["cvt_pk"] += v
        elif k.startswith("v_"):
            cat["valu_other"] += v
        elif k.startswith("ds_"):
            cat["lds"] += v
        elif k.startswith(("global_", "buffer_", "scratch_", "flat_")):
            cat["vmem"] += v
        else:
            cat["scalar_and_waits"] += v
    cost = {"transcendental": 8.0, "valu_other": 4.0, "cvt_pk": 4.5, "mfma": 8.0}
    valu_cycles = sum(cat[k] * c for k, c in cost.items())
    out = {"kernel": "siren_bf16_kernel<bf16, 3 hidden, F 16, E 16, C 3, loss + backward, 16-bit inputs> (rcb_siren_loss_bwd, BASELINE configs[1])",
           "source_sha16": source_sha16(), "unit": "per 32-pixel tile and wave (one trip of the tile loop)",
           "instructions": int(sum(ops.values())), "by_class": dict(cat), "issue_cycles_per_class": cost,
           "vector_issue_cycles_per_tile": valu_cycles,
           "transcendental_cycles_per_tile": cat["transcendental"] * 8.0,
           "scratch_instructions": int(sum(v for k, v in ops.items() if k.startswith("scratch_"))),
           "top": sorted(ops.items(), key=lambda kv: -kv[1])[:16],
           "note": "vector-issue cycles = sum over VALU + MFMA instructions of their issue cost on the SIMD (MI355X_MICROARCH.md, "
                   "per-instruction cycle constants); the floor of the kernel = tiles per SIMD x these cycles / shader clock"}
    path = os.path.join(ROOT, "profiles", "r04_siren_isa_census.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
