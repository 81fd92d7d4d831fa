"""Instruction census of the tile loops of the SIREN loss / backward kernels from the compiler's own assembly, priced with the
per-pipe costs of MI355X_MICROARCH.md.  Runs in the build container (hipcc cross-compiles):

    python tools/siren_census.py            -> profiles/r05_siren_isa_census.json, stamped with the hash of the kernel sources
    python tools/siren_census.py --check    -> exit code 1 if an instance the training step selects (wave family, 16-bit-input-row
                                               instances) has a scratch instruction inside its tile loop; __graft_entry__.build() runs it

Per wave and 32-pixel tile (one trip of the innermost loop that holds the transcendentals):
  * vector issue  : transcendental 8 cycles, v_cvt_pk 4.5, other VALU 4, an MFMA holds the issue for 8 of its 32 -- the cost of ONE
                    wave's stream; two waves of a SIMD interleave, so the SIMD-level floor is between half of this and this
  * matrix pipe   : 32 cycles per v_mfma_f32_32x32x16, per SIMD
  * transcendental: 8 cycles per instruction on the quarter-rate unit, per SIMD
  * LDS           : per CU (four SIMDs share it): ds_write_b64 6, ds_write_b128 13, ds_read_b128 4, ds_read_b64 / _tr_b16 / b32 2
bench.py reads the headline entry for the `valu` object of its roofline (floors at the shader clock measured in the same run).
The `instances` table lists every loss / backward instance of the three 16-bit SIREN kernel files with its scratch bytes and the
number of scratch instructions INSIDE its tile loop (a reload there drains vmcnt: it waits for the next tile's input prefetch)."""
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
FILES = ("siren_mlp_wave.hip", "siren_mlp_bf16.hip", "siren_mlp_wide.hip")
# the headline instance: siren_wave_kernel<__bf16, 3 hidden, F 16, E 16, C 3, MODE_LOSS, dpe written>
HEADLINE = ("siren_mlp_wave.hip", "siren_wave_kernelIDF16bLi3ELi16ELi16ELi3ELi2ELb1EEE")
ISSUE = {"transcendental": 8.0, "valu_other": 4.0, "cvt_pk": 4.5, "mfma": 8.0}
LDS_COST = (("ds_write_b64", 6), ("ds_write_b128", 13), ("ds_write2", 13), ("ds_write_b32", 4), ("ds_read_b128", 4), ("ds_read", 2), ("ds_", 4))


def source_sha16():
    h = hashlib.sha256()
    for f in ("siren_mlp_wave.hip", "siren_mlp_bf16.hip", "siren_mlp_wide.hip", "siren_common.h", "siren_op16.h"):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def compile_asm(src, td):
    asm = os.path.join(td, src + ".s")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S",
                          "-Rpass-analysis=kernel-resource-usage", "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, src), "-o", asm],
                         stderr=subprocess.PIPE, cwd=td, text=True, check=True)
    res, name = {}, None
    for l in out.stderr.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            name = m.group(1)
            res[name] = {}
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch_bytes", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, l)
            if m and name:
                res[name][key] = int(m.group(1))
    return open(asm).read().split("\n"), res


def kernel_bodies(lines):
    """-> {mangled name: body lines} for every kernel in an assembly file"""
    out, i = {}, 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_Z") and ":" in l and not l.startswith("_ZN3rcb"):
            name = l.split(":")[0].strip()
            j = i
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            if j < len(lines) and any("amdhsa" in x or "s_load" in x or "v_" in x for x in lines[i:min(j, i + 50)]):
                out[name] = lines[i:j + 1]
            i = j
        i += 1
    return out


def tile_loop(body):
    """the innermost loop that holds the transcendentals -> its lines (from its header label to the next label outside it)"""
    best = None
    for i, l in enumerate(body):
        m = re.match(r"^\.L(BB\d+_\d+):", l)
        # (the "Loop Header" remark sits on the label's line or on the comment line(s) right behind it)
        if not m or not any("Loop Header" in x for x in [l] + [b for b in body[i + 1:i + 3] if b.lstrip().startswith(";")]):
            continue
        cand, j = m.group(1), i + 1
        while j < len(body) and not (body[j].startswith(".LBB") and ("Header=" + cand) not in body[j]):
            j += 1
        if any("v_sin_f32" in b or "v_cos_f32" in b for b in body[i:j]):
            best = (i, j)
    if best is None:
        return None
    seg = body[best[0]:best[1]]
    # instructions behind the loop's back edge (reloads of the exit path) belong to the code after the loop
    last_branch = max((k for k, b in enumerate(seg) if re.search(r"s_cbranch_\w+\s+\.L" + re.match(r"^\.L(BB\d+_\d+)", seg[0]).group(1) + r"\b", b)),
                      default=None)
    if last_branch is not None:
        tail = seg[last_branch + 1:]
        if tail and not any(t.startswith(".LBB") for t in tail):
            seg = seg[:last_branch + 1]
    return seg


def census(seg):
    ops = collections.Counter()
    for l in seg:
        t = l.strip().split()
        if t and not t[0].startswith((";", ".")) and not t[0].endswith(":"):
            ops[t[0]] += 1
    cat = collections.Counter()
    lds_cycles = 0
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
            lds_cycles += v * next(c for p, c in LDS_COST if k.startswith(p))
        elif k.startswith("scratch_"):
            cat["scratch"] += v
        elif k.startswith(("global_", "buffer_", "flat_")):
            cat["vmem"] += v
        else:
            cat["scalar_and_waits"] += v
    issue = sum(cat[k] * c for k, c in ISSUE.items())
    return {"instructions": int(sum(ops.values())), "by_class": dict(cat), "vector_issue_cycles_per_tile": issue,
            "transcendental_cycles_per_tile": cat["transcendental"] * 8.0, "mfma_pipe_cycles_per_tile": cat["mfma"] * 32.0,
            "lds_cycles_per_tile_and_wave": float(lds_cycles), "scratch_instructions_in_tile_loop": int(cat["scratch"]),
            "top": sorted(ops.items(), key=lambda kv: -kv[1])[:14]}


def main():
    check = "--check" in sys.argv
    instances, headline = [], None
    with tempfile.TemporaryDirectory() as td:
        import concurrent.futures
        with concurrent.futures.ThreadPoolExecutor(len(FILES)) as pool:       # (three hipcc processes side by side)
            compiled = list(pool.map(lambda f: compile_asm(f, td), FILES))
        for src, (lines, res) in zip(FILES, compiled):
            for name, body in kernel_bodies(lines).items():
                if "siren" not in name:
                    continue
                seg = tile_loop(body)
                if seg is None:
                    continue
                c = census(seg)
                # loss / backward instances evaluate cosines (forward-only ones do not reach here with v_cos) -- keep all with MFMAs
                r = res.get(name, {})
                instances.append({"file": src, "kernel": name, "vgprs": r.get("vgprs"), "scratch_bytes_per_lane": r.get("scratch_bytes"),
                                  "waves_per_simd": r.get("occupancy"), "tile_loop_instructions": c["instructions"],
                                  "scratch_instructions_in_tile_loop": c["scratch_instructions_in_tile_loop"],
                                  "mfma_per_trip": c["by_class"].get("mfma", 0), "by_class": c["by_class"],
                                  # one trip = one 32-pixel tile of one wave, in all three families
                                  "vector_issue_cycles_per_tile": c["vector_issue_cycles_per_tile"],
                                  "mfma_pipe_cycles_per_tile": c["mfma_pipe_cycles_per_tile"],
                                  "transcendental_cycles_per_tile": c["transcendental_cycles_per_tile"],
                                  "lds_cycles_per_tile_and_wave": c["lds_cycles_per_tile_and_wave"]})
                if src == HEADLINE[0] and HEADLINE[1] in name:
                    headline = c
    assert headline is not None, "headline instance not found"
    out = {"kernel": "siren_wave_kernel<bf16, 3 hidden, F 16, E 16, C 3, loss + backward, dpe written> (rcb_siren_loss_bwd, BASELINE "
                     "configs[1]; one wave per row, two waves per SIMD)",
           "source_sha16": source_sha16(), "unit": "per 32-pixel tile and wave (one trip of the tile loop)",
           "issue_cycles_per_class": ISSUE, **headline,
           "floors_per_tile": {
               "vector_issue_one_wave_stream": headline["vector_issue_cycles_per_tile"],
               "matrix_pipe_per_simd": headline["mfma_pipe_cycles_per_tile"],
               "transcendental_unit_per_simd": headline["transcendental_cycles_per_tile"],
               "lds_per_cu_for_one_tile_on_each_simd": 4 * headline["lds_cycles_per_tile_and_wave"]},
           "instances": sorted(instances, key=lambda r: (r["file"], r["kernel"])),
           "note": "floors: a SIMD works on two tiles at a time (two waves); per tile and SIMD the matrix pipe needs "
                   "mfma_pipe_cycles, the transcendental unit its cycles, the CU's LDS 4 x lds_cycles for the four tiles its SIMDs "
                   "hold; the vector issue of ONE wave's stream is vector_issue_cycles (two waves interleave)"}
    if "--no-write" not in sys.argv:
        path = os.path.join(ROOT, "profiles", "r05_siren_isa_census.json")
        json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "instances"}, indent=1))
    bad = [r for r in instances if r["scratch_instructions_in_tile_loop"]]
    print("%d instances; scratch inside the tile loop in %d:" % (len(instances), len(bad)))
    for r in bad:
        print("  ", r["file"], r["kernel"][:90], r["scratch_instructions_in_tile_loop"], "per trip,", r["scratch_bytes_per_lane"], "B")
    # the instances the models' training step selects in the 16-bit modes: the wave family and the 16-bit-input-row (IN16)
    # instances of the two workgroup families (template argument Lb1 behind MODE_LOSS = Li2); the fp32-input-row instances
    # (Lb0) remain for callers that hold pe in fp32 or have no shared coordinate grid
    shipped_bad = [r for r in bad if r["file"] == "siren_mlp_wave.hip" or r["kernel"].endswith(("Li2ELb1EEEvN3rcb9SirenArgsE",))]
    if check and (headline["scratch_instructions_in_tile_loop"] or shipped_bad):
        print("FAIL: scratch inside the tile loop of an instance the training step selects:", [r["kernel"] for r in shipped_bad] or "headline")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
