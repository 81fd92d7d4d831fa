"""Heuristic scan of the compiler's assembly for SERIALISED global loads -- a load whose s_waitcnt vmcnt(0) comes before the next
load is issued (a prologue that waits one L2 / HBM round trip per load instead of one for all):   python tools/serial_loads.py
Two patterns: (1) straight-line code: load, vmcnt(0), ..., load, vmcnt(0) with few instructions between; (2) a loop block of
< 40 instructions that holds a global load, its vmcnt(0) wait and a store.  Prints kernel, pattern and count; reading the source
at those places decides whether it matters (a loop with thousands of trips is a streaming loop, not a prologue)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "recombiner_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
with tempfile.TemporaryDirectory() as td:
    for f in files:
        asm = os.path.join(td, f + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, f), "-o", asm], check=True, stderr=subprocess.DEVNULL, cwd=td)
        lines = open(asm).read().split("\n")
        name, body = None, []
        kernels = {}
        for l in lines:
            m = re.match(r"^(_Z\w+):", l)
            if m:
                name, body = m.group(1), []
                kernels[name] = body
            elif name is not None:
                t = l.strip()
                if t and not t.startswith((";", ".")) or t.startswith(".LBB"):
                    body.append(t)
        for k, b in kernels.items():
            ins = [x for x in b]
            # pattern 1: consecutive (load ... vmcnt(0)) pairs within 12 instructions of each other
            idx = [i for i, x in enumerate(ins) if x.startswith(("global_load", "buffer_load")) and "lds" not in x]
            waits = [i for i, x in enumerate(ins) if x.startswith("s_waitcnt") and "vmcnt(0)" in x]
            chain = 0
            for a, c in zip(idx, idx[1:]):
                if any(a < w < c for w in waits) and c - a <= 12:
                    chain += 1
            # pattern 2: small loop blocks with load + vmcnt(0) + store
            loops, cur, cur_name = 0, [], None
            for x in ins + [".LBBend:"]:
                if x.startswith(".LBB"):
                    if cur_name and len(cur) < 40 and any(y.startswith("global_load") for y in cur) and any("vmcnt(0)" in y for y in cur) \
                            and any(y.startswith(("ds_write", "global_store")) for y in cur) and any(cur_name.rstrip(":") in y for y in cur if y.startswith("s_cbranch")):
                        loops += 1
                    cur, cur_name = [], x.split(":")[0] + ":"
                else:
                    cur.append(x)
            if chain >= 3 or loops:
                print("%-22s %-90s serial pairs %3d  load-wait-store loops %d" % (f, k[:90], chain, loops))
