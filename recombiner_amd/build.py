"""Builds librcb_hip.so (gfx950) in-tree with hipcc.  Usage: python -m recombiner_amd.build [--force]

Every .hip file is compiled to its own object (in parallel, cached under recombiner_amd/lib/obj by the modification
times of the source and the shared headers), then linked: a change to one kernel file costs one compile, not eleven."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "librcb_hip.so")
SOURCES = ["atrans.hip", "phase_weight.hip", "phaseconv.hip", "posterior.hip", "rec_score.hip", "siren_mlp.hip",
           "siren_mlp_bf16.hip", "siren_mlp_wave.hip", "siren_mlp_wide.hip", "siren_mlp_generic.hip", "stage1.hip", "tiles.hip", "upconv.hip",
           "upconv_weff.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(HERE, "..", "include", "rcb.h")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    deps = [os.path.join(CSRC, s) for s in SOURCES] + _headers()
    return _stale(LIB, deps)


def build(force=False, verbose=True, jobs=None, variant=None, extra_flags=(), only=()):
    """variant: a diagnostic build of the same ABI -- the sources named in `only` recompiled with `extra_flags` (-D
    switches), the other objects shared with the main build -- written to lib/librcb_<variant>.so and selected at run
    time with RCB_LIB (same-box A/B timing of kernel changes)."""
    if not variant and not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    cc, hdrs = hipcc(), _headers()
    todo, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            todo.append([cc] + FLAGS + ["-c", src, "-o", obj])
        if variant and s in only:
            obj = os.path.join(OBJDIR, f"{s[:-4]}.{variant}.o")
            todo.append([cc] + FLAGS + list(extra_flags) + ["-c", src, "-o", obj])
        objs.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs = jobs or min(len(todo), os.cpu_count() or 4, 8) or 1
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(run, todo))
    lib = os.path.join(LIBDIR, f"librcb_{variant}.so") if variant else LIB
    run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib])
    if variant and needs_build():        # (the main library is kept current as well)
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [os.path.join(OBJDIR, s[:-4] + ".o") for s in SOURCES] + ["-o", LIB])
    return lib


if __name__ == "__main__":
    # python -m recombiner_amd.build [--force] [--variant NAME --only a.hip,b.hip -DX=1 ...]
    argv = sys.argv[1:]
    var = argv[argv.index("--variant") + 1] if "--variant" in argv else None
    only = argv[argv.index("--only") + 1].split(",") if "--only" in argv else ()
    print(build(force="--force" in argv, variant=var, extra_flags=[f for f in argv if f.startswith("-D")], only=only))
