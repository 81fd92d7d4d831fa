"""One-off: tune the library GEMMs of a workload on this GPU and write a TunableOp result file.
   python tools/tune_gemms.py <out.csv> [bench|compress|presets [label ...]]   then   python tools/tune_gemms.py --merge out.csv a.csv b.csv ..."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def merge(out, files):
    head, rows = None, {}
    for f in files:
        lines = [l.rstrip("\n") for l in open(f) if l.strip()]
        v = [l for l in lines if l.startswith("Validator,")]
        if head is None:
            head = v
        elif v != head:
            raise SystemExit("validator lines of %s differ from %s" % (f, files[0]))
        for l in lines:
            if not l.startswith("Validator,"):
                op, key = l.split(",")[:2]
                rows[(op, key)] = l          # later files win
    with open(out, "w") as fh:
        fh.write("\n".join(head + list(rows.values())) + "\n")
    print("merged %d entries into %s" % (len(rows), out))


if __name__ == "__main__":
    if sys.argv[1] == "--merge":
        merge(sys.argv[2], sys.argv[3:])
        sys.exit(0)
    from recombiner_amd import tuning
    out = os.path.abspath(sys.argv[1])
    what = sys.argv[2] if len(sys.argv) > 2 else "bench"
    tuning.enable_tuned_gemms(out, tune=True)
    here = os.path.dirname(os.path.abspath(__file__))
    if what == "bench":
        sys.argv = ["bench.py", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"]
        sys.path.insert(0, os.path.dirname(here))
        import bench
        bench.main()
    elif what == "presets":            # the patched presets' small-row GEMMs (tools/bench_presets.py: kodak, audio, video, ...)
        sys.argv = ["bench_presets.py"] + sys.argv[3:]
        runpy.run_path(os.path.join(here, "bench_presets.py"), run_name="__main__")
    else:
        sys.argv = ["bench_compress.py", "bf16"]
        runpy.run_path(os.path.join(here, "bench_compress.py"), run_name="__main__")
    print("TunableOp writes", out, "at interpreter exit")
