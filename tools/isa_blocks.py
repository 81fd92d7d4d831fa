"""Per-basic-block instruction census of one kernel in a hipcc assembly file (-S --offload-device-only).
    python tools/isa_blocks.py file.s <substring of the mangled kernel name> [--min N] [--dump LABEL]
Prints, per basic block with at least N instructions: matrix / transcendental / conversion / other vector / LDS / vector-memory /
scratch / accvgpr copy / wait / nop counts and the vector-issue cycles the block costs one wave (MI355X_MICROARCH.md: transcendental
8, v_cvt_pk 4.5, other VALU 4, an MFMA holds the issue for 8).  --dump prints one block's instructions."""
import re
import sys


def classify(op):
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_accvgpr_write"):
        return "accw"
    if op.startswith("v_accvgpr_read"):
        return "accr"
    if "mfma" in op:
        return "mfma"
    if op.startswith(("v_sin", "v_cos", "v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")):
        return "trans"
    if op.startswith("v_cvt_pk"):
        return "cvt"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    return "salu"


COST = {"trans": 8.0, "cvt": 4.5, "valu": 4.0, "mfma": 8.0, "accw": 4.0, "accr": 4.0}


def kernel_body(path, sub):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l and l.split(":")[0].strip().endswith(("E", "_")))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


def main():
    path, sub = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 20
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    body = kernel_body(path, sub)
    blk, order, cnt, text = "entry", ["entry"], {}, {}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            blk = m.group(1)
            order.append(blk)
            text.setdefault(blk, []).append(l)
            continue
        t = l.strip().split()
        text.setdefault(blk, []).append(l)
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        d = cnt.setdefault(blk, {})
        k = classify(t[0])
        d[k] = d.get(k, 0) + 1
    tot = {}
    for b in order:
        d = cnt.get(b, {})
        n = sum(d.values())
        for k, v in d.items():
            tot[k] = tot.get(k, 0) + v
        if n >= minn:
            cyc = sum(COST.get(k, 0) * v for k, v in d.items())
            hdr = [x for x in text.get(b, [""])[:1]]
            print(f"{b:12s} n={n:5d} issue_cyc={cyc:7.0f} ", " ".join(f"{k}={v}" for k, v in sorted(d.items())), "|", hdr[0][len(b) + 1:].strip()[:60] if hdr else "")
    print("total", " ".join(f"{k}={v}" for k, v in sorted(tot.items())))
    if dump:
        print("\n".join(text.get(dump, [])))


if __name__ == "__main__":
    main()
