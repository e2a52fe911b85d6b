"""Instruction mix of the loops of one kernel in a hipcc -S listing (no GPU needed).

  hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S ilqg_api.hip -o d14.s
  python scripts/isa_loops.py d14.s ilq_lq_kernelIdLi14ELi3ELi2ELi1E

Prints, for every backward branch (a loop), the number of instructions between the label and the branch by class
(fp64 / other VALU, scalar, LDS, VMEM, MFMA, waits, s_and_saveexec regions)."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu64" if "_f64" in op else "valu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels = {}
    instrs = []  # (index in instrs, op, text)
    for i in range(start, end + 1):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not l or l.startswith((";", ".", "//")) or l.endswith(":"):
            continue
        op = l.split()[0]
        instrs.append((op, l))
    print("kernel instructions:", len(instrs))
    for idx, (op, l) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = l.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                body = instrs[labels[tgt]:idx + 1]
                if len(body) < 40:
                    continue
                mix = {}
                for o, t in body:
                    c = classify(o)
                    mix[c] = mix.get(c, 0) + 1
                saveexec = sum(1 for o, t in body if "saveexec" in o)
                readlane = sum(1 for o, t in body if "readlane" in o or "readfirstlane" in o)
                print("loop %s: %d instrs %s saveexec=%d readlane=%d" % (tgt, len(body), dict(sorted(mix.items())), saveexec, readlane))


if __name__ == "__main__":
    main()
