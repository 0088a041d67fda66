"""Recording stand-ins for the EXTERNAL programs the pipeline driver starts (Minia, minimap2, fastutils — absent from this image) and,
for the CPU tests, for haslr_assemble. Each call appends {"tool", "argv"} to $STUB_LOG and writes small, deterministic products so that
the next step has something to read. Test doubles for the orchestration only: nothing here computes what the real tools compute.
Started through tiny shell wrappers that tests/driverlib.py writes:  stub_tool.py <tool name> <arguments...>"""
import hashlib
import json
import os
import sys


def records(path):
    name, seq = None, []
    with open(path) as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if ln.startswith(">"):
                if name is not None:
                    yield name, "".join(seq)
                name, seq = ln[1:], []
            elif ln:
                seq.append(ln)
    if name is not None:
        yield name, "".join(seq)


def opt(argv, flag):
    return argv[argv.index(flag) + 1]


def main():
    tool, argv = sys.argv[1], sys.argv[2:]
    if argv == ["-h"]:
        return 0
    with open(os.environ["STUB_LOG"], "a") as f:
        f.write(json.dumps({"tool": tool, "argv": argv}) + "\n")
    fail = os.environ.get("STUB_FAIL", "")
    if tool == "fastutils":
        sub = argv[0]
        if fail == "fastutils:" + sub:
            sys.stdout.write(">partial\n")
            return 3
        if "--fofn" in argv:                                  # long reads: numeric names
            n = 0
            for fn in open(opt(argv, "-i")).read().split():
                for _, seq in records(fn):
                    sys.stdout.write(">%d\n%s\n" % (n, seq))
                    n += 1
        else:                                                 # contigs: length filter, comments kept
            m = int(opt(argv, "-m"))
            for name, seq in records(opt(argv, "-i")):
                if len(seq) >= m:
                    sys.stdout.write(">%s\n%s\n" % (name, seq))
    elif tool == "minia":
        print("minia stand-in: stdout")
        print("minia stand-in: stderr", file=sys.stderr)
        if fail == "minia":
            return 3
        prefix = opt(argv, "-out")
        src = open(os.environ["STUB_MINIA_CONTIGS"]).read()
        for kind in ("contigs", "unitigs"):
            open(prefix + "." + kind + ".fa", "w").write(src)
        open(prefix + ".unitigs.fa.glue.1", "w").write("x")
        open(prefix + ".h5", "w").write("x")
    elif tool == "minimap2":
        print("minimap2 stand-in: stderr", file=sys.stderr)
        if fail == "minimap2":
            sys.stdout.write("partial")
            return 3
        sys.stdout.write(open(os.environ["STUB_PAF"]).read())
    elif tool == "haslr_assemble":
        print("haslr_assemble stand-in: stdout")
        print("haslr_assemble stand-in: stderr", file=sys.stderr)
        if fail == "haslr_assemble":
            return 3
        d = opt(argv, "-d")
        os.makedirs(d, exist_ok=True)
        h = hashlib.sha1()
        for flag in ("-c", "-l", "-m"):
            h.update(open(opt(argv, flag), "rb").read())
        open(os.path.join(d, "asm.final.fa"), "w").write(">digest\n%s\n" % h.hexdigest())
    else:
        return 9
    return 0


if __name__ == "__main__":
    sys.exit(main())
