"""helpers shared by tests/test_driver.py and tests/golden/make_driver_golden.py: run a pipeline driver over recording stand-in tools
and return everything observable about the run with the temporary paths replaced by $OUT / $BIN / $DATA"""
import json
import os
import re
import shutil
import stat
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "stub_tool.py")
STAMP = re.compile(r"\[\d\d-\w{3}-\d{4} \d\d:\d\d:\d\d\]")

CONTIGS = """>0 LN:i:400 KC:i:8000 km:f:20.0 L:+:1:+
%s
>1 LN:i:320 KC:i:6400 km:f:20.0 L:-:0:- L:+:2:-
%s
>2 LN:i:120 KC:i:2400 km:f:20.0 L:+:1:-
%s
""" % ("ACGT" * 100, "TTGCA" * 64, "GATTACA" * 17 + "C")


def make_data(data):
    """small input files (contents do not matter to the orchestration, names and flow do)"""
    os.makedirs(data, exist_ok=True)
    files = {"contigs.fa": CONTIGS,
             "lr1.fa": ">readA\n" + "ACGTTGCA" * 40 + "\n>readB\n" + "GGATCC" * 50 + "\n",
             "lr2.fa": ">readC\n" + "TTAGGC" * 45 + "\n",
             "sr1.fq": "@s1\nACGT\n+\nIIII\n", "sr2.fq": "@s2\nTTGA\n+\nIIII\n",
             "map.paf": "0\t320\t0\t300\t+\t0\t400\t10\t310\t290\t300\t60\tcg:Z:300M\n"}
    for k, v in files.items():
        open(os.path.join(data, k), "w").write(v)
    open(os.path.join(data, "long.fofn"), "w").write(os.path.join(data, "lr1.fa") + "\n" + os.path.join(data, "lr2.fa") + "\n")
    open(os.path.join(data, "short.fofn"), "w").write(os.path.join(data, "sr1.fq") + "\n" + os.path.join(data, "sr2.fq") + "\n")


def make_bin(bindir, driver, real=()):
    """a bin/ like the reference's: haslr.py + the five tools. `driver` is copied unless it is a path under /root/reference (then it is
    linked: the reference may be run in place, never copied). Tools in `real` = {name: path} are linked, the others are stand-ins."""
    os.makedirs(bindir, exist_ok=True)
    dst = os.path.join(bindir, "haslr.py")
    if driver.startswith("/root/reference/"):
        os.symlink(driver, dst)
    else:
        shutil.copy(driver, dst)
    for tool in ("haslr_assemble", "minia_nooverlap", "fastutils", "minia", "minimap2"):
        p = os.path.join(bindir, tool)
        if tool in real:
            os.symlink(real[tool], p)
        else:
            open(p, "w").write('#!/bin/sh\nexec %s %s %s "$@"\n' % (sys.executable, STUB, tool))
            os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)


def run(bindir, data, out, args, fail="", extra_env=None):
    """-> {"rc", "stdout", "calls", "tree"} with paths normalised"""
    log = os.path.join(os.path.dirname(bindir), os.path.basename(bindir) + ".calls.jsonl")
    open(log, "w").close()
    env = dict(os.environ, STUB_LOG=log, STUB_FAIL=fail, STUB_MINIA_CONTIGS=os.path.join(data, "contigs.fa"), STUB_PAF=os.path.join(data, "map.paf"))
    env.update(extra_env or {})
    pr = subprocess.run([sys.executable, os.path.join(bindir, "haslr.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=data)

    def norm(s):
        for path, tag in ((out, "$OUT"), (bindir, "$BIN"), (data, "$DATA")):
            s = s.replace(path, tag)
        return STAMP.sub("[T]", s)

    calls = [json.loads(ln) for ln in open(log)]
    for c in calls:
        c["argv"] = [norm(x) for x in c["argv"]]
    tree = {}
    for d, _, fs in os.walk(out):
        for f in fs:
            p = os.path.join(d, f)
            tree[os.path.relpath(p, out)] = norm(open(p, errors="replace").read())
    return {"rc": pr.returncode, "stdout": norm(pr.stdout.decode()), "calls": calls, "tree": tree}


# name -> (arguments with $OUT / $DATA, stand-in failure, run twice in the same directory?)
SCENARIOS = {
    "short_reads_pacbio": (["-o", "$OUT", "-g", "4600k", "-l", "$DATA/lr1.fa", "$DATA/lr2.fa", "-x", "pacbio", "-s", "$DATA/sr1.fq", "$DATA/sr2.fq", "-t", "4"], "", False),
    "resume_everything_exists": (["-o", "$OUT", "-g", "4600k", "-l", "$DATA/lr1.fa", "$DATA/lr2.fa", "-x", "pacbio", "-s", "$DATA/sr1.fq", "$DATA/sr2.fq", "-t", "4"], "", True),
    "contigs_given_nanopore_all_options": (["-o", "$OUT", "-g", "5m", "-l", "$DATA/long.fofn", "--long-fofn", "-x", "nanopore", "-c", "$DATA/contigs.fa", "--cov-lr", "0",
                                            "--aln-block", "700", "--aln-sim", "0.9", "--edge-sup", "2", "--minia-kmer", "31", "--minia-solid", "2", "--minia-asm", "unitigs",
                                            "--min-src", "300"], "", False),
    "corrected_short_fofn_unitigs": (["-o", "$OUT", "-g", "1g", "-l", "$DATA/lr1.fa", "-x", "corrected", "-s", "$DATA/short.fofn", "--short-fofn", "--minia-asm", "unitigs",
                                      "--threads", "0"], "", False),
    "relative_paths": (["-o", "rel_out", "-g", "1m", "-l", "lr1.fa", "-x", "pacbio", "-c", "contigs.fa"], "", False),
    "minia_fails": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "-x", "pacbio", "-s", "$DATA/sr1.fq"], "minia", False),
    "fastutils_subsample_fails": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "-x", "pacbio", "-s", "$DATA/sr1.fq"], "fastutils:subsample", False),
    "minimap2_fails": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "-x", "pacbio", "-c", "$DATA/contigs.fa"], "minimap2", False),
    "haslr_assemble_fails": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "-x", "pacbio", "-c", "$DATA/contigs.fa"], "haslr_assemble", False),
    "no_arguments": ([], "", False),
    "long_reads_missing": (["-o", "$OUT", "-g", "1m", "-x", "pacbio", "-c", "$DATA/contigs.fa"], "", False),
    "short_and_contig_missing": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "-x", "pacbio"], "", False),
    "input_file_missing": (["-o", "$OUT", "-g", "1m", "-l", "$DATA/lr1.fa", "$DATA/nope.fa", "-x", "pacbio", "-c", "$DATA/contigs.fa"], "", False),
}


def run_scenario(name, bindir, data, out):
    args, fail, twice = SCENARIOS[name]
    if name == "relative_paths":
        out = os.path.join(data, "rel_out")
        shutil.rmtree(out, ignore_errors=True)
    args = [a.replace("$OUT", out).replace("$DATA", data) for a in args]
    first = run(bindir, data, out, args, fail)
    if twice:
        return [first, run(bindir, data, out, args, fail)]
    return [first]
