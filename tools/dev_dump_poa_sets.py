"""Writes the consensus operator's inputs of a data set in the format tools/make_spoa_vectors.cpp reads: one paragraph per backbone edge
(">edge<i>", then the gap sub-sequences of its supporting long reads in alignment order, "-" for an empty one; Assemble.cpp:503-543).
Test infrastructure (it drives the CPU oracle backend, like the tests do): a maintainer with a real SPOA 1.1.3 turns the output into
tests/golden/spoa/*.json.

    tools/hxsim --genome-len 300000 --seed 7 --out-prefix /tmp/v
    python tools/dev_dump_poa_sets.py /tmp/v [max_edges] > sequences.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def read_text(ds, rid, strand):
    n, off = int(ds.reads.len[rid]), int(ds.reads.off[rid])
    s = "".join("ACGT"[(ds.reads.packed[off + (i >> 2)] >> ((i & 3) * 2)) & 3] for i in range(n))
    return s if strand == 0 else s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def main():
    from haslr_amd import host
    import orclib
    pre = sys.argv[1]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 30
    ds = host.Dataset(pre + ".contigs.fa", pre + ".reads.fa", pre + ".paf")
    be = orclib.OracleBackend(ds, os.cpu_count() or 1)
    run = host.Run(ds, ds.params(), be.table, None)
    run.chain(); run.graph(); run.coords()
    c = run.coords_out()
    cache = {}
    for e in range(min(limit, len(c["supp_off"]) - 1)):
        print(f">edge{e}")
        for k in range(int(c["supp_off"][e]), int(c["supp_off"][e + 1])):
            rid, strand = int(c["supp_lr"][k]) & 0x7fffffff, int(c["supp_lr"][k]) >> 31
            if (rid, strand) not in cache:
                cache[(rid, strand)] = read_text(ds, rid, strand)
            t = cache[(rid, strand)]
            sp, ep = int(c["spos"][k]), int(c["epos"][k])
            want = (ep - sp + 1) & 0xffffffff                    # evaluated in 32 bits like the reference (Assemble.cpp:530-532)
            s = t[sp:sp + want]                                   # std::string::substr clamps to the end of the read
            print(s if s else "-")
        print()
    run.close(); be.close(); ds.close()


if __name__ == "__main__":
    main()
