"""inputs for minia_nooverlap (tests/test_nooverlap.py, tests/golden/make_nooverlap_golden.py): Minia-style assemblies and odd text"""
import random


def minia_like(seed, n=40, k=31, min_len=None, crlf=False, wrap=0, blank_lines=False, tabs=False):
    rng = random.Random(seed)
    min_len = k if min_len is None else min_len
    nl = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        ln = rng.randrange(min_len, min_len + 300)
        seq = "".join(rng.choice("ACGT") for _ in range(ln))
        links = []
        for _ in range(rng.choice([0, 0, 1, 2, 3])):
            links.append("L:%s:%d:%s" % (rng.choice("+-"), rng.randrange(n), rng.choice("+-")))
        sep = "\t" if tabs and i % 3 == 0 else " "
        out.append(">%d%sLN:i:%d KC:i:%d km:f:%.3f%s" % (i, sep, ln, ln * 20, rng.uniform(5, 60), "".join(" " + x for x in links)))
        if wrap:
            out.extend(seq[j:j + wrap] for j in range(0, ln, wrap))
        else:
            out.append(seq)
        if blank_lines and i % 4 == 0:
            out.append("")
    return nl.join(out) + nl


# name -> (text, k as given on the command line)
def cases():
    c = {}
    c["plain_k31"] = (minia_like(1), "31")
    c["plain_k49"] = (minia_like(2, k=49), "49")
    c["even_k"] = (minia_like(3, k=32), "32")
    c["k1_and_k2"] = (minia_like(4, k=2), "2")
    c["k_one"] = (minia_like(4, k=2), "1")
    c["k_zero"] = (minia_like(5), "0")
    c["k_not_a_number"] = (minia_like(5), "abc")
    c["k_with_trailing_text"] = (minia_like(6), "31x")
    c["crlf"] = (minia_like(7, crlf=True), "31")
    c["wrapped_blank_lines_tabs"] = (minia_like(8, wrap=60, blank_lines=True, tabs=True), "31")
    c["crlf_wrapped"] = (minia_like(9, crlf=True, wrap=70, blank_lines=True), "31")
    c["no_final_newline"] = (minia_like(10).rstrip("\n"), "31")
    c["text_before_first_record"] = ("junk line\nmore junk\n" + minia_like(11), "31")
    # a header without a comment after one with a comment: the previous comment is printed and decides the trim
    c["stale_comment"] = (">a LN:i:60 KC:i:1 km:f:1.0 L:-:5:+\n" + "ACGT" * 15 + "\n>b\n" + "TTGCA" * 12 + "\n>c LN:i:40 KC:i:1 km:f:2.0\n" + "G" * 40 + "\n>d\n" + "C" * 40 + "\n", "21")
    # only a successor and shorter than half the overlap: kept whole
    c["short_with_successor"] = (">a LN:i:5 KC:i:1 km:f:1.0 L:+:1:+\nACGTA\n>b LN:i:30 KC:i:1 km:f:1.0 L:+:0:-\n" + "ACG" * 10 + "\n", "31")
    # a trim that takes everything
    c["trimmed_to_nothing"] = (">a LN:i:30 KC:i:1 km:f:1.0 L:-:1:+ L:+:1:+\n" + "ACG" * 10 + "\n>b LN:i:15 KC:i:1 km:f:1.0 L:-:1:+\n" + "ACG" * 5 + "\n", "31")
    # FASTQ records, one of them with '@' and '>' opening its quality line
    c["fastq"] = ("@r1 LN:i:40 KC:i:1 km:f:1.0 L:+:2:+\n" + "ACGT" * 10 + "\n+\n" + "@" + "I" * 39 + "\n@r2 LN:i:30 KC:i:1 km:f:1.0 L:-:1:+\n" + "TTG" * 10 + "\n+r2\n" + ">" + "#" * 29 + "\n", "11")
    c["fastq_truncated"] = ("@r1 LN:i:40 KC:i:1 km:f:1.0\n" + "ACGT" * 10 + "\n+\n" + "I" * 40 + "\n@r2 LN:i:30 KC:i:1 km:f:1.0\n" + "TTG" * 10 + "\n+\nIII\n", "11")
    # short fields, fields that are not links, blanks inside a sequence line, lower case
    c["odd_fields"] = (">a LN:i:50 KC:i:1 km:f:1.0 x L: L:+ ab-cd zz+\n" + "acgtn" * 10 + "\n>b  LN:i:50   KC:i:1 km:f:1.0   L:-:0:+  \n" + "AC GT\tAC" * 8 + "\n", "9")
    # where the reference dies on an uncaught exception: only the exit status is compared
    c["predecessor_trim_too_long"] = (">a LN:i:30 KC:i:1 km:f:1.0\n" + "ACG" * 10 + "\n>b LN:i:9 KC:i:1 km:f:1.0 L:-:0:+\nACGTACGTA\n", "31")
    c["first_record_without_comment"] = (">a\n" + "ACG" * 10 + "\n", "31")
    c["negative_k_with_predecessor"] = (">a LN:i:30 KC:i:1 km:f:1.0 L:-:0:+\n" + "ACG" * 10 + "\n", "-7")
    c["negative_k_with_successor"] = (">a LN:i:30 KC:i:1 km:f:1.0 L:+:0:+\n" + "ACG" * 10 + "\n", "-7")
    c["empty_file"] = ("", "31")
    c["header_only"] = (">a LN:i:0 KC:i:0 km:f:0.0\n", "31")
    return c
