"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares; ctypes mirrors have the
C sizes; the CLI keeps the reference's exit-code conventions and fails loudly without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from haslr_amd import ctypes_defs as T
from haslr_amd import hip, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hxh?_[a-z_0-9]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol(built):
    L = hip.lib()
    names = declared_functions(os.path.join(ROOT, "include", "haslr_hip.h"))
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), f"libhaslr_hip.so does not export {n}"
    assert sorted(hip.SYMBOLS) == names


def test_host_library_exports_every_declared_symbol(built):
    L = host.lib()
    for n in declared_functions(os.path.join(ROOT, "haslr_amd", "csrc", "host", "haslr_host.h")):
        assert hasattr(L, n), f"libhaslr_host.so does not export {n}"


def test_struct_sizes_match_c(built, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "haslr_types.h"\n#include "haslr_host.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(hx_params),sizeof(hx_contigs),sizeof(hx_reads),sizeof(hx_hits),sizeof(hx_chain_out),sizeof(hx_rec_side),"
                   "sizeof(hx_edges_out),sizeof(hx_coords_out),sizeof(hx_cns_out),sizeof(hx_poa_params),sizeof(hx_backend));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "haslr_amd", "csrc", "host"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    want = [C.sizeof(t) for t in (T.Params, T.Contigs, T.Reads, T.Hits, T.ChainOut, T.RecSide, T.EdgesOut, T.CoordsOut, T.CnsOut, T.PoaParams, T.Backend)]
    assert got == want


def test_cli_help_and_version_exit_zero(built):
    exe = os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble")
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--aln-block" in r.stderr and "--edge-sup" in r.stderr
    r = subprocess.run([exe, "--version"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "0.8a1"     # haslr.py check_program needs exit 0 (bin/haslr.py:271-272)
    assert subprocess.run([exe], capture_output=True).returncode != 0
    r = subprocess.run([exe, "-c", "x.fa", "-l", "y.fa"], capture_output=True, text=True)
    assert r.returncode != 0 and "option -m is required" in r.stderr


def test_no_silent_cpu_fallback(built, sim, tmp_path):
    """Without a HIP device the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.HipError):
        hip.HipContext(0)
    pre = sim("--genome-len", "40000", "--seed", "3", "--cov", "6")
    exe = os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble")
    r = subprocess.run([exe, "-c", pre + ".contigs.fa", "-l", pre + ".reads.fa", "-m", pre + ".paf", "-d", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode != 0 and "[ERROR]" in r.stderr
    assert not os.path.exists(tmp_path / "o" / "asm.final.fa")


def test_loader_rejects_bad_inputs(built, tmp_path):
    c = tmp_path / "c.fa"
    c.write_text(">0 LN:i:8 KC:i:10 km:f:30.0\nACGTACGT\n")
    r = tmp_path / "r.fa"
    r.write_text(">0\nACGTACGTAC\n>1\nACGT\n")
    p = tmp_path / "m.paf"
    p.write_text("1\t4\t0\t4\t+\t0\t8\t0\t4\t4\t4\t60\tcg:Z:4M\n0\t10\t0\t8\t+\t0\t8\t0\t8\t8\t8\t60\tcg:Z:8M\n")
    with pytest.raises(host.HostError, match="ascending query"):
        host.Dataset(str(c), str(r), str(p))
    p.write_text("0\t10\t0\t8\t+\t7\t8\t0\t8\t8\t8\t60\tcg:Z:8M\n")
    with pytest.raises(host.HostError, match="not a loaded contig"):
        host.Dataset(str(c), str(r), str(p))
    c.write_text(">0\nACGT\n")
    with pytest.raises(host.HostError, match="KC:i:"):
        host.Dataset(str(c), str(r), str(p))
    with pytest.raises(host.HostError, match="could not open"):
        host.Dataset(str(tmp_path / "nope.fa"), str(r), str(p))


def test_loader_fastq_gz_and_multiline(built, tmp_path):
    import gzip
    c = tmp_path / "c.fa"
    c.write_text(">0 LN:i:8 KC:i:10 km:f:30.5\nACGT\nACGT\n>1 x KC:i:7 km:f:61\nTTTTGGGG\n")
    with gzip.open(tmp_path / "r.fq.gz", "wt") as f:
        f.write("@0 desc\nACGTNACGTA\n+\n@>!!!!!!!!\n@1\nAC\nGT\n+\n!!\n!!\n")
    p = tmp_path / "m.paf"
    p.write_text("")
    ds = host.Dataset(str(c), str(tmp_path / "r.fq.gz"), str(p))
    assert ds.contigs.n == 2 and ds.reads.n == 2 and ds.hits.n == 0
    assert [ds.reads.len[i] for i in range(2)] == [10, 4] and ds.total_read_bases == 14
    assert abs(ds.uniq_freq - (30.5 + 61) / 2) < 1e-12
    # N packs as A (Compressed_sequence.cpp: table value & 3)
    b = ds.reads.packed[1]
    assert (b & 3) == 0
