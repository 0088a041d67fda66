"""The pipeline driver (haslr_amd/driver/haslr_pipeline.py, installed as haslr_amd/bin/haslr.py) against the reference's
bin/haslr.py: same exit status, same stdout, same tool command lines in the same order, same output tree — in every scenario of
tests/driverlib.py, with the external tools replaced by recording stand-ins. The reference's side comes from tests/golden/driver/
(made by tests/golden/make_driver_golden.py) and, when /root/reference is present, also from the reference driver run live."""
import json
import os
import subprocess
import sys

import pytest

import driverlib

DRIVER = os.path.join(driverlib.ROOT, "haslr_amd", "driver", "haslr_pipeline.py")
REF = "/root/reference/bin/haslr.py"
GOLDEN = os.path.join(driverlib.ROOT, "tests", "golden", "driver")


@pytest.fixture(scope="module")
def noov(built):
    p = os.path.join(driverlib.ROOT, "haslr_amd", "bin", "minia_nooverlap")
    assert os.path.isfile(p), "haslr_amd/bin/minia_nooverlap is not built"
    return p


def ours(name, tmp_path, noov):
    data, bindir, out = str(tmp_path / "data"), str(tmp_path / "bin"), str(tmp_path / "out")
    driverlib.make_data(data)
    driverlib.make_bin(bindir, DRIVER, real={"minia_nooverlap": noov})
    return driverlib.run_scenario(name, bindir, data, out)


def same_runs(got, want, failed_step):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["rc"] == w["rc"]
        assert g["stdout"] == w["stdout"]
        assert g["calls"] == w["calls"]
        if not failed_step:
            assert g["tree"] == w["tree"]
        else:
            # the one deliberate difference: the failed step's captured product is not left behind under its final name
            gone = sorted(set(w["tree"]) - set(g["tree"]))
            extra = sorted(set(g["tree"]) - set(w["tree"]))
            assert [x + ".part" for x in gone] == extra or (not gone and not extra), (gone, extra)
            for k in set(g["tree"]) & set(w["tree"]):
                assert g["tree"][k] == w["tree"][k], k


@pytest.mark.parametrize("name", sorted(driverlib.SCENARIOS))
def test_same_as_the_recorded_reference_run(name, tmp_path, noov):
    want = json.load(open(os.path.join(GOLDEN, name + ".json")))
    same_runs(ours(name, tmp_path, noov), want, driverlib.SCENARIOS[name][1])


@pytest.mark.skipif(not os.path.isfile(REF), reason="/root/reference is not here")
@pytest.mark.parametrize("name", ["short_reads_pacbio", "contigs_given_nanopore_all_options", "minimap2_fails", "short_and_contig_missing"])
def test_same_as_the_reference_run_live(name, tmp_path, noov):
    data, bindir, out = str(tmp_path / "rdata"), str(tmp_path / "rbin"), str(tmp_path / "rout")
    driverlib.make_data(data)
    driverlib.make_bin(bindir, REF, real={"minia_nooverlap": noov})
    want = driverlib.run_scenario(name, bindir, data, out)
    same_runs(ours(name, tmp_path, noov), want, driverlib.SCENARIOS[name][1])


@pytest.mark.skipif(not os.path.isfile(REF), reason="/root/reference is not here")
def test_help_and_version_text(tmp_path, noov):
    outs = []
    for drv, sub in ((DRIVER, "a"), (REF, "b")):
        bindir = str(tmp_path / sub)
        driverlib.make_bin(bindir, drv, real={"minia_nooverlap": noov})
        for flag in ("-h", "-v"):
            pr = subprocess.run([sys.executable, os.path.join(bindir, "haslr.py"), flag], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            outs.append((pr.returncode, pr.stdout, pr.stderr))
    assert outs[0] == outs[2] and outs[1] == outs[3]


def test_a_missing_tool_stops_the_run(tmp_path, noov):
    data, bindir, out = str(tmp_path / "data"), str(tmp_path / "bin"), str(tmp_path / "out")
    driverlib.make_data(data)
    driverlib.make_bin(bindir, DRIVER, real={"minia_nooverlap": noov})
    os.remove(os.path.join(bindir, "minimap2"))
    env_path = os.environ.get("PATH", "")
    r = driverlib.run(bindir, data, out, ["-o", out, "-g", "1m", "-l", os.path.join(data, "lr1.fa"), "-x", "pacbio", "-c", os.path.join(data, "contigs.fa")],
                      extra_env={"PATH": "/nonexistent"})
    assert r["rc"] == os.EX_SOFTWARE and r["stdout"].endswith("checking $BIN/minimap2: not found\n") and not r["calls"]
    # an external tool may also come from PATH (the reference wants all five beside the driver)
    other = tmp_path / "elsewhere"
    driverlib.make_bin(str(other), DRIVER)
    r = driverlib.run(bindir, data, out, ["-o", out, "-g", "1m", "-l", os.path.join(data, "lr1.fa"), "-x", "pacbio", "-c", os.path.join(data, "contigs.fa")],
                      extra_env={"PATH": str(other) + os.pathsep + env_path})
    assert r["rc"] == 0 and [c["tool"] for c in r["calls"]] == ["fastutils", "fastutils", "minimap2", "haslr_assemble"]


def test_a_failed_step_is_redone_by_the_next_run(tmp_path, noov):
    data, bindir, out = str(tmp_path / "data"), str(tmp_path / "bin"), str(tmp_path / "out")
    driverlib.make_data(data)
    driverlib.make_bin(bindir, DRIVER, real={"minia_nooverlap": noov})
    args = ["-o", out, "-g", "1m", "-l", os.path.join(data, "lr1.fa"), "-x", "pacbio", "-c", os.path.join(data, "contigs.fa")]
    r1 = driverlib.run(bindir, data, out, args, fail="minimap2")
    assert r1["rc"] == os.EX_SOFTWARE
    r2 = driverlib.run(bindir, data, out, args)
    assert r2["rc"] == 0 and [c["tool"] for c in r2["calls"]] == ["minimap2", "haslr_assemble"]
    assert not [k for k in r2["tree"] if k.endswith(".part")]
    assert r2["tree"]["map_contigs_k49_a3_c250_lr25x.paf"].startswith("0\t320")


def test_extra_assembler_arguments(tmp_path, noov):
    data, bindir, out = str(tmp_path / "data"), str(tmp_path / "bin"), str(tmp_path / "out")
    driverlib.make_data(data)
    driverlib.make_bin(bindir, DRIVER, real={"minia_nooverlap": noov})
    args = ["-o", out, "-g", "1m", "-l", os.path.join(data, "lr1.fa"), "-x", "pacbio", "-c", os.path.join(data, "contigs.fa")]
    r = driverlib.run(bindir, data, out, args, extra_env={"HASLR_ASSEMBLE_ARGS": "--device 1"})
    assert r["rc"] == 0 and r["calls"][-1]["argv"][-2:] == ["--device", "1"]
