"""include/spoa_hx.hpp — the spoa.hpp-shaped C++ header over libhaslr_hip.so: a caller written against the reference's
five spoa symbols compiles and links against it (CPU), fails loudly without a device (no fallback), and on the GPU returns the oracle's
consensus for every set, through the per-edge objects and through the batch entry."""
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def caller(built, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("spoa") / "spoa_caller")
    lib = os.path.join(ROOT, "haslr_amd", "lib")
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "spoa_caller.cpp"), "-o", exe,
                           "-L", lib, "-lhaslr_hip", "-pthread", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def sets():
    rnd = random.Random(3)
    out = [["ACGTACGTTTGACCA"] * 3, ["ACGTACGTACGTTTGACCAGTACGGATC", "ACGTACGTACATTTGACCAGTACGGATC", "ACGTACGTACGTTTGACCAGTACGGATC"], ["A"], ["-", "ACGT"]]
    for L in (40, 333, 900):
        t = "".join(rnd.choice("ACGT") for _ in range(L))
        out.append(["".join(c for c in t if rnd.random() > 0.06) for _ in range(7)])
    return out


def text(ss):
    return "\n\n".join("\n".join(st) for st in ss) + "\n"


def test_caller_compiles_and_has_no_cpu_fallback(caller):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: covered by the gpu test")
    r = subprocess.run([caller], input=text(sets()), capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [[], ["--batch"]])
def test_caller_returns_the_oracle_consensus(caller, mode):
    import orclib
    ss = sets()
    r = subprocess.run([caller] + mode, input=text(ss), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = [orclib.poa_consensus([q for q in st if q != "-"]) for st in ss]
    assert r.stdout.split("\n")[:-1] == want


@pytest.mark.gpu
def test_concurrent_callers_are_combined_into_few_device_calls(caller):
    """the reference's own fan-out (asm_cal_cns_seq_MT: N pthreads, an engine + a graph each, Assemble.cpp:562-605) through the five symbols: 16 threads
    over 128 edges return the oracle's consensus for every edge, and their generate_consensus() calls are flat-combined - at most 128 / 8 device calls"""
    import orclib
    rnd = random.Random(9)
    ss = []
    for k in range(128):
        t = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(60, 700)))
        ss.append(["".join(c for c in t if rnd.random() > 0.07) for _ in range(rnd.randrange(3, 9))])
    r = subprocess.run([caller, "--threads", "16"], input=text(ss), capture_output=True, text=True, env=dict(os.environ, HASLR_SPOA_BATCH_US="3000"))
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:-1] == [orclib.poa_consensus(st) for st in ss]
    calls = int(r.stderr.split("device_calls=")[1].split()[0])
    assert int(r.stderr.split("sets=")[1].split()[0]) == 128 and calls <= 16, r.stderr


@pytest.mark.gpu
def test_a_bad_set_fails_its_own_caller_only_and_a_lone_caller_does_not_wait(caller):
    """flat combining must not spread a failure: 8 threads over 33 edges, one of which holds a sequence no POA kernel instance takes (2^20 bases:
    hx_poa_sequences fails the call it is in) - that edge's caller gets the exception, every other edge the oracle's consensus. And a lone caller
    (--threads 1, the reference with -t 1) does not sit through the batching window: with a window of 0.3 s, 12 edges take far less than 12 windows."""
    import time

    import orclib
    rnd = random.Random(10)
    ss = []
    for k in range(32):
        t = "".join(rnd.choice("ACGT") for _ in range(rnd.randrange(60, 400)))
        ss.append(["".join(c for c in t if rnd.random() > 0.07) for _ in range(rnd.randrange(3, 7))])
    bad = 13
    ss.insert(bad, ["A" * (1 << 20), "ACGT"])
    r = subprocess.run([caller, "--threads", "8", "--tolerant"], input=text(ss), capture_output=True, text=True, env=dict(os.environ, HASLR_SPOA_BATCH_US="20000"))
    assert r.returncode == 0, r.stderr
    got = r.stdout.split("\n")[:-1]
    assert len(got) == 33 and got[bad].startswith("ERROR") and "longer than" in got[bad], got[bad][:200]
    for k in range(33):
        if k != bad:
            assert got[k] == orclib.poa_consensus(ss[k]), k
    t0 = time.perf_counter()
    r = subprocess.run([caller, "--threads", "1"], input=text(ss[:12]), capture_output=True, text=True, env=dict(os.environ, HASLR_SPOA_BATCH_US="300000"))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr
    assert r.stdout.split("\n")[:-1] == [orclib.poa_consensus(st) for st in ss[:12]]
    assert "device_calls=12 sets=12" in r.stderr and dt < 12 * 0.3, (dt, r.stderr)   # (process start + context creation included: still below 12 windows)
