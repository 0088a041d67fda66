"""Regenerates tests/golden/driver/*.json: what the REFERENCE's pipeline driver (bin/haslr.py, run in place from /root/reference through a
link, never copied) does in each scenario of tests/driverlib.py when all five tools are recording stand-ins (tests/stub_tool.py) except
minia_nooverlap, which is this build's tool (itself pinned to the compiled reference tool by tests/test_nooverlap.py). Stored per scenario
and per run: exit status, stdout with time stamps and temporary paths masked, the tools' command lines, the output tree.
    python tests/golden/make_driver_golden.py        (in the build container; needs /root/reference and a built haslr_amd/bin)"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import driverlib  # noqa: E402

REF = "/root/reference/bin/haslr.py"


def main():
    assert os.path.isfile(REF), "the reference driver is needed to regenerate these fixtures"
    noov = os.path.join(driverlib.ROOT, "haslr_amd", "bin", "minia_nooverlap")
    os.makedirs(os.path.join(HERE, "driver"), exist_ok=True)
    for name in driverlib.SCENARIOS:
        with tempfile.TemporaryDirectory() as tmp:
            data, bindir, out = os.path.join(tmp, "data"), os.path.join(tmp, "bin"), os.path.join(tmp, "out")
            driverlib.make_data(data)
            driverlib.make_bin(bindir, REF, real={"minia_nooverlap": noov})
            runs = driverlib.run_scenario(name, bindir, data, out)
        with open(os.path.join(HERE, "driver", name + ".json"), "w") as f:
            json.dump(runs, f, indent=1, sort_keys=True)
        print(name, [r["rc"] for r in runs], len(runs[-1]["calls"]), "calls", len(runs[-1]["tree"]), "files")


if __name__ == "__main__":
    main()
