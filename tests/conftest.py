import os
import subprocess
import sys

import pytest

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
try:   # torch bundles its own HIP runtime: let it initialise first so that libhaslr_hip.so binds to the same one
    import torch
    torch.cuda.is_available()
except Exception:   # noqa: BLE001
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Everything compiled in-tree (no-op when already built; the GPU box receives the built files)."""
    need = [os.path.join(ROOT, "haslr_amd", "lib", "libhaslr_host.so"), os.path.join(ROOT, "haslr_amd", "lib", "libhaslr_hip.so"),
            os.path.join(ROOT, "haslr_amd", "bin", "haslr_assemble"), os.path.join(ROOT, "haslr_amd", "bin", "minia_nooverlap"),
            os.path.join(ROOT, "haslr_amd", "bin", "haslr.py"), os.path.join(ROOT, "oracle", "liboracle.so"),
            os.path.join(ROOT, "tools", "hxsim"), os.path.join(ROOT, "tools", "hxident")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    return ROOT


@pytest.fixture(scope="session")
def sim(built, tmp_path_factory):
    """sim(args) -> prefix of a seeded synthetic data set (cached per argument list for the session)."""
    cache = {}

    def make(*args):
        key = tuple(str(a) for a in args)
        if key not in cache:
            d = tmp_path_factory.mktemp("sim")
            pre = str(d / "in")
            subprocess.check_call([os.path.join(ROOT, "tools", "hxsim"), *key, "--out-prefix", pre], stderr=subprocess.DEVNULL)
            cache[key] = pre
        return cache[key]

    return make


@pytest.fixture(scope="session")
def ref_front(built):
    p = os.path.join(ROOT, "oracle", "_ref", "ref_front")
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/ref_front not built (needs /root/reference in the build container)")
    return p
