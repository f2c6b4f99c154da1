import json
import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The product library is a build artefact (git-ignored).  If the tests run before build() has
    # been called, build it now (hipcc cross-compiles without a GPU); the package itself never does
    # this -- it fails loudly when the library is missing.
    so = ROOT / "basebandboard_amd" / "libbbb_hip.so"
    if not so.exists():
        import shutil
        import subprocess
        if shutil.which("hipcc") or pathlib.Path("/opt/rocm/bin/hipcc").exists():
            subprocess.check_call(["make", "-C", str(ROOT / "basebandboard_amd" / "csrc"), "-j4"])


@pytest.fixture(scope="session")
def golden_lutopt():
    return json.load(open(GOLDEN / "lutopt_clt.json"))


@pytest.fixture(scope="session")
def golden_prbs():
    return json.load(open(GOLDEN / "prbs.json"))


@pytest.fixture(scope="session")
def golden_gf2():
    return json.load(open(GOLDEN / "gf2.json"))


@pytest.fixture(scope="session")
def golden_shaper():
    return json.load(open(GOLDEN / "shaper.json"))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """The product package on cuda:0.  GPU tests must fail -- not skip -- when the HIP library is
    missing, so that a silent fallback can never pass."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import basebandboard_amd as bbb
    bbb._lib.lib()
    return bbb
