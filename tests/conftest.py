import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # the shared objects are git-ignored build artefacts: build whatever is missing (hipcc cross-compiles
    # gfx950 without a GPU; on the GPU box the prebuilt files travel with the snapshot)
    lib = os.path.join(ROOT, "mgpreconditionedgcr_amd", "libmgcr_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(ROOT, "mgpreconditionedgcr_amd", "csrc")],
                       check=True, capture_output=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sample_gold():
    return dict(np.load(os.path.join(GOLDEN, "sample_4x4.npz")))


@pytest.fixture(scope="session")
def poisson_gold():
    return dict(np.load(os.path.join(GOLDEN, "poisson.npz")))


@pytest.fixture(scope="session")
def hsparse_gold():
    return dict(np.load(os.path.join(GOLDEN, "hsparse.npz")))


@pytest.fixture(scope="session")
def mg_gold():
    return dict(np.load(os.path.join(GOLDEN, "mg_4x4.npz")))


@pytest.fixture(scope="session")
def sample_matrix_path(tmp_path_factory):
    """The reference's data/sample_matrix/4x4parsed.txt, unpacked from the committed fixture."""
    d = tmp_path_factory.mktemp("sample_matrix")
    dst = os.path.join(str(d), "4x4parsed.txt")
    with gzip.open(os.path.join(GOLDEN, "4x4parsed.txt.gz"), "rb") as fi, open(dst, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    return dst
