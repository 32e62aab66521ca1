import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _oracle_comparable_bins():
    """The library default of the binning's angle function is "ocml" (= the reference build, bit for bit; a CPU cannot
    reproduce it).  The tests compare bins with the CPU oracle, so every test runs in "shared" mode (the correctly rounded
    atan2f kernels and oracle share) unless it selects "ocml" itself; the default is restored afterwards."""
    from sph3d_gcn_amd import tf_buildkernel
    tf_buildkernel.set_atan2("shared")
    yield
    tf_buildkernel.set_atan2(tf_buildkernel.DEFAULT_ATAN2)
