import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as O

    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu():
    """torch + the product package on a real GPU; fails loudly when the HIP library cannot run."""
    import torch

    assert torch.cuda.is_available(), "the -m gpu tests need a GPU"
    import phastft_amd as P

    info = P.device_info()
    assert "gfx950" in info["name"], info
    return P


@pytest.fixture
def static_rules():
    """For tests that assert on the plans of plan.hpp's static rules (pass counts, kernel names in describe()): planners made
    inside the test do not see the library's built-in wisdom (csrc/builtin_wisdom.inc), which replaces those plans wherever the
    tuner measured a faster one."""
    import phastft_amd as P

    was = P.wisdom_builtin(False)
    yield
    P.wisdom_builtin(was)   # what it WAS: a suite run under PHAST_BUILTIN_WISDOM=0 stays on the static rules
