import os
import sys
from pathlib import Path

import numpy as np
import pytest

# The tests choose their kernels explicitly (specialise=True / the generic_kernels fixture): the automatic run-time
# specialisation of large float32 populations would put a hipcc run behind every odd layer spec the suite uses.
os.environ.setdefault("LD_SPECIALISE", "0")

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(GOLDEN / name, allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand."""
    from oracle import oracle as O
    O.build()
    O.set_dot_variant(0)
    return O


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_step.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_mlp.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_episodes.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_pid.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_events.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_ga.npz")


@pytest.fixture(scope="session")
def g7():
    return load_golden("g7_flags.npz")


@pytest.fixture(scope="session")
def g8():
    return load_golden("g8_frames.npz")


@pytest.fixture
def lib():
    """The loaded libld_drones.so (ctypes); the ld_debug_* switches are back at their defaults when the test ends."""
    from learning_drones_amd import _native
    handle = _native.lib()
    yield handle
    handle.ld_debug_force_generic(0)
    handle.ld_debug_lds_keep(-1)


@pytest.fixture
def generic_kernels():
    """generic_kernels(True) routes every layer spec to the run-time-spec kernels (ld_debug_force_generic);
    the unrolled specialisations are back when the test ends."""
    from learning_drones_amd import _native
    lib = _native.lib()
    yield lambda on: lib.ld_debug_force_generic(1 if on else 0)
    lib.ld_debug_force_generic(0)
