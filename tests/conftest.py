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


class ThreadComm:
    """`world` "ranks" as threads of ONE process, with a mailbox and a barrier standing in for the collectives: the
    communicator interface of learning_drones_amd.parallel.DistComm.  A GPU box admits only a few processes on its card
    at a time, so the 8-rank shapes of BASELINE configs 4 and 5 are rehearsed this way (the torch.distributed code
    path itself is covered with 2-4 gloo processes)."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world, self.box, self.barrier = world, {}, threading.Barrier(world)

    device_alltoall = True

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, item):
        self.shared.box[self.rank] = item
        self.shared.barrier.wait()
        items = [self.shared.box[r] for r in range(self.world)]
        self.shared.barrier.wait()
        return items

    def all_gather(self, tensor):
        import torch
        return torch.cat(self._exchange(tensor.contiguous()), dim=0)

    def broadcast(self, tensor, src):
        items = self._exchange(tensor.clone() if self.rank == src else None)
        tensor.copy_(items[src])
        return tensor

    def all_to_all(self, send):
        items = self._exchange(list(send))
        return [items[r][self.rank] for r in range(self.world)]


def run_ranks(world, fn, timeout=1200):
    """fn(comm) on `world` threads, one ThreadComm each; returns the results in rank order, re-raises the first error."""
    import threading
    shared, out, errors = ThreadComm.Shared(world), [None] * world, []

    def body(r):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(0)
            out[r] = fn(ThreadComm(shared, r))
        except BaseException as e:                                        # surfaced below; release the others
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=timeout)
    if errors:
        raise errors[0]
    return out

