"""learning-drones_amd — MI355X-native batched quadcopter simulator and neuroevolution fitness
evaluator behind the reference's Drone / fly() / evolve API (mbecker12/learning-drones).

Layout
  csrc/            HIP kernels + the C ABI of include/ld_drones.h  (libld_drones.so)
  build.py         hipcc build of that library (parallel translation units)
  _native.py       ctypes binding of the ABI (fails loudly without the library or a GPU)
  params.py        POD mirrors + the reference's constants
  population.py    Population: SoA device state, genome packing, fly(), breed(), checkpoints
  parallel.py      one-process-per-GPU sharding, fitness all-gather, DeviceEvolution (GA loop on device)
  pid.py           batched PID policy (BASELINE config 1 on the device)
  controllers.py   hand-built hover controllers in the reference's weights format
  drone/ evolve/ flight/ objects/   drop-in mirrors of the reference's modules
"""
from .params import (EpisodeParams, FitnessRecord, MlpSpec, SimParams, coin_chain_reference,  # noqa: F401
                     coin_table, flatten_weights, spec_of_weights, unflatten_weights)

__version__ = "0.1.0"


def __getattr__(name):
    # heavy / GPU-touching members are resolved lazily so that host-only use needs no torch
    if name in ("Population",):
        from . import population
        return getattr(population, name)
    if name in ("Drone",):
        from .drone.drone import Drone
        return Drone
    if name == "specialise":
        from ._native import specialise
        return specialise
    if name in ("fly", "fly_population"):
        from . import flight
        return getattr(flight, name)
    raise AttributeError(name)
