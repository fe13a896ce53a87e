"""Replay a saved best drone: counterpart of the reference's fly_best_drone.py:8-25 (which loads the
jsonpickle checkpoint and re-flies it with the TCP viewers attached).  Here the checkpoint is the JSON
written by evolve.evolution.save_drone, the episode runs on the GPU, and the trajectory is written as
a Results.csv in the reference's 25-column layout, which its plotting tools read.

    python -m learning_drones_amd.fly_best_drone <execution_time | path.json> [timesteps] [out_dir]
"""
from __future__ import annotations

import os
import sys

import numpy as np


def main(argv) -> int:
    if len(argv) < 2:
        print(__doc__)
        return 2
    ref = argv[1]
    path = ref if ref.endswith(".json") else f"evolution/best_drone_{ref}.json"
    timesteps = int(argv[2]) if len(argv) > 2 else 2500
    out_dir = argv[3] if len(argv) > 3 else "rewatch"
    from .evolve.evolution import load_drone
    from .flight import fly
    from .population import Population
    drone = load_drone(path)
    from .params import controller_spec
    pop = Population(1, controller_spec(drone.controller).hidden_widths, precision=drone.precision)
    pop.set_genomes(drone.genes[None, :])
    os.makedirs(out_dir, exist_ok=True)
    pop.record_trajectory(timesteps, slots=(0,), csv_prefix=os.path.join(out_dir, "Results_"))
    os.replace(os.path.join(out_dir, "Results_0.csv"), os.path.join(out_dir, "Results.csv"))
    reward, real_time, _ = fly(drone, timesteps=timesteps)
    print(f"Final Score: {reward}, coins {drone.coins}, crashed {drone.crashed}, time in air {drone.flight_time}, "
          f"flight time {real_time:.2f} s, distance to coin {drone.distance_to_coin:.3f}; trajectory in {out_dir}/Results.csv")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
