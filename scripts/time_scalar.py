"""What the scalar drop-in surface costs per timestep: Drone.measure() + Drone.translate_input_to_thrust() driven by hand
(the loop body of the reference's fly(), flight/__init__.py:39-89), against the reference's own 105-120 us per timestep on
one CPU core (108 us measured in the build container with scripts/time_scalar.py --reference).

    python scripts/time_scalar.py [steps]            # this package, on the GPU
    PYTHONPATH=/root/reference python scripts/time_scalar.py --reference [steps]   # the reference itself, CPU (build container only)
"""
import os
import sys
import time

import numpy as np

steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 2000
if "--reference" in sys.argv:
    os.chdir("/tmp")
    np.Infinity = np.inf
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        from drone.drone import Drone
        from objects.coins import Coin
        d = Drone(visualize=False, n_servers=0)
else:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.objects.coins import Coin
    d = Drone(visualize=False, n_servers=0)
np.random.seed(5)
coin = Coin(np.array([[5.0], [5.0], [10.0]]), 2, 2000)
wind = np.zeros((3, 1))


def body():
    d.reward += 2
    d.flight_time += 1
    coin.is_in_vicinity(d.position)
    d.measure(wind)
    d.translate_input_to_thrust(coin.position)


for _ in range(50):
    body()
d.reset()
t0 = time.perf_counter()
for _ in range(steps):
    body()
    if d.crashed:
        d.reset()
dt = time.perf_counter() - t0
print(f"{'reference (CPU)' if '--reference' in sys.argv else 'learning_drones_amd scalar surface'}: {dt / steps * 1e6:.1f} us per timestep "
      f"(measure + translate_input_to_thrust, {steps} steps)")
