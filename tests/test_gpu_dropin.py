"""GPU tests of the drop-in object API (Drone / fly / evolve) against golden episodes produced by
the reference's own fly()."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fly_returns_reference_tuple(g3):
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.flight import fly
    for seed in (0, 1, 2, 3, 9):
        np.random.seed(seed)
        d = Drone(visualize=False, n_servers=0)              # float64 reference-tracking kernels
        reward, real_time, idx = fly(d, timesteps=2500, idx=7)
        fin = g3[f"final_{seed}"]
        assert (reward, idx) == (int(fin[0]), 7)
        assert abs(real_time - fin[1]) < 1e-12
        assert (d.coins, d.flight_time, d.crashed) == (int(fin[2]), int(fin[3]), bool(fin[4]))
        assert abs(d.distance_to_coin - fin[5]) < 1e-8
        assert np.allclose(d.position.ravel(), g3[f"state_{seed}"][-1, :3], rtol=1e-8, atol=1e-10)
        assert d.position.shape == (3, 1) and d.thrust.shape == (4, 1)


def test_manual_loop_with_measure_and_controller(g3):
    """Drive the reference's loop body by hand through Drone.measure / translate_input_to_thrust."""
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.objects.coins import Coin
    ei = 5
    np.random.seed(ei)
    d = Drone(visualize=False, n_servers=0)
    coin = Coin(np.array([[5.0], [5.0], [10.0]]), 2, 2000)
    wind = np.zeros((3, 1))
    steps = g3[f"steps_{ei}"]
    for t in range(60):
        d.reward += 2
        d.flight_time += 1
        assert not coin.is_in_vicinity(d.position)
        assert d.measure(wind) is False
        d.translate_input_to_thrust(coin.position)
        k = int(np.where(steps == t)[0][0])
        got = np.concatenate([d.position.ravel(), d.velocity.ravel(), d.angle.ravel(), d.angle_vel.ravel()])
        ref = g3[f"state_{ei}"][k]
        assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 1e-9, t
        assert np.allclose(d.thrust.ravel(), g3[f"thrust_{ei}"][k], rtol=1e-9, atol=1e-12)
        assert d.reward == g3[f"reward_{ei}"][t]
        assert d.sensors[2].current_acceleration == pytest.approx(g3[f"acc_{ei}"][k][2], rel=1e-9)


def test_fly_population_matches_scalar_fly():
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.flight import fly, fly_population
    np.random.seed(3)
    drones = [Drone(visualize=False, n_servers=0) for _ in range(6)]
    single = []
    for d in drones:
        c = Drone(visualize=False, n_servers=0)
        c.controller.weights = [w.copy() for w in d.controller.weights]
        single.append(fly(c, timesteps=500)[:2])
    scores, times = fly_population(drones, timesteps=500)
    assert list(zip(scores, times)) == single
    # rewards accumulate across calls unless reset (as in the reference)
    before = drones[0].reward
    drones[0].crashed = False
    fly_population(drones[:1], timesteps=1)
    assert drones[0].reward != before or drones[0].flight_time > 0


def test_evolve_two_generations(tmp_path):
    from learning_drones_amd.evolve.evolution import evolve, load_drone
    np.random.seed(0)
    logs = []
    ckpt = tmp_path / "best.json"
    best, history = evolve(n_generations=2, n_drones=8, timesteps=400, precision="f32", checkpoint=str(ckpt), log=logs.append)
    assert len(history) == 2 and len(history[0][2]) == 8 and len(logs) == 2
    assert history[1][1] >= history[0][1]                    # the global best never decreases
    again = load_drone(str(ckpt))
    assert np.array_equal(again.genes, best.genes)
