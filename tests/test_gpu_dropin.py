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


def test_foreign_controller_and_foreign_drone_plug_in(g3):
    """north_star: "evolve/brain.py controllers plug in unchanged".  The reference's contract for a controller is an
    object with a .weights list (evolve/brain.py:10-24, drone/drone.py:65,83-86) and for a drone the public attributes
    of drone/drone.py:32-76: a bare class carrying only .weights on this package's Drone, and a bare drone-shaped
    object that is not this package's class at all, both fly to the reference's golden (reward, real_time, idx)."""
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.evolve.brain import NeuralNetwork
    from learning_drones_amd.flight import fly

    class BareController:                                    # nothing but the weights
        def __init__(self, weights):
            self.weights = weights

    class BareSensor:
        def __init__(self):
            self.current_acceleration = 0.0
            self.last_acceleration = 0.0
            self.previous_pos = 0.0
            self.previous_vel = 0.0

    class BareDrone:                                         # drone/drone.py:32-76, public attributes only
        def __init__(self, controller):
            self.controller = controller
            self.position = np.array([[0.0], [0.0], [10.0]])
            self.velocity, self.angle, self.angle_vel = np.zeros((3, 1)), np.zeros((3, 1)), np.zeros((3, 1))
            self.thrust = np.zeros((4, 1))
            self.sensors = [BareSensor() for _ in range(6)]
            self.reward, self.coins, self.flight_time, self.crashed = 0, 0, 0, False
            self.distance_to_coin = np.inf

    for seed in (0, 2, 9):
        np.random.seed(seed)
        weights = NeuralNetwork(layer_spec={1: 10, 2: 10, 3: 4}).weights      # the draw Drone() makes (drone.py:65)
        fin = g3[f"final_{seed}"]
        d = Drone(visualize=False, n_servers=0)
        d.controller = BareController([w.copy() for w in weights])
        reward, real_time, idx = fly(d, timesteps=2500, idx=3)
        assert (reward, idx) == (int(fin[0]), 3) and abs(real_time - fin[1]) < 1e-12
        assert (d.coins, d.flight_time, d.crashed) == (int(fin[2]), int(fin[3]), bool(fin[4]))
        b = BareDrone(BareController([w.copy() for w in weights]))
        reward, real_time, idx = fly(b, timesteps=2500, idx=4)
        assert (reward, idx) == (int(fin[0]), 4) and abs(real_time - fin[1]) < 1e-12
        assert (b.coins, b.flight_time, b.crashed) == (int(fin[2]), int(fin[3]), bool(fin[4]))
        assert abs(b.distance_to_coin - fin[5]) < 1e-8
        assert np.allclose(b.position.ravel(), g3[f"state_{seed}"][-1, :3], rtol=1e-8, atol=1e-10)
        assert b.position.shape == (3, 1) and b.thrust.shape == (4, 1)
    with pytest.raises(TypeError):
        d.controller = object()
        fly(d, timesteps=10)


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


def test_reference_format_checkpoint_replays(tmp_path):
    """A best_drone_*.json in the reference's layout (evolution.py:85-87, jsonpickle string dumped as JSON) loads
    data-only and flies like a Drone built from the same weights (fly_best_drone.py:10-16)."""
    import json
    from test_dropin_host import _jsonpickle_ndarray
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.evolve.evolution import load_drone
    from learning_drones_amd.flight import fly
    np.random.seed(3)
    d = Drone(visualize=False, n_servers=0)
    doc = {"py/object": "drone.drone.Drone", "controller": {"py/object": "evolve.brain.NeuralNetwork",
           "layer_spec": {str(k): v for k, v in d.controller.layer_spec.items()},
           "weights": [_jsonpickle_ndarray(w, i == 0) for i, w in enumerate(d.controller.weights)]}}
    path = tmp_path / "best_drone_01_01_2020.json"
    path.write_text(json.dumps(json.dumps(doc)))
    again = load_drone(str(path))
    assert all(np.array_equal(a, b) for a, b in zip(again.controller.weights, d.controller.weights))
    fly(d, timesteps=300); fly(again, timesteps=300)
    assert (again.reward, again.coins, again.flight_time, again.crashed) == (d.reward, d.coins, d.flight_time, d.crashed)


# ------------------------------------------------------------------------------------------- device GA
def _breed_once(n, hidden, fitness, ga, elite_slot=None, seed_genes=0):
    import torch
    from learning_drones_amd.population import Population
    pop = Population(n, hidden)
    rng = np.random.default_rng(seed_genes)
    genes = rng.normal(0, 1, (n, pop.P)).astype(np.float32)
    pop.set_genomes(genes)
    elite = pop.genome_of(elite_slot) if elite_slot is not None else None
    pop.breed(torch.as_tensor(fitness, dtype=torch.int32), ga, elite=elite)
    torch.cuda.synchronize()
    return genes, pop.get_genomes().cpu().numpy(), pop.last_parents.cpu().numpy(), pop.last_cuts.cpu().numpy(), pop


def test_device_ga_selection_copies_parents():
    """No crossover, no mutation: every child is an exact copy of its selected parent, the better of
    a tournament wins with probability p_tournament, and slot 0 is the elite."""
    from learning_drones_amd.params import GaParams
    n, hidden = 4096, (10, 10)
    fitness = np.random.default_rng(1).permutation(n).astype(np.int32)          # all distinct
    ga = GaParams.reference(seed=7, generation=3, p_crossover=0.0, mutation_rate=0.0)
    genes, children, parents, cuts, _ = _breed_once(n, hidden, fitness, ga, elite_slot=int(np.argmax(fitness)))
    assert (cuts == -1).all()
    assert np.array_equal(children[0], genes[np.argmax(fitness)])                # elitism
    assert np.array_equal(children[1:], genes[parents[1:]])                      # exact copies
    assert parents.min() >= 0 and parents.max() < n
    # selection pressure: P(parent fitness rank in upper half) for a binary tournament with p = 0.6
    # P(winner is the better of two) = 0.6  ->  E[rank quantile] = 0.6 * 2/3 + 0.4 * 1/3 = 0.5333
    q = fitness[parents] / (n - 1)
    assert abs(q.mean() - (0.6 * 2 / 3 + 0.4 / 3)) < 0.01
    # a different generation or seed draws different parents; the same inputs reproduce them
    again = _breed_once(n, hidden, fitness, ga)[2]
    assert np.array_equal(again, parents)
    other = _breed_once(n, hidden, fitness, GaParams.reference(seed=7, generation=4, p_crossover=0.0, mutation_rate=0.0))[2]
    assert (other != parents).mean() > 0.9


def test_device_ga_tie_rule():
    """evolve/util.py:30-45: on equal fitness the second index wins when r < p, else the first —
    so with all-equal fitness and p = 1 every tournament returns its second draw."""
    from learning_drones_amd.params import GaParams
    n = 1024
    ga1 = GaParams.reference(seed=3, p_tournament=1.0, p_crossover=0.0, mutation_rate=0.0)
    ga0 = GaParams.reference(seed=3, p_tournament=0.0, p_crossover=0.0, mutation_rate=0.0)
    p1 = _breed_once(n, (10, 10), np.zeros(n, np.int32), ga1)[2]
    p0 = _breed_once(n, (10, 10), np.zeros(n, np.int32), ga0)[2]
    assert (p1 != p0).mean() > 0.99                                              # second vs first draw
    fit = np.arange(n, dtype=np.int32)
    best = _breed_once(n, (10, 10), fit, ga1)[2]
    worst = _breed_once(n, (10, 10), fit, ga0)[2]
    assert np.array_equal(np.sort(np.stack([best, worst]), axis=0)[1], best)     # p=1 picks the max of each pair


def test_device_ga_crossover_structure():
    """p_crossover = 1: per layer, child A = parent A before the cut and parent B from the cut on,
    child B the mirror image (evolve/util.py:58-68)."""
    from learning_drones_amd.params import GaParams, MlpSpec
    n, hidden = 512, (16, 16)
    spec = MlpSpec.from_hidden(hidden)
    ga = GaParams.reference(seed=11, p_crossover=1.0, mutation_rate=0.0)
    fitness = np.random.default_rng(2).integers(-5000, 5000, n).astype(np.int32)
    genes, children, parents, cuts, _ = _breed_once(n, hidden, fitness, ga)
    assert (cuts >= 0).all()
    sizes = [a * b for a, b in spec.layer_shapes]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    for c in range(n):
        a, b = genes[parents[c]], genes[parents[c ^ 1]]
        want = a.copy()
        for l, size in enumerate(sizes):
            cut = cuts[c // 2, l]
            assert 0 <= cut < size
            want[offs[l] + cut:offs[l + 1]] = b[offs[l] + cut:offs[l + 1]]
        assert np.array_equal(children[c], want), c
    # rate check at the reference's p = 0.05 per layer
    ga = GaParams.reference(seed=12, p_crossover=0.05, mutation_rate=0.0)
    cuts = _breed_once(8192, hidden, np.zeros(8192, np.int32), ga)[3]
    assert abs((cuts >= 0).mean() - 0.05) < 0.006
    for l, size in enumerate(sizes):
        got = cuts[:, l][cuts[:, l] >= 0]
        assert got.min() >= 0 and got.max() < size


def test_device_ga_mutation_statistics():
    """Per-gene Bernoulli(0.05) x N(0, 0.3) (drone/drone.py:167-172)."""
    from learning_drones_amd.params import GaParams
    n, hidden = 2048, (32, 32)
    ga = GaParams.reference(seed=5, p_crossover=0.0)
    genes, children, parents, _, _ = _breed_once(n, hidden, np.zeros(n, np.int32), ga)
    delta = (children - genes[parents]).astype(np.float64)
    mutated = delta != 0
    assert abs(mutated.mean() - 0.05) < 0.001
    d = delta[mutated]
    assert abs(d.mean()) < 0.002 and abs(d.std() - 0.3) < 0.003
    assert abs(np.mean(np.abs(d) < 0.3) - 0.6827) < 0.005                         # Gaussian shape
    assert abs(mutated.mean(axis=0).std() - np.sqrt(0.05 * 0.95 / n)) < 0.002    # no gene position favoured


def test_device_evolution_loop():
    """Whole loop on the device: the global best never decreases, slot 0 holds the unmutated global
    best, and re-evaluating it next generation reproduces its score exactly (deterministic rollout)."""
    import torch
    from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
    pop = ShardedPopulation(2048, (16, 16))
    pop.local.init_genomes_reference(seed=1)
    evo = DeviceEvolution(pop, seed=42)
    prev_global = None
    for g in range(6):
        best, gbest, records = evo.step(600)
        if prev_global is not None:
            assert int(records[0, 0]) == prev_global          # the elite flies exactly as before
            assert best >= prev_global
        assert torch.equal(pop.local.genome_of(0), evo.elite)     # slot 0 = global best, unmutated
        assert not torch.equal(pop.local.genome_of(1), evo.elite)
        prev_global = gbest
    scores = [h[1] for h in evo.history]
    assert scores == sorted(scores) and evo.generation == 6
    assert int(pop.local.flight_time.sum()) == 0                   # children start reset


def test_breed_from_gathered_segments():
    """The multi-GPU generation step breeds from the all-gathered genome shards (n_segments arrays of
    n_per_segment drones back to back); that addressing must give the same children as breeding from
    one whole-population array."""
    import torch
    from learning_drones_amd.params import GaParams
    from learning_drones_amd.population import Population
    n, hidden, world = 512, (16, 16), 4
    rng = np.random.default_rng(0)
    whole = Population(n, hidden)
    genes = rng.normal(0, 1, (n, whole.P)).astype(np.float32)
    whole.set_genomes(genes)
    fitness = torch.as_tensor(rng.integers(-4000, 4000, n).astype(np.int32))
    ga = GaParams.reference(seed=9, generation=2, p_crossover=0.5)
    elite = whole.genome_of(17)
    whole.breed(fitness, ga, elite=elite)
    want = whole.get_genomes().cpu().numpy()
    shards = []
    for r in range(world):
        p = Population(n // world, hidden)
        p.set_genomes(genes[r * (n // world):(r + 1) * (n // world)])
        shards.append(p)
    gathered = torch.stack([p.genomes for p in shards]).contiguous()      # what all_gather_into_tensor builds
    got = []
    for r, p in enumerate(shards):
        p.breed(fitness, ga, elite=elite, child_begin=r * (n // world), src=gathered, n_per_segment=n // world,
                n_segments=world)
        got.append(p.get_genomes().cpu().numpy())
    assert np.array_equal(np.concatenate(got), want)


def test_checkpoint_resume_and_replay(tmp_path):
    import torch
    from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
    from learning_drones_amd.population import Population
    pop = ShardedPopulation(256, (10, 10))
    pop.local.init_genomes_reference(seed=3)
    evo = DeviceEvolution(pop, seed=5)
    for _ in range(2):
        evo.step(300)
    evo.save(str(tmp_path / "evo.npz"))
    a = [evo.step(300)[:2] for _ in range(2)]
    genes_a = pop.local.get_genomes().cpu().numpy()
    pop2 = ShardedPopulation(256, (10, 10))
    evo2 = DeviceEvolution(pop2)
    evo2.load(str(tmp_path / "evo.npz"))
    b = [evo2.step(300)[:2] for _ in range(2)]
    assert a == b and np.array_equal(pop2.local.get_genomes().cpu().numpy(), genes_a)      # bit-exact resume
    # population checkpoint mid-episode, then replay log in the reference's Results.csv column order
    p = Population(8, (10, 10))
    p.init_genomes_reference(seed=1)
    for _ in range(50):
        p.step()
    p.save(str(tmp_path / "pop.npz"))
    q = Population.load(str(tmp_path / "pop.npz"))
    log_p = p.record_trajectory(100, slots=(0, 5), csv_prefix=str(tmp_path / "drone_"))
    log_q = q.record_trajectory(100, slots=(0, 5))
    assert np.array_equal(log_p, log_q) and log_p.shape == (2, 100, 25)
    assert np.allclose(log_p[0, :, 0], np.arange(100) * 0.01)
    assert np.allclose(log_p[1, -1, 4:7], p.position[:, 5].double().cpu().numpy())
    assert np.array_equal(log_p[0, 0, 22:25], [5.0, 5.0, 10.0])                 # flying to the first coin
    text = (tmp_path / "drone_5.csv").read_text().splitlines()
    assert text[0] == "# " + Population.RESULTS_CSV_HEADER and len(text) == 101   # the reference's Results.csv header
    assert np.allclose(np.loadtxt(tmp_path / "drone_5.csv", delimiter=","), log_p[1])
    assert torch.equal(p.fstate, q.fstate) and torch.equal(p.istate, q.istate)


# ------------------------------------------------------------------------------------------- config 1 on the device (8f-4)
def test_pid_policy_matches_reference_episodes(g4):
    """All ten gain sets of choose_paramset.py as one batch; the three recorded in golden G4 must
    reproduce the reference's trajectories, scores and coin counts (float64 kernels), and config 1
    (paramset "11", 1000 steps) must end with score 3000 and 3 coins."""
    from learning_drones_amd.pid import pid_episodes
    names = [str(s) for s in g4["paramset_names"]]
    out = pid_episodes(g4["paramsets"], 1600, coins=g4["coins"], precision="f64", trace=True)
    for name in ("00", "11", "22"):
        k = names.index(name)
        S, U = g4[f"state_{name}"], g4[f"thrust_{name}"]
        steps = int(out["steps"][k])
        assert steps == len(S)
        tr = out["trace"][:steps, k, :]
        # closed loop over ~1 500 steps; the derivative term (kd * diff / dt) amplifies the few-ulp
        # differences between device and glibc sin/cos/tan
        assert np.max(np.abs(tr[:, :12] - S)) < 1e-7 and np.max(np.abs(tr[:, 12:] - U)) < 1e-7
        # these episodes end by taking the last coin in the iteration after the last recorded step
        assert steps < 1600 and int(out["score"][k]) == int(g4[f"score_{name}"][-1]) - 1 + 1000
        assert int(out["coins"][k]) == int(g4[f"target_{name}"][-1]) == 3
    cfg1 = pid_episodes(g4["paramsets"][names.index("11")][None], 1000, coins=g4["coins"], precision="f64")
    assert [int(cfg1["score"][0]), int(cfg1["coins"][0]), int(cfg1["steps"][0])] == [int(v) for v in g4["cfg1_final"]]
    # float32 policy: same coin counts, trajectory within the float32 open-loop envelope early on
    f32 = pid_episodes(g4["paramsets"], 1600, coins=g4["coins"], precision="f32", trace=True)
    k = names.index("11")
    assert np.max(np.abs(f32["trace"][:300, k, :3] - g4["state_11"][:300, :3])) < 1e-2
    assert np.array_equal(f32["coins"], out["coins"])


@pytest.mark.parametrize("permuted", [False, True])
def test_alltoall_parent_exchange_equals_allgather(permuted):
    """Multi-GPU generation step, both exchange variants, simulated on one GPU with one thread per
    virtual rank and a mailbox standing in for the collective: breeding from the fetched distinct
    parents (all-to-all) must give exactly the children that breeding from the all-gathered shards
    gives, and those must equal the single-population result.  permuted: the previous generation was placed
    out of canonical order (place="parent"), so the selected canonical parents are translated to global SLOT
    indices (slot_of) before owner = slot // n_local is computed -- same children."""
    import threading
    import torch
    from learning_drones_amd.parallel import exchange_parent_genomes
    from learning_drones_amd.params import GaParams
    from learning_drones_amd.population import Population
    n, hidden, world = 768, (16, 16), 4
    n_local = n // world
    rng = np.random.default_rng(0)
    genes = rng.normal(0, 1, (n, 596)).astype(np.float32)
    fitness = torch.as_tensor(rng.integers(-4000, 4000, n).astype(np.int32))
    ga = GaParams.reference(seed=9, generation=5, p_crossover=0.5)
    whole = Population(n, hidden)
    whole.set_genomes(genes)
    elite = whole.genome_of(3)
    whole.breed(fitness, ga, elite=elite)
    want = whole.get_genomes().cpu().numpy()

    slot_of, by_slot = None, genes
    if permuted:
        perm = rng.permutation(n)
        slot_of = torch.as_tensor(perm, dtype=torch.int64, device="cuda")          # canonical index -> slot
        by_slot = np.empty_like(genes)
        by_slot[perm] = genes
    shards = []
    for r in range(world):
        p = Population(n_local, hidden)
        p.set_genomes(by_slot[r * n_local:(r + 1) * n_local])
        shards.append(p)
    original = [p.genomes.clone() for p in shards]
    # the all-gather variant: every rank breeds from all the shards back to back (ld_ga_breed's segments)
    gathered = torch.cat(original, dim=0)
    for r in range(world):
        shards[r].breed(fitness, ga, elite=elite, child_begin=r * n_local, src=gathered, n_per_segment=n_local, n_segments=world,
                        slot_of=slot_of)
    got_gather = np.concatenate([p.get_genomes().cpu().numpy() for p in shards])
    assert np.array_equal(got_gather, want)
    for r in range(world):
        shards[r].genomes = original[r].clone()

    class ThreadComm:
        box, barrier = {}, threading.Barrier(world)

        def __init__(self, rank):
            self.rank, self.world = rank, world

        def all_to_all(self, send):
            ThreadComm.box[self.rank] = send
            ThreadComm.barrier.wait()
            recv = [ThreadComm.box[r][self.rank] for r in range(world)]
            ThreadComm.barrier.wait()
            return recv

    errors = []

    def run(r):
        try:
            torch.cuda.set_device(0)
            shards[r].breed(fitness, ga, elite=elite, child_begin=r * n_local, slot_of=slot_of,
                            fetch_parents=lambda parents: exchange_parent_genomes(original[r], n_local, parents, ThreadComm(r)))
        except Exception as e:                                         # surfaced below
            errors.append(e)
            ThreadComm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    got = np.concatenate([p.get_genomes().cpu().numpy() for p in shards])
    assert np.array_equal(got, want)


def test_fly_keyword_variants(g7):
    """fly(drone, timesteps=, delta_t=, punish_time=, reward_time=) against the reference's own
    results for the same seeded drones: the airtime term switches sign / off, delta_t only scales the
    returned time, timesteps bounds the episode."""
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.flight import fly
    from learning_drones_amd.params import MlpSpec, unflatten_weights
    spec = MlpSpec.from_hidden((10, 10))
    for w, a, fin in zip(g7["weights"], g7["args"], g7["final"]):
        d = Drone(visualize=False, n_servers=0)
        d.controller.weights = unflatten_weights(w, spec)
        reward, real_time, idx = fly(d, timesteps=int(a[0]), delta_t=float(a[1]), punish_time=bool(a[2]),
                                     reward_time=bool(a[3]), idx=3)
        assert (reward, idx, d.coins, d.flight_time, d.crashed) == (int(fin[0]), 3, int(fin[2]), int(fin[3]), bool(fin[4]))
        assert abs(real_time - fin[1]) < 1e-9 and abs(d.distance_to_coin - fin[5]) < 1e-8
    with pytest.raises(NotImplementedError):
        fly(Drone(visualize=False, n_servers=0), visualize=True)


def test_c_host_program_uses_the_abi_without_python(tmp_path):
    """examples/c_host.c: plain C + hipMalloc against libld_drones.so — the boundary has no Python or
    PyTorch in it.  Its printed record must equal what the Population wrapper gets for the same run."""
    import shutil
    import subprocess
    from pathlib import Path
    from learning_drones_amd.params import coin_chain_reference
    from learning_drones_amd.population import Population
    repo = Path(__file__).resolve().parent.parent
    gcc = shutil.which("gcc")
    exe = tmp_path / "c_host"
    subprocess.check_call([gcc, str(repo / "examples" / "c_host.c"), f"-I{repo / 'include'}", "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", f"-L{repo / 'learning-drones_amd'}", "-lld_drones", "-L/opt/rocm/lib",
                           "-lamdhip64", f"-Wl,-rpath,{repo / 'learning-drones_amd'}", "-Wl,-rpath,/opt/rocm/lib",
                           "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    pop = Population(100, (16, 16), coins=coin_chain_reference(8))
    pop.fly(500)
    r = pop.results()[0]
    want = (f"reward {int(r['reward'])} flight_time {int(r['flight_time'])} coins {int(r['coins'])} "
            f"crashed {int(r['crashed'])} distance {float(r['distance_to_coin']):.4f}, all drones equal: yes")
    assert want in out, out
    assert int(r["crashed"]) == 1                    # zero genomes climb through the ceiling


def test_fly_best_drone_replay(tmp_path, g3, capsys):
    """Save a controller, replay it with the fly_best_drone entry point: the printed score is the
    reference's score for that controller and the Results.csv has the reference's layout."""
    from learning_drones_amd import fly_best_drone
    from learning_drones_amd.drone.drone import Drone
    from learning_drones_amd.evolve.evolution import save_drone
    from learning_drones_amd.params import MlpSpec, unflatten_weights
    ei = 37                                                     # the (10, 10) hover controller episode
    assert str(g3[f"kind_{ei}"]) == "hover"
    d = Drone(visualize=False, n_servers=0)
    d.controller.weights = unflatten_weights(g3[f"weights_{ei}"], MlpSpec.from_hidden((10, 10)))
    save_drone(d, str(tmp_path / "best.json"))
    T = int(g3[f"timesteps_{ei}"])
    assert fly_best_drone.main(["fly_best_drone", str(tmp_path / "best.json"), str(T), str(tmp_path / "rewatch")]) == 0
    out = capsys.readouterr().out
    fin = g3[f"final_{ei}"]
    assert f"Final Score: {int(fin[0])}, coins {int(fin[2])}, crashed False" in out
    log = np.loadtxt(tmp_path / "rewatch" / "Results.csv", delimiter=",")
    assert log.shape == (T, 25)
    assert np.allclose(log[-1, 4:7], g3[f"state_{ei}"][-1, :3], atol=1e-6)
