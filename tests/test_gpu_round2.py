"""GPU tests added in round 2: the unbounded coin chain at and beyond the table's end, checkpoints of
non-default physics / episode blocks, the accelerations-only entry point and its drop-in shim, the
reference's marginal ("noisy") closed-loop episodes on the float32 kernels, BASELINE config 4 at its
shard size, and ShardedPopulation over several ranks on one GPU at the config-4 and config-5 shard sizes."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO
from learning_drones_amd.params import EpisodeParams, MlpSpec, SimParams, coin_chain_reference
from test_gpu_parity import CHAOTIC

pytestmark = pytest.mark.gpu
RTOL_F32 = 1e-4                                  # BASELINE.json north_star: "matching CPU reference trajectories to rtol=1e-4"


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.fixture(scope="module")
def Population(torch):
    from learning_drones_amd.population import Population
    return Population


def _hover_genes(hidden, n, sigma=1e-4, seed=0):
    from learning_drones_amd import controllers
    rng = np.random.default_rng(seed)
    bases = [controllers.hover_net(hidden, gains=controllers.hover_gains(pole_xy=p)) for p in (1.2, 1.6, 2.0, 2.5)]
    return np.stack([np.concatenate([w.ravel() for w in controllers.perturbed(bases[i % 4], sigma, rng)])
                     for i in range(n)]).astype(np.float32).astype(np.float64)


# ------------------------------------------------------------------------------------------- coin chain
@pytest.mark.parametrize("precision,engine", [("f64", "step"), ("f32", "step"), ("f32", "resident")])
def test_coin_chain_beyond_the_table(Population, oracle, precision, engine):
    """flight/__init__.py:47-52 respawns a coin forever.  A one-row table with coin_respawn (the default
    Population) must give what a long finite table of the same chain gives, on every engine and on the
    oracle, for drones that collect well past the first rows; and a finite table stops at its last row."""
    hidden, n, T = (16, 16), 96, 2500
    genes = _hover_genes(hidden, n)
    spec = MlpSpec.from_hidden(hidden)
    mode = oracle.FAITHFUL_F64 if precision == "f64" else oracle.CLOSED_F32
    threads = min(oracle.max_threads(), 16)

    def run(**kw):
        pop = Population(n, hidden, precision=precision, **kw)
        pop.set_genomes(genes)
        pop.fly(T, engine=engine)
        return pop, pop.results()

    pop_chain, chain = run()                                             # reference episode: 1 row + respawn
    assert pop_chain.episode.n_coins == 1 and pop_chain.episode.coin_respawn == 1
    _, table = run(coins=coin_chain_reference(40))                       # finite, long enough for everyone
    _, two = run(coins=coin_chain_reference(2), coin_respawn=True)       # continuation from a 2-row table
    _, capped = run(coins=coin_chain_reference(2))                       # finite: ends at 2
    assert chain["coins"].max() >= 6 and chain["coins"].max() < 40
    for f in ("reward", "coins", "flight_time", "crashed", "distance_to_coin"):
        assert np.array_equal(chain[f], table[f]), f
        assert np.array_equal(chain[f], two[f]), f
    assert capped["coins"].max() == 2 and (capped["coins"] <= 2).all()
    # a drone that took both coins of the finite chain keeps the last row as its target (controller input and
    # distance penalty); on the unbounded chain the same drone flies on to the third coin and beyond
    took_all = capped["coins"] == 2
    assert took_all.any() and (chain["coins"][took_all] >= 2).all() and (chain["coins"][took_all] > 2).any()
    assert (capped["reward"][took_all] != chain["reward"][took_all]).any()
    assert (capped["distance_to_coin"][took_all] < 3.0).mean() > 0.8     # hovering at the last coin of the table
    # the oracle restates the same rule
    ref_chain = oracle.fly(pop_chain.sim, pop_chain.episode, spec, genes, T, n_threads=threads, mode=mode)
    ep_capped = EpisodeParams.reference(n_coins=2, coin_respawn=False)
    ref_capped = oracle.fly(pop_chain.sim, ep_capped, spec, genes, T, coins=coin_chain_reference(2), n_threads=threads, mode=mode)
    if precision == "f64":
        assert np.array_equal(chain["reward"], ref_chain.reward) and np.array_equal(chain["coins"], ref_chain.coins)
        assert np.array_equal(capped["reward"], ref_capped.reward) and np.array_equal(capped["coins"], ref_capped.coins)
    else:                                                                # stable flyers: float32 GPU == float32 oracle
        assert np.mean(chain["reward"] == ref_chain.reward) >= 0.98 and np.array_equal(chain["coins"], ref_chain.coins)
        assert np.mean(capped["reward"] == ref_capped.reward) >= 0.98 and np.array_equal(capped["coins"], ref_capped.coins)


def test_reference_hover_episodes_on_the_unbounded_chain(Population, g3):
    """The reference's own coin-collecting episodes (3-8 coins) through the default Population, whose coin
    table is ONE row continued in the kernel: every integer outcome exact (float64 kernels)."""
    for ei in range(int(g3["n_episodes"])):
        if str(g3[f"kind_{ei}"]) != "hover":
            continue
        hidden = tuple(int(h) for h in g3[f"hidden_{ei}"])
        pop = Population(1, hidden, precision="f64")
        pop.set_genomes(g3[f"weights_{ei}"][None])
        pop.fly(int(g3[f"timesteps_{ei}"]))
        r, fin = pop.results()[0], g3[f"final_{ei}"]
        assert (int(r["reward"]), int(r["coins"]), int(r["flight_time"]), bool(r["crashed"])) == \
               (int(fin[0]), int(fin[2]), int(fin[3]), bool(fin[4]))
        assert abs(r["distance_to_coin"] - fin[5]) < 1e-8 and int(r["coins"]) >= 3


# ------------------------------------------------------------------------------------------- checkpoints
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_checkpoint_keeps_physics_and_episode_blocks(Population, torch, tmp_path, precision):
    """A population with wind, payload, punish_time, other crash planes, a finite coin table and walls,
    saved mid-episode and continued after load(), must end bit for bit like the uninterrupted run
    (the float64 distance included)."""
    hidden, n = (16, 16), 130
    genes = _hover_genes(hidden, n, sigma=0.004, seed=3)
    sim = SimParams.reference(wind=(0.4, -0.2, 0.1), mass_payload=0.03, x_payload=0.02, y_payload=-0.01, dt=0.008)
    ep = EpisodeParams.reference(punish_time=True, reward_time=False, coin_respawn=False)      # a finite coin table
    ep.z_ceiling, ep.angle_limit = 30.0, 0.6
    walls = np.array([[6.0, 4.0, 8.0, 7.0, 8.0, 12.0]])

    def make():
        pop = Population(n, hidden, precision=precision, sim=SimParams.from_buffer_copy(sim), episode=EpisodeParams.from_buffer_copy(ep),
                         coins=coin_chain_reference(2), walls=walls)
        pop.set_genomes(genes)
        return pop

    straight = make()
    for _ in range(300):
        straight.step()
    straight.finalize()
    first = make()
    for _ in range(120):
        first.step()
    first.save(str(tmp_path / "mid.npz"))
    resumed = Population.load(str(tmp_path / "mid.npz"))
    assert bytes(resumed.sim) == bytes(first.sim) and bytes(resumed.episode) == bytes(first.episode)
    assert resumed.episode.reward_time == -2 and resumed.episode.coin_respawn == 0 and resumed.sim.dt == 0.008
    for _ in range(180):
        resumed.step()
    resumed.finalize()
    assert torch.equal(resumed.fstate, straight.fstate) and torch.equal(resumed.istate, straight.istate)
    assert torch.equal(resumed.records, straight.records)
    if precision == "f64":
        assert torch.equal(resumed.distance, straight.distance) and float(resumed.distance.max()) > 0
    # a finalized float64 population keeps its distances through save / load
    straight.save(str(tmp_path / "end.npz"))
    again = Population.load(str(tmp_path / "end.npz"))
    assert np.array_equal(again.results(), straight.results())
    crashed = straight.results()["crashed"]
    assert crashed.any() and not crashed.all()


# ------------------------------------------------------------------------------------------- accelerations
def test_accelerations_entry_point_and_shim(torch, g1):
    """ld_accelerations_* = QuadcopterPhysics.calculate_accelerations (physical_model.py:75-122) alone, against
    the reference's own vectors; and the drop-in shim drone.quadcopter.calculate_accelerations."""
    import ctypes as C
    from learning_drones_amd import _native
    from learning_drones_amd.drone.drone import Drone
    n = 1500                                                             # the wind-free vectors of G1
    state, thrust, acc = g1["state_in"][:n], g1["thrust"][:n], g1["acc_out"][:n]
    sim = SimParams.reference()
    tame = np.abs(np.cos(state[:, 7])) > 0.05                            # float32: away from the cos(pitch) = 0 pole
    for name, dtype, tol, rows in (("f64", torch.float64, 1e-11, np.ones(n, bool)), ("f32", torch.float32, 1e-3, tame)):
        ang = torch.as_tensor(state[:, 6:9].T.copy()).to("cuda", dtype).contiguous()
        thr = torch.as_tensor(thrust.T.copy()).to("cuda", dtype).contiguous()
        lin, rot = torch.empty_like(ang), torch.empty_like(ang)
        fn = getattr(_native.lib(), f"ld_accelerations_{name}")
        _native.check(fn(C.byref(sim), C.c_void_p(ang.data_ptr()), C.c_void_p(thr.data_ptr()), C.c_void_p(lin.data_ptr()),
                         C.c_void_p(rot.data_ptr()), n, None), "ld_accelerations")
        got = torch.cat([lin, rot]).t().double().cpu().numpy()
        scale = np.maximum(np.abs(acc), 1.0)
        assert np.max((np.abs(got - acc) / scale)[rows]) < tol, name
        assert fn(None, None, None, None, None, 1, None) == -1 and fn(C.byref(sim), None, None, None, None, 0, None) == 0
    # wind + payload vectors: the reference's calculate_accelerations outputs themselves
    pp = g1["payload_params"]
    from learning_drones_amd.drone.quadcopter import QuadcopterPhysics
    q = QuadcopterPhysics(mass_payload=pp[0], x_payload=pp[1], y_payload=pp[2], I_x=pp[3], I_y=pp[4], I_z=pp[5])
    for i in (0, 57, len(g1["payload_angle"]) - 1):
        lin, rot = q.calculate_accelerations(g1["payload_angle"][i].reshape(3, 1), g1["payload_wind"][i].reshape(3, 1),
                                             g1["payload_thrust"][i].reshape(4, 1), True)
        ref = np.concatenate([g1["payload_lin_acc"][i], g1["payload_rot_acc"][i]])
        assert np.max(np.abs(np.concatenate([lin, rot]).ravel() - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-11
    d = Drone(visualize=False)
    k = 7
    lin, rot = d.quadcopter.calculate_accelerations(state[k, 6:9].reshape(3, 1), np.zeros((3, 1)), thrust[k].reshape(4, 1),
                                                    lin_acc_drone_2_lab=True)
    assert lin.shape == (3, 1) and rot.shape == (3, 1)
    assert np.max(np.abs(np.concatenate([lin, rot]).ravel() - acc[k]) / np.maximum(np.abs(acc[k]), 1.0)) < 1e-11
    assert d.quadcopter.G == pytest.approx(3.924) and d.quadcopter.c_f == 5 and d.quadcopter.m_c == 0.2
    # lin_acc_drone_2_lab=False: at rest and level the body frame is the lab frame (free fall, no rotation), and the call
    # leaves the identity in self.Rot (test_accelerations_in_both_frames_match_the_reference has the reference's vectors)
    lin, rot = d.quadcopter.calculate_accelerations(np.zeros((3, 1)), np.zeros((3, 1)), np.zeros((4, 1)), lin_acc_drone_2_lab=False)
    assert lin.dtype == np.float32 and np.allclose(lin.ravel(), [0, 0, -9.81], rtol=1e-6) and not rot.any()
    assert np.array_equal(d.quadcopter.Rot, np.identity(3))
    nn = d.controller                                                     # evolve/brain.py:26-39 helpers
    assert np.array_equal(nn.relu(np.array([[-1.0], [2.0], [0.0]])), [[0.0], [2.0], [0.0]]) and nn._relu(-3) == 0.0
    assert nn.sigmoid(0.0) == 0.5 and abs(nn.sigmoid(2.5) - 1 / (np.exp(-1.0) + 1)) < 1e-15


# ------------------------------------------------------------------------------------------- float32 on marginal flights
def test_noisy_reference_episodes_f32(Population, g3):
    """The six marginal 32-32 episodes of golden G3 (hover controllers with destabilising gene noise: one
    survives 2 000 steps, five tumble and crash after 144-460) on the float32 production kernels.  While
    a flight is still regular the float32 trajectory follows the reference to north_star's rtol = 1e-4 of
    the distance flown; the comparison stops where the reference flight starts tumbling (|angle| beyond the
    stability limit: from there a 1-ulp difference is amplified ~10x every 15 steps, tests above), and
    the crash itself must come within a few steps of the reference's."""
    seen = 0
    for ei in range(int(g3["n_episodes"])):
        if str(g3[f"kind_{ei}"]) != "noisy":
            continue
        hidden = tuple(int(h) for h in g3[f"hidden_{ei}"])
        steps, ref, fin = g3[f"steps_{ei}"], g3[f"state_{ei}"], g3[f"final_{ei}"]
        pop = Population(1, hidden, precision="f32")
        pop.set_genomes(g3[f"weights_{ei}"][None])
        tumbling = np.nonzero((np.abs(ref[:, 6]) > np.pi / 4) | (np.abs(ref[:, 7]) > np.pi / 4))[0]
        horizon = int(steps[tumbling[0]]) if len(tumbling) else int(steps[-1])
        done, worst, checked = 0, 0.0, 0
        for s, want in zip(steps, ref):
            if s + 1 > horizon:
                break
            pop.steps(int(s) + 1 - done, 0)                              # state after loop iteration s
            done = int(s) + 1
            got = pop.state_numpy()[0, :12]
            flown = np.linalg.norm(want[:3] - [0, 0, 10]) + 1.0
            worst = max(worst, np.max(np.abs(got[:3] - want[:3])) / flown)
            checked += 1
        assert checked >= 5 and worst < RTOL_F32, (ei, horizon, worst)
        pop.steps(int(g3[f"timesteps_{ei}"]) - done, 0)
        pop.finalize()
        r = pop.results()[0]
        assert bool(r["crashed"]) == bool(fin[4]), ei
        if fin[4]:
            assert abs(int(r["flight_time"]) - int(fin[3])) <= max(3, int(0.1 * fin[3])), (ei, int(r["flight_time"]), int(fin[3]))
        else:
            assert int(r["flight_time"]) == int(fin[3]) and int(r["coins"]) == int(fin[2])
        seen += 1
    assert seen == 6


# ------------------------------------------------------------------------------------------- config 4 and sharding
def _replicated(hidden, m, n, seed):
    """n genomes drawn from an m-drone mixed population under a fixed permutation: (small genomes [m, P],
    source index of every big-population slot [n])."""
    from test_gpu_parity import _mixed_genomes
    spec, genes_m, _ = _mixed_genomes(hidden, m, seed=seed)
    src = (np.arange(n) % m)[np.random.default_rng(seed).permutation(n)]
    return genes_m, src


def test_full_size_config4_shard(Population, torch):
    """One GPU's share of BASELINE config 4: 32 768 drones x 4 000 steps, 15-32-32-4, 3-coin chain (the
    launch geometry this shard gets: 512 timesteps per launch, all 32 768 drones in one launch, one workgroup of 8 wavefronts per
    genome block (fly_wide_kernel), all of its 64 drones' genes held in registers (304 words per drone) and LDS (124) for the launch).  Size-independent
    properties: genomes replicated from a 256-drone population under a random permutation give the
    permuted small-population results bit for bit; a re-run is bit-identical; one launch per timestep
    and the resident engine give the same records."""
    hidden, n, T, m = (32, 32), 32768, 4000, 256
    genes_m, src = _replicated(hidden, m, n, seed=4)
    coins = coin_chain_reference(3)
    small = Population(m, hidden, coins=coins)
    small.set_genomes(genes_m)
    small.fly(T)
    want = small.results()
    pop = Population(n, hidden, coins=coins)
    pop.set_genomes(torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")[torch.as_tensor(src, device="cuda")])
    geo = pop.rollout_geometry(T)
    assert geo == {"steps_per_launch": 512, "waves_per_block": 8, "launches": 8, "tile_drones": 32768, "lds_words": 124, "register_words": 304, "streamed_words": 0}
    pop.fly(T)
    got = pop.results()
    for f in ("reward", "flight_time", "coins", "crashed", "distance_to_coin"):
        assert np.array_equal(got[f], want[f][src]), f
    rec = pop.records.clone()
    for kw in (dict(steps_per_launch=1), dict(tile_drones=16384, waves_per_block=16), dict(tile_drones=4096 + 64, waves_per_block=2),
               dict(waves_per_block=11), dict(engine="resident"), {}):
        pop.reset()
        pop.fly(T, **kw)
        assert torch.equal(pop.records, rec), kw
    assert want["coins"].max() == 3 and want["crashed"].any() and not want["crashed"].all()


def _shard_worker(rank, world, port, cfg, out_dir):
    """One rank of a ShardedPopulation on the shared GPU (gloo carries the collectives)."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
    import torch
    import torch.distributed as dist
    from learning_drones_amd.parallel import ShardedPopulation
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        hidden, n_total, T, m, seed, n_coins, walls = cfg
        genes_m, src = _replicated(hidden, m, n_total, seed)
        pop = ShardedPopulation(n_total, hidden, process_group=dist.group.WORLD, device="cuda:0",
                                coins=coin_chain_reference(n_coins), walls=walls)
        g = torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")
        pop.local.set_genomes(g[torch.as_tensor(src[pop.begin:pop.end], device="cuda")])
        records = pop.fly(T)                                              # rollout + all-gather: global order on every rank
        torch.cuda.synchronize()
        np.save(os.path.join(out_dir, f"records_{rank}.npy"), records.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,cfg,world", [
    ("config 4", ((32, 32), 4 * 32768, 4000, 256, 14, 3, None), 4),
    ("config 5", ((64, 64, 64), 2 * 131072, 400, 192, 15, 5,
                  np.array([[12.0, 3.0, 8.0, 13.0, 12.0, 12.0], [-20.0, -20.0, 30.0, 20.0, 20.0, 31.0]])), 2),
])
def test_sharded_population_equals_unsharded(Population, torch, tmp_path, name, cfg, world):
    """ShardedPopulation with `world` ranks (one process each, sharing this GPU; gloo carries the
    all-gather) at the shard sizes of BASELINE configs 4 and 5: every rank's gathered records equal the
    records of the same population flown unsharded, in global drone order."""
    import torch.multiprocessing as mp
    hidden, n_total, T, m, seed, n_coins, walls = cfg
    port = 29600 + (os.getpid() + len(name)) % 300
    mp.spawn(_shard_worker, args=(world, port, cfg, str(tmp_path)), nprocs=world, join=True)
    genes_m, src = _replicated(hidden, m, n_total, seed)
    whole = Population(n_total, hidden, coins=coin_chain_reference(n_coins), walls=walls)
    whole.set_genomes(torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")[torch.as_tensor(src, device="cuda")])
    whole.fly(T)
    want = whole.records.cpu().numpy()
    for r in range(world):
        got = np.load(tmp_path / f"records_{r}.npy")
        assert got.shape == want.shape and np.array_equal(got, want), (name, r)
    res = whole.results()
    assert res["crashed"].any() and not res["crashed"].all() and res["coins"].max() >= 1


# ------------------------------------------------------------------------------------------- placement of the children
def test_child_placement_is_the_same_ga_run(Population, torch):
    """DeviceEvolution places the children of a generation sorted by the slot of their first parent (gather
    locality of the breed kernel) or, optionally, by that parent's flight time.  The GA works on canonical
    indices and every draw is keyed by the child, so the run must be THE SAME run whatever the placement:
    identical best / global-best history and an identical multiset of (reward, flight time, coins) in every
    generation, with the elite in slot 0."""
    from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
    n, hidden, T, gens = 4096, (16, 16), 600, 5
    genes = _hover_genes(hidden, n, sigma=0.02, seed=8)                   # a population on the edge: some fly, many crash
    runs = {}
    for place in (None, "parent", "lifetime"):
        pop = ShardedPopulation(n, hidden)
        pop.local.set_genomes(genes)
        evo = DeviceEvolution(pop, seed=11, place=place)
        per_gen = []
        for g in range(gens):
            best, gbest, rec = evo.step(T)
            rows = rec[:, :3].cpu().numpy()
            per_gen.append(rows[np.lexsort(rows.T[::-1])])
            if g > 0 and evo.elite is not None:
                assert torch.equal(pop.local.genome_of(0), evo.elite)     # the elite sits in slot 0 of the new generation
            if place == "parent":
                first = pop.local.last_parents.long()[pop.local.last_order.long()][1:]
                assert bool((first[1:] >= first[:-1]).all())              # neighbours in the gather
        runs[place] = (list(evo.history), per_gen)
    for place in ("parent", "lifetime"):
        assert runs[None][0] == runs[place][0]
        for a, b in zip(runs[None][1], runs[place][1]):
            assert np.array_equal(a, b)
    assert len({h for h in runs[None][0]}) > 1                            # the run does evolve


@pytest.mark.parametrize("hidden,n,T", [((32, 32), 3000, 1500), ((16, 16), 1030, 2000), ((64, 64, 64), 300, 700), ((10, 10), 9, 300)])
def test_work_queue_rollout_is_bit_identical(Population, torch, hidden, n, T):
    """The resident kernel hands drones to lane groups from a work queue (a group whose drone has crashed
    takes the next one), so groups of one wavefront are at different timesteps of different drones.  Every
    drone still executes its own timesteps with its own genome: final states, integer bookkeeping and
    records equal the step engine bit for bit -- on a population where most drones crash at scattered
    times, a few fly the whole episode, some hit a wall, and there are more drones than the persistent grid
    has groups only at the largest size; an episode continued in two parts equals the whole."""
    from test_gpu_parity import _mixed_genomes
    spec, genes, cls = _mixed_genomes(hidden, n, seed=n)
    rng = np.random.default_rng(1)
    genes[cls == 1] += rng.normal(0, 0.02, genes[cls == 1].shape)          # marginal flyers: crash at scattered times
    genes = genes.astype(np.float32).astype(np.float64)
    walls = np.array([[3.0, 3.0, 8.0, 4.5, 7.0, 12.0]])
    out = []
    for engine in ("step", "resident"):
        pop = Population(n, hidden, walls=walls)
        pop.set_genomes(genes)
        pop.fly(T, engine=engine)
        out.append((pop.fstate.clone(), pop.istate.clone(), pop.records.clone()))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    life = (out[0][1][2] >> 1).cpu().numpy()
    assert (life < T).mean() > 0.3 and (life == T).any()
    import ctypes as C
    from learning_drones_amd import _native
    pop = Population(n, hidden, walls=walls)                               # the same episode in two resident launches
    pop.set_genomes(genes)
    for k in (T // 3, T - T // 3):
        _native.check(_native.lib().ld_fly_steps_resident_f32(C.byref(pop.sim), C.byref(pop.episode), C.byref(pop.spec),
                                                              pop._ptr(pop.coins_dev), pop._ptr(pop.walls_dev), pop._ptr(pop.genomes),
                                                              pop._ptr(pop.fstate), pop._ptr(pop.istate), n, k, pop._ptr(pop.queue),
                                                              pop._stream()), "ld_fly_steps_resident_f32")
    pop.finalize()
    assert torch.equal(pop.records, out[0][2]) and torch.equal(pop.fstate, out[0][0])


# ------------------------------------------------------------------------------------------- float64 resident engine
@pytest.mark.parametrize("hidden,n,T", [((10, 10), 130, 900), ((16, 16), 200, 700), ((32, 32), 70, 500), ((8, 10), 65, 400), ((16,), 40, 300),
                                        ((32,), 40, 300), ((64,), 40, 300), ((8, 8), 40, 300), ((24, 24), 40, 300)])
def test_resident_engine_f64_is_bit_identical(Population, torch, hidden, n, T):
    """ld_fly_rollout_resident_f64 (one launch per episode, the reference's arithmetic with its float32
    rounding points) against one launch per timestep of the float64 step kernels: bit for bit, distances
    included."""
    from test_gpu_parity import _mixed_genomes
    spec, genes, cls = _mixed_genomes(hidden, n, seed=3 * n)
    out = []
    for engine in ("step", "resident"):
        pop = Population(n, hidden, precision="f64", walls=np.array([[3.0, 3.0, 8.0, 4.5, 7.0, 12.0]]))
        pop.set_genomes(genes)
        pop.fly(T, engine=engine)
        out.append((pop.fstate.clone(), pop.istate.clone(), pop.records.clone(), pop.distance.clone()))
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    crashed = (out[0][1][2] & 1).bool()
    assert crashed.any() and not crashed.all()
    with pytest.raises(ValueError):
        Population(4, (64, 64, 64), precision="f64").fly(10, engine="resident")   # does not fit in registers as doubles


def test_closed_loop_f64_resident_tracks_reference(Population, g3):
    """The reference's own closed-loop episodes (golden G3) through the float64 RESIDENT engine -- what the scalar
    Drone / fly() drop-in now runs (engine="auto"), one launch per episode: every integer outcome exact, the final
    distance to 1e-8, for every episode whose spec has a float64 resident kernel."""
    seen = 0
    for ei in range(int(g3["n_episodes"])):
        hidden = tuple(int(h) for h in g3[f"hidden_{ei}"])
        if hidden not in Population.RESIDENT_SPECS_F64:
            continue
        pop = Population(1, hidden, precision="f64")
        pop.set_genomes(g3[f"weights_{ei}"][None])
        pop.fly(int(g3[f"timesteps_{ei}"]), engine="resident")
        r, fin = pop.results()[0], g3[f"final_{ei}"]
        assert (int(r["reward"]), int(r["coins"]), int(r["flight_time"]), bool(r["crashed"])) == \
               (int(fin[0]), int(fin[2]), int(fin[3]), bool(fin[4])), ei
        if ei in CHAOTIC:                                # grazes the cos(pitch) = 0 pole: integers + a loose distance, as in L0 / L3
            assert abs(r["distance_to_coin"] - fin[5]) < 0.05, ei
        else:
            assert abs(r["distance_to_coin"] - fin[5]) <= 1e-8 * max(1.0, abs(fin[5])), ei
            ref = g3[f"state_{ei}"][-1]
            got = pop.state_numpy()[0, :12]
            assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 1e-8, ei
        seen += 1
    assert seen >= 40


# ------------------------------------------------------------------------------------------- run-time specialisation
@pytest.mark.parametrize("hidden", [(12, 12), (20, 20), (9, 13, 11), (16, 32)])
def test_runtime_specialised_spec_equals_generic_kernels(Population, torch, generic_kernels, hidden):
    """Population(..., specialise=True) compiles the unrolled kernels for a spec the library does not ship
    (hipcc at first use) and registers them; every launch geometry of the specialised kernels, and
    feed_forward, must reproduce the run-time-spec kernels bit for bit."""
    from test_gpu_parity import _mixed_genomes
    n, T = 200, 300
    spec, genes, cls = _mixed_genomes(hidden, n, seed=17)
    x = np.random.default_rng(2).normal(0, 3, (15, n))
    generic_kernels(True)
    ref = Population(n, hidden)
    ref.set_genomes(genes)
    y_ref = ref.feed_forward(x).clone()
    for _ in range(T):
        ref.step()
    ref.finalize()
    generic_kernels(False)
    pop = Population(n, hidden, specialise=True)
    assert pop.rollout_geometry(T)["steps_per_launch"] in (64, 512)      # 512 where the spec keeps its genes on chip (up to 512 gene words)
    pop.set_genomes(genes)
    assert torch.equal(pop.feed_forward(x), y_ref)
    pop.fly(T)
    assert torch.equal(pop.fstate, ref.fstate) and torch.equal(pop.istate, ref.istate) and torch.equal(pop.records, ref.records)
    for waves, k in ((1, 1), (2, 7), (8, 300), (16, 64), (11, 64), (32, 9)):   # (8 and up with >= 8 timesteps: the gene-keeping kernels;
        pop.reset()                                                               #  16 of (16, 32): the halves of a wavefront paired)
        pop.fly(T, steps_per_launch=k, waves_per_block=waves)
        assert torch.equal(pop.records, ref.records), (waves, k)
    assert pop.has_resident_engine()                                     # the specialisation brings the resident engine too
    pop.reset()
    pop.fly(T, engine="resident")
    assert torch.equal(pop.records, ref.records) and torch.equal(pop.fstate, ref.fstate)
    crashed = (ref.istate[2] & 1).bool()
    assert crashed.any() and not crashed.all()


def test_large_float32_population_specialises_by_itself(Population, torch, monkeypatch):
    """specialise=None (the default): a float32 population of AUTO_SPECIALISE_MIN_DRONES or more compiles and
    registers the unrolled kernels of a spec the library does not ship; a smaller one, a float64 one, and any
    with LD_SPECIALISE=0 do not; a failing compile leaves the run-time-spec kernels in place with a warning."""
    from learning_drones_amd import _native, build as ld_build
    from learning_drones_amd.params import MlpSpec
    import ctypes as C
    hidden = (11, 7)
    has = lambda: _native.lib().ld_has_specialisation_f32(C.byref(MlpSpec.from_hidden(hidden))) == 1
    assert not has()
    Population(Population.AUTO_SPECIALISE_MIN_DRONES, hidden)             # conftest: LD_SPECIALISE=0
    assert not has()
    monkeypatch.setenv("LD_SPECIALISE", "1")
    Population(64, hidden)
    Population(Population.AUTO_SPECIALISE_MIN_DRONES, hidden, precision="f64")
    assert not has()
    with monkeypatch.context() as m:                                      # hipcc "fails": warning, generic kernels
        m.setattr(ld_build, "build_specialisation", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no compiler")))
        with pytest.warns(RuntimeWarning, match="no run-time specialisation"):
            pop = Population(Population.AUTO_SPECIALISE_MIN_DRONES, hidden)
        assert not has() and pop.rollout_geometry(100)["waves_per_block"] == 1
        with pytest.raises(RuntimeError):
            Population(Population.AUTO_SPECIALISE_MIN_DRONES, hidden, specialise=True)
    pop = Population(Population.AUTO_SPECIALISE_MIN_DRONES, hidden)
    assert has() and pop.rollout_geometry(100)["waves_per_block"] > 1 and pop.has_resident_engine()


def test_short_soak_of_engines_and_geometries():
    """scripts/soak_engines.py (random sizes, layer specs, episode lengths, coin tables, walls, wind; every engine and a
    random launch geometry against the run-time-spec kernel) for a dozen rounds: no mismatch.  The long runs (120
    rounds) are done by hand; this keeps the script and the property alive in the suite."""
    import subprocess, sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(repo / "scripts" / "soak_engines.py"), "12"], cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches: 0" in out.stdout and out.stdout.count(" ok") >= 12, out.stdout[-2000:]


@pytest.mark.parametrize("precision,hidden,force_generic", [("f64", (32, 32), False), ("f64", (10, 10), True), ("f64", (9, 13, 11), False),
                                                            ("f32", (13, 9), False), ("f32", (32, 32), True)])
def test_one_wavefront_kernels_fly_several_timesteps_per_launch(Population, torch, generic_kernels, precision, hidden, force_generic):
    """ld_fly_steps_f64 / ld_fly_steps_f32 on a run-time layer spec: the one-wavefront-per-block kernel (float64
    unrolled and run-time-spec, float32 run-time-spec) with the timestep loop inside the launch equals one launch
    per timestep bit for bit, and fly() (64 timesteps per launch) equals that plus finalize."""
    from test_gpu_parity import _mixed_genomes
    n, T = 150, 330
    spec, genes, cls = _mixed_genomes(hidden, n, seed=23)
    walls = np.array([[3.0, 3.0, 8.0, 4.5, 7.0, 12.0]])
    generic_kernels(force_generic)

    def fresh():
        pop = Population(n, hidden, precision=precision, walls=walls, coins=coin_chain_reference(2), specialise=False)
        pop.set_genomes(genes)
        return pop

    ref = fresh()
    for _ in range(T):
        ref.step()
    crashed = (ref.istate[2] & 1).bool()
    assert crashed.any() and not crashed.all()
    for k in (1, 7, 64, T):
        pop = fresh()
        done = 0
        while done < T:
            pop.steps(min(k, T - done))
            done += min(k, T - done)
        assert torch.equal(pop.fstate, ref.fstate) and torch.equal(pop.istate, ref.istate), k
    geo = fresh().rollout_geometry(T)
    assert (geo["steps_per_launch"], geo["waves_per_block"], geo["launches"]) == (64, 1, 6)
    ref.finalize()
    pop = fresh()
    pop.fly(T)
    assert torch.equal(pop.records, ref.records) and torch.equal(pop.fstate, ref.fstate)
    assert precision == "f32" or torch.equal(pop.distance, ref.distance)
    if precision == "f32":
        pop = fresh()
        pop.fly(T, steps_per_launch=50)                                # explicit geometry on a run-time layer spec
        assert torch.equal(pop.records, ref.records)
        with pytest.raises(_native_error()):
            fresh().steps(3, 4)                                        # one wavefront per block is its only geometry
    else:
        with pytest.raises(ValueError):
            fresh().steps(3, 4)


def _native_error():
    from learning_drones_amd._native import LdError
    return LdError


def test_bench_prints_one_json_line_with_the_contract_fields():
    """bench.py on a small workload: exactly one JSON line on stdout carrying the driver's contract fields, the
    roofline object (live launch duration from HIP events) and, at N = 1, the cpu_baseline object."""
    import json, subprocess, sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--drones", "4096",
                          "--timesteps", "200"], cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 200 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    # every level the launch's bytes can come from is priced; `bound` names the one with the highest fraction, which is
    # therefore the line's frac -- a fraction of the ceiling that binds, never above 1
    assert {"hbm", "l2", "lds"} <= set(r["levels"]) and all(abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-9 for v in r["levels"].values())
    top = max(r["levels"].values(), key=lambda v: v["frac"])
    assert r["frac"] == top["frac"] and r["achieved"] == top["achieved"] and r["peak"] == top["peak"] and 0 < r["frac"] <= 1.0
    assert r["levels"]["hbm"]["peak"] == 8000.0 and r["unit"] in ("GB/s", "G wave-instructions/s")
    assert r["launch_us"] > 0 and "traffic" in r and r["lds_words_per_drone"] == 124 and r["register_words_per_drone"] == 304 and r["streamed_words_per_drone"] == 0   # workgroups of 64 drones, fly_wide_kernel
    assert "roofline_hbm_bound" in d and d["one_timestep_per_launch"]["bound"] == "hbm"
    assert "exchange_bytes_per_generation_per_rank" in d["generation"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["one_thread"]["cores"] == 1
    assert d["ranks_seen"] == [0] and d["records"]["rows"] == 4096


def test_bench_two_ranks_on_one_gpu_rehearses_the_multi_gpu_flow():
    """bench.py --gpus 2 launched the way the driver launches it, two ranks sharing this GPU with gloo carrying the
    collectives (RCCL refuses two ranks on one device): the N > 1 control flow -- sharding, the records all-gather in
    every episode, max-over-ranks timing, the strong-scaling case, the generation step with its parent exchange --
    runs to the one JSON line, which proves both ranks were seen."""
    import json, socket, subprocess, sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--drones", "8192",
           "--timesteps", "150", "--backend", "gloo"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == [0, 1] and d["records"]["rows"] == 2 * 8192 and d["scaling"] == "weak"
    s = d["strong_scaling"]
    assert s["scaling"] == "strong" and s["population"] == 8192 and s["drones_per_gpu"] == 4096 and s["records"]["rows"] == 8192
    assert d["generation"]["error"] is None and d["generations_per_s"] > 0
    assert "cpu_baseline" not in d                                        # rank 0 at N = 1 only


def test_kept_genes_are_bit_identical_on_a_ragged_population(Population, torch, lib):
    """Launches of 8 timesteps or more keep 136 KiB of every workgroup's genes in LDS (fly_split_kernel<..., 136, ...>): a
    ragged population (not a multiple of 64, of 48 or of the tile), walls, coins collected, crashes -- state, bookkeeping
    and records equal the same episode flown without the kept genes (ld_debug_lds_keep(0): Infinity-Cache tiles), in every
    geometry that keeps (workgroups of 64, 48, 32, 16 drones), run after run, also from a non-default stream of the
    caller's, and work enqueued behind fly() sees the finished episode."""
    hidden, m, T = (32, 32), 256, 330
    n = 2 * 16384 + 16384 + 192 + 37
    genes_m, src = _replicated(hidden, m, n, seed=11)
    walls = np.array([[3.0, 3.0, 8.0, 4.5, 7.0, 12.0]])
    pop = Population(n, hidden, walls=walls, coins=coin_chain_reference(2))
    pop.set_genomes(torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")[torch.as_tensor(src, device="cuda")])
    geo = pop.rollout_geometry(T)
    assert geo["tile_drones"] == n and geo["waves_per_block"] == 8 and geo["lds_words"] + geo["register_words"] == 428 and geo["streamed_words"] == 0   # fly_wide_kernel
    assert lib.ld_debug_lds_keep(0) == -1
    try:
        assert pop.rollout_geometry(T)["tile_drones"] == 16384 and pop.rollout_geometry(T)["lds_words"] == 0
        pop.fly(T)                                                        # nothing kept: tiles, four wavefronts per block
    finally:
        assert lib.ld_debug_lds_keep(-1) == 0
    want = (pop.fstate.clone(), pop.istate.clone(), pop.records.clone())
    crashed = (want[1][2] & 1).bool()
    assert crashed.any() and not crashed.all() and int(want[1][1].max()) >= 1
    for rep, kw in enumerate(({}, {}, dict(waves_per_block=16), dict(waves_per_block=11), dict(waves_per_block=32),
                              dict(waves_per_block=11, tile_drones=12288), dict(steps_per_launch=9), dict(steps_per_launch=7),
                              dict(waves_per_block=16, steps_per_launch=9), dict(unpaired=True, waves_per_block=16),
                              dict(unpaired=True, waves_per_block=8))):
        pop.reset()
        if kw.pop("unpaired", False):                                     # the same workgroups of 32 drones without the paired halves
            assert lib.ld_debug_lds_keep(2) == -1
        pop.fly(T, **kw)
        lib.ld_debug_lds_keep(-1)
        summed = pop.records.sum(dim=0)                                   # enqueued behind fly() on the same stream
        assert torch.equal(pop.fstate, want[0]) and torch.equal(pop.istate, want[1]) and torch.equal(pop.records, want[2]), (rep, kw)
        assert torch.equal(summed, want[2].sum(dim=0))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                                         # the caller's own stream
        pop.reset()
        pop.fly(T)
    side.synchronize()
    assert torch.equal(pop.fstate, want[0]) and torch.equal(pop.records, want[2])


def test_two_populations_from_two_threads_and_two_streams(Population, torch):
    """SURVEY 8b "Ownership / Threading": the library keeps no per-population, per-stream or per-thread launch state (it
    owns no stream and allocates nothing), so two populations flown at once -- from one host thread on two streams, and
    from two host threads -- give exactly what each gives alone."""
    import threading
    from bench import make_genomes
    T = 300
    pops, want = [], []
    for k, (n, hidden) in enumerate(((16384 + 64, (32, 32)), (8192, (16, 16)))):
        p = Population(n, hidden, coins=coin_chain_reference(3))
        p.set_genomes(make_genomes(torch, n, hidden, p.device, k))
        p.fly(T)
        want.append((p.fstate.clone(), p.records.clone()))
        pops.append(p)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for p, st in zip(pops, streams):                                      # one thread, two streams, nothing waits in between
        with torch.cuda.stream(st):
            p.reset()
            p.fly(T)
    torch.cuda.synchronize()
    for p, w in zip(pops, want):
        assert torch.equal(p.fstate, w[0]) and torch.equal(p.records, w[1])
    errors = []

    def worker(p, st):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(st):
                for _ in range(3):
                    p.reset()
                    p.fly(T)
            st.synchronize()
        except Exception as e:                                            # surfaced below
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(p, st)) for p, st in zip(pops, streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for p, w in zip(pops, want):
        assert torch.equal(p.fstate, w[0]) and torch.equal(p.records, w[1])


def test_rollout_is_capturable_in_a_hip_graph(Population, torch):
    """ld_fly_rollout_f32 enqueues everything from C on the caller's stream (the library owns no stream), so a whole
    episode can be captured into a HIP graph and replayed: same records."""
    from bench import make_genomes
    n, T = 16384, 200
    pop = Population(n, (32, 32), coins=coin_chain_reference(3))
    pop.set_genomes(make_genomes(torch, n, (32, 32), pop.device, 0))
    assert pop.rollout_geometry(T)["lds_words"] > 0                       # the captured launches keep genes in LDS
    pop.fly(T)
    want = pop.records.clone()
    pop.reset()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pop.fly(T)
    for _ in range(2):
        pop.records.zero_()
        pop.reset()
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(pop.records, want)


# ------------------------------------------------------------------------------------------- both frames
def test_accelerations_in_both_frames_match_the_reference(torch, lib, g8):
    """calculate_accelerations(lin_acc_drone_2_lab=False) and the self.Rot it leaves behind (physical_model.py:90,
    :114-120; golden G8, with wind and with the payload parameter set): the float64 kernel reproduces the reference's
    body-frame pair and rotation matrix to rounding, the float32 closed form to 1e-3 of the scale; the lab-frame pair
    of the same entry point equals ld_accelerations_* bit for bit; and the QuadcopterPhysics shim returns what the
    reference's method returns, shapes and self.Rot included."""
    import ctypes as C
    from learning_drones_amd.drone.quadcopter import QuadcopterPhysics
    from test_oracle_golden import frames_sim
    for tag in ("plain", "payload"):
        ang, thr, wind = g8[f"{tag}_angle"], g8[f"{tag}_thrust"], g8[f"{tag}_wind"]
        sim = frames_sim(g8, tag)
        for w in np.unique(wind, axis=0)[:40]:                        # the wind is uniform per launch
            rows = np.where((wind == w).all(axis=1))[0]
            sim.wind[:] = list(w)
            n = len(rows)
            for sfx, dtype, tol in (("f64", torch.float64, 1e-12), ("f32", torch.float32, 1e-3)):
                a = torch.as_tensor(ang[rows].T.copy(), dtype=dtype, device="cuda").contiguous()
                u = torch.as_tensor(thr[rows].T.copy(), dtype=dtype, device="cuda").contiguous()
                out = {}
                for lab in (0, 1):
                    lin, rot = torch.empty((3, n), dtype=dtype, device="cuda"), torch.empty((3, n), dtype=dtype, device="cuda")
                    mat = torch.empty((9, n), dtype=dtype, device="cuda")
                    rc = getattr(lib, f"ld_accelerations_frames_{sfx}")(C.byref(sim), a.data_ptr(), u.data_ptr(), lin.data_ptr(),
                                                                        rot.data_ptr(), mat.data_ptr(), lab, n, None)
                    assert rc == 0
                    out[lab] = (lin.double().cpu().numpy().T, rot.double().cpu().numpy().T, mat.double().cpu().numpy().T.reshape(n, 3, 3))
                scale = lambda ref: np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
                for lab, key in ((0, "body"), (1, "lab")):
                    ok = np.abs(np.cos(ang[rows][:, 1])) > 0.05 if (lab and sfx == "f32") else np.ones(n, bool)   # away from the tan(pitch) pole
                    for got, ref in ((out[lab][0], g8[f"{tag}_{key}_lin"][rows]), (out[lab][1], g8[f"{tag}_{key}_rot"][rows])):
                        assert np.max(np.abs(got - ref)[ok] / scale(ref)[ok]) < tol, (tag, sfx, key)
                    assert np.max(np.abs(out[lab][2] - g8[f"{tag}_Rot"][rows])) < (1e-15 if sfx == "f64" else 1e-6)
                if sfx == "f64" and n > 50:                            # the reference keeps forces / m in float32: the same float32 values
                    assert np.mean(out[0][0] == g8[f"{tag}_body_lin"][rows]) > 0.99   # (the device's sin / cos differ from libm's in the last bit now and then)
                lin2, rot2 = torch.empty((3, n), dtype=dtype, device="cuda"), torch.empty((3, n), dtype=dtype, device="cuda")
                assert getattr(lib, f"ld_accelerations_{sfx}")(C.byref(sim), a.data_ptr(), u.data_ptr(), lin2.data_ptr(), rot2.data_ptr(), n, None) == 0
                assert np.array_equal(lin2.double().cpu().numpy().T, out[1][0]) and np.array_equal(rot2.double().cpu().numpy().T, out[1][1])
    qp = QuadcopterPhysics()
    assert qp.Rot.shape == (3, 3)
    for i in (0, 120, 250):
        col = lambda v: np.asarray(v, dtype=np.float64).reshape(-1, 1)
        lin, rot = qp.calculate_accelerations(col(g8["plain_angle"][i]), col(g8["plain_wind"][i]), col(g8["plain_thrust"][i]), False)
        assert lin.shape == (3, 1) and rot.shape == (3, 1) and lin.dtype == np.float32
        assert np.allclose(lin.ravel().astype(np.float64), g8["plain_body_lin"][i], rtol=2e-7) and np.allclose(rot.ravel(), g8["plain_body_rot"][i], rtol=1e-13)
        assert np.allclose(qp.Rot, g8["plain_Rot"][i], rtol=0, atol=1e-15)
        lin, rot = qp.calculate_accelerations(col(g8["plain_angle"][i]), col(g8["plain_wind"][i]), col(g8["plain_thrust"][i]), True)
        assert np.allclose(lin.ravel(), g8["plain_lab_lin"][i], rtol=1e-12, atol=1e-12) and np.allclose(rot.ravel(), g8["plain_lab_rot"][i], rtol=1e-12, atol=1e-12)

