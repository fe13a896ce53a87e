"""GPU tests added in round 3: the 8-rank shapes of BASELINE configs 4 and 5 (eight shards assembled == unsharded), the
generation step over eight ranks with both parent exchanges, and the scalar drop-in surface kept resident."""
import numpy as np
import pytest

from conftest import run_ranks
from learning_drones_amd.params import coin_chain_reference
from test_gpu_round2 import _hover_genes, _replicated

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.mark.parametrize("name,cfg", [
    ("config 4", ((32, 32), 8 * 32768, 4000, 256, 14, 3, None)),
    ("config 5", ((64, 64, 64), 8 * 131072, 4000, 192, 15, 5,
                  np.array([[12.0, 3.0, 8.0, 13.0, 12.0, 12.0], [-20.0, -20.0, 30.0, 20.0, 20.0, 31.0]]))),
])
def test_eight_shards_assemble_to_the_unsharded_population(torch, name, cfg):
    """BASELINE configs 4 and 5 at their FULL sizes (262 144 x 4 000, 32-32, 3 coins; 1 048 576 x 4 000, 64-64-64, 5 coins
    and walls), sharded over 8 ranks exactly as parallel.ShardedPopulation shards them over 8 GPUs -- contiguous ranges,
    a rollout per shard with no communication, one all-gather of the 16-byte records -- on the one GPU of this box:
    eight threads of one process, a mailbox standing in for RCCL.  Every rank's gathered records equal the records of
    the same population flown unsharded, in global drone order."""
    from learning_drones_amd.parallel import ShardedPopulation
    from learning_drones_amd.population import Population
    hidden, n_total, T, m, seed, n_coins, walls = cfg
    genes_m, src = _replicated(hidden, m, n_total, seed)
    g = torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")
    src_dev = torch.as_tensor(src, device="cuda")

    def rank_body(comm):
        pop = ShardedPopulation(n_total, hidden, process_group=comm, device="cuda:0", coins=coin_chain_reference(n_coins), walls=walls)
        assert (pop.rank, pop.world) == (comm.rank, 8) and pop.local.n == n_total // 8
        pop.local.set_genomes(g[src_dev[pop.begin:pop.end]])
        records = pop.fly(T)                                              # rollout + all-gather: global order on every rank
        torch.cuda.synchronize()
        return records.cpu().numpy()

    gathered = run_ranks(8, rank_body)
    whole = Population(n_total, hidden, coins=coin_chain_reference(n_coins), walls=walls)
    whole.set_genomes(g[src_dev])
    whole.fly(T)
    want = whole.records.cpu().numpy()
    for r in range(8):
        assert gathered[r].shape == want.shape and np.array_equal(gathered[r], want), (name, r)
    res = whole.results()
    assert res["crashed"].any() and not res["crashed"].all() and res["coins"].max() >= 1


@pytest.mark.parametrize("exchange,place", [("allgather", None), ("alltoall", None), ("alltoall", "parent"), ("allgather", "parent")])
def test_generation_step_over_eight_ranks_is_the_single_rank_run(torch, exchange, place):
    """parallel.DeviceEvolution over 8 ranks (threads + mailbox, as above): rollout, all-gather of the records, elite
    broadcast from its owner, parent exchange (all-gather of the shards, or the all-to-all of exactly the selected
    parents -- with place="parent" also through the canonical-index -> slot translation), device select + breed.  The
    GA is keyed by canonical child indices, so the run must be the 1-rank run: same best / global-best history, and after
    four generations the same genomes (as a set of rows when the children were placed out of order)."""
    from learning_drones_amd.parallel import DeviceEvolution, ShardedPopulation
    n, hidden, T, gens = 8 * 512, (16, 16), 500, 4
    genes = _hover_genes(hidden, n, sigma=0.02, seed=21)

    def run(comm):
        pop = ShardedPopulation(n, hidden, process_group=comm, device="cuda:0")
        pop.set_genomes(genes)
        evo = DeviceEvolution(pop, seed=5, exchange=exchange, place=place)
        history = [evo.step(T)[:2] for _ in range(gens)]
        assert evo.exchange == exchange                                  # (nothing fell back)
        torch.cuda.synchronize()
        return history, pop.local.get_genomes().cpu().numpy(), evo.exchange_bytes()

    want_history, want_genes, _ = run(None)
    ranks = run_ranks(8, run)
    for r, (history, _, moved) in enumerate(ranks):
        assert history == want_history, r
        assert moved > 0
    got = np.concatenate([g for _, g, _ in ranks])
    order = lambda a: a[np.lexsort(a.T[::-1])]
    if place is None:
        assert np.array_equal(got, want_genes)
    else:
        assert np.array_equal(got[0], want_genes[0]) and np.array_equal(order(got), order(want_genes))   # the elite keeps slot 0
    assert want_history[-1][1] >= want_history[0][1]


def test_compacting_rollout_flies_lists_of_the_drones_still_flying(torch):
    """ld_fly_rollout_compacting_f32 (engine="compacting"): before every launch a kernel lists the drones whose status word
    says "not crashed" and fly_wide_kernel flies the list, 64 listed drones per workgroup, gathering their genes.  A mixed
    population (crashes scattered through the episode, coins collected, a wall, a size that is not a multiple of 64): state,
    bookkeeping and records equal the ordinary rollout's bit for bit, at several launch lengths, run after run; a layer spec
    without a fly_wide_kernel takes the ordinary path and says the same."""
    from learning_drones_amd.population import Population
    hidden, m, T = (32, 32), 256, 700
    n = 3 * 4096 + 64 + 29
    genes_m, src = _replicated(hidden, m, n, seed=21)
    walls = np.array([[3.0, 3.0, 8.0, 4.5, 7.0, 12.0]])
    pop = Population(n, hidden, walls=walls, coins=coin_chain_reference(2))
    pop.set_genomes(torch.as_tensor(genes_m, dtype=torch.float32, device="cuda")[torch.as_tensor(src, device="cuda")])
    pop.fly(T)
    want = (pop.fstate.clone(), pop.istate.clone(), pop.records.clone())
    crashed = (want[1][2] & 1).bool()
    assert crashed.any() and not crashed.all() and int(want[1][1].max()) >= 1
    for kw in ({}, dict(steps_per_launch=64), dict(steps_per_launch=9), dict(steps_per_launch=T), {}):
        pop.reset()
        pop.fly(T, engine="compacting", **kw)
        assert torch.equal(pop.fstate, want[0]) and torch.equal(pop.istate, want[1]) and torch.equal(pop.records, want[2]), kw
    # an episode continued: the list is made from the status words the first half left
    pop.reset()
    pop.steps(300)
    pop.fly(T - 300, engine="compacting")
    assert torch.equal(pop.fstate, want[0]) and torch.equal(pop.records, want[2])
    wide = Population(2048, (64, 64, 64), coins=coin_chain_reference(3))
    wide.set_genomes(torch.as_tensor(_hover_genes((64, 64, 64), 2048, seed=3), dtype=torch.float32, device="cuda"))
    wide.fly(150)
    rec = wide.records.clone()
    wide.reset()
    wide.fly(150, engine="compacting")
    assert torch.equal(wide.records, rec)
