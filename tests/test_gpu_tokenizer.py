"""Parity proper: HIP tokenizer / BFS vs the reference's golden vectors and vs the C oracle, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps
from oracle import oracle as orc
from tests.helpers import GOLDEN, load_tok, load_tokp, sha_rows, tok_cases, tokp_cases

pytestmark = pytest.mark.gpu


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def replay_gpu(case, n_inst=1, check_dist=False):
    from mapf_gpt_amd.observation_generator import BatchedTokenizer, InputParameters
    grid, P, G, A = case["grid"], case["pos"], case["goal"], case["actions"]
    S, n = P.shape[0], P.shape[1]
    cfg = InputParameters(grid_step=int(case["grid_step"]) if "grid_step" in case else 64)
    if "params" in case:
        cfg = InputParameters(*[int(v) for v in case["params"]], 64, False)
    tok = BatchedTokenizer(grid, n_inst, n, cfg)
    out = []
    for t in range(S):
        pos = _dev(np.broadcast_to(P[t], (n_inst, n, 2)), torch.int16)
        goal = _dev(np.broadcast_to(G[t], (n_inst, n, 2)), torch.int16)
        act = _dev(np.broadcast_to(A[t].astype(np.int32), (n_inst, n)), torch.int32)
        if t == 0:
            tok.create_agents(pos, goal)
            if check_dist:
                d = tok.distance_fields()
                for a in range(0, n, max(1, n // 8)):
                    assert np.array_equal(d[0, a], orc.bfs(grid, G[0][a])), f"BFS field of agent {a} differs"
        tok.update_agents(pos, goal, act, goals_may_change=True)
        out.append(tok.generate_observations().cpu().numpy().reshape(n_inst, n, 256))
    return np.array(out)       # [S, n_inst, n, 256]


def test_known_answer_cpp_main():
    """observation_generator.cpp:530-544 on the device (unpadded 256x256 grid: BFS runs in global-memory mode)."""
    from mapf_gpt_amd.observation_generator import InputParameters, ObservationGenerator
    g = np.load(os.path.join(GOLDEN, "tok_known_answer.npz"))
    gen = ObservationGenerator([[0] * 256 for _ in range(256)], InputParameters(20, 13, 5, 256, 5, 5, 64, False))
    gen.create_agents([(120, 120)], [(20, 200)])
    gen.update_agents([(120, 120)], [(20, 200)], [0])
    row = np.array(gen.generate_observations(), dtype=np.uint8)
    assert np.array_equal(row, g["tokens"])
    assert sha_rows(row[0]) == "896eb85aa89a369759917e5903f237dc28387e6b7d431fbd6703f302a97585e1"


@pytest.mark.parametrize("name", tok_cases())
def test_golden_trajectories_bit_exact(name):
    case = load_tok(name)
    got = replay_gpu(case, n_inst=1, check_dist=True)[:, 0]
    assert sha_rows(got) == str(case["sha256_all_rows"])
    assert np.array_equal(got[:, case["keep"]], case["tokens"])


@pytest.mark.parametrize("name", tokp_cases())
@pytest.mark.parametrize("n_inst", [1, 3])
def test_non_default_input_parameters_vs_reference_goldens(name, n_inst):
    """struct InputParameters (observation_generator.h:22-40) beyond what inference.py passes: other value limits (vocabulary), record
    slots, history lengths (7-, 9- and 5-token records: odd record alignment), obs radii (49- and 25-cell windows: lanes without a
    window cell) and agents radii, bit-exact against rows written by the compiled reference (tests/golden/make_golden_params.py)."""
    case = load_tokp(name)
    got = replay_gpu(case, n_inst=n_inst)
    for i in range(n_inst):
        assert np.array_equal(got[:, i][:, case["keep"]], case["tokens"]), f"instance slot {i}"
    assert sha_rows(got[:, 0]) == str(case["sha256_all_rows"])


def test_batched_instances_share_one_map():
    """Several instances on one shared map: every instance slot must reproduce the golden rows."""
    case = load_tok("mazes000")
    got = replay_gpu(case, n_inst=5)
    for i in range(5):
        assert np.array_equal(got[:, i], case["tokens"]), f"instance slot {i}"


@pytest.mark.parametrize("n_inst,n_agents,h,w", [(7, 24, 20, 21), (3, 130, 40, 44), (2, 1, 8, 8), (4, 70, 90, 30),
                                                  (2, 300, 48, 50), (1, 700, 64, 60), (1, 1100, 80, 72)])
def test_many_maps_vs_oracle(n_inst, n_agents, h, w):
    """Distinct map per instance (n_grids == n_inst), ragged agent counts (1, 70, 130: not multiples of 16/64; 300, 700,
    1100: 8, 16 and 32 candidate passes per row), goal changes mid-way; checker = C oracle."""
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    rng = np.random.Generator(np.random.PCG64(n_inst * 1000 + n_agents))
    grids = np.stack([maps.pad(maps.random_map(h, w, 0.12 + 0.03 * i, 50 + i)) for i in range(n_inst)])
    pg = [maps.place_agents(grids[i], n_agents, i) for i in range(n_inst)]
    pos = np.stack([p for p, _ in pg]).astype(np.int32)
    goal = np.stack([g for _, g in pg]).astype(np.int32)
    gens = [orc.OracleGenerator(grids[i]) for i in range(n_inst)]
    tok = BatchedTokenizer(grids, n_inst, n_agents)
    last = np.full((n_inst, n_agents), -1, np.int32)
    for t in range(8):
        if t == 4:      # change a third of the goals
            for i in range(n_inst):
                free = np.argwhere(maps.largest_component(grids[i] == 0))
                who = rng.permutation(n_agents)[: max(1, n_agents // 3)]
                goal[i, who] = free[rng.permutation(len(free))[: len(who)]]
        dp, dg, da = _dev(pos, torch.int16), _dev(goal, torch.int16), _dev(last, torch.int32)
        if t == 0:
            tok.create_agents(dp, dg)
            for i in range(n_inst):
                gens[i].create_agents(pos[i], goal[i])
        tok.update_agents(dp, dg, da, goals_may_change=True)
        got = tok.generate_observations().cpu().numpy().reshape(n_inst, n_agents, 256)
        for i in range(n_inst):
            gens[i].update_agents(pos[i], goal[i], last[i])
            assert np.array_equal(got[i], gens[i].generate_observations()), f"step {t} instance {i}"
        last = rng.integers(-1, 6, (n_inst, n_agents)).astype(np.int32)     # includes out-of-range ids -> "n"
        for i in range(n_inst):
            pos[i], _ = orc.env_step(grids[i], pos[i], goal[i], np.clip(last[i], 0, 4))


def test_static_goal_fast_path_equals_checked_path():
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    case = load_tok("random000")
    grid, P, G, A = case["grid"], case["pos"], case["goal"], case["actions"]
    n = P.shape[1]
    tok = BatchedTokenizer(grid, 1, n)
    for t in range(6):
        pos, goal, act = _dev(P[t][None], torch.int16), _dev(G[t][None], torch.int16), _dev(A[t][None].astype(np.int32), torch.int32)
        if t == 0:
            tok.create_agents(pos, goal)
        tok.update_agents(pos, goal, act, goals_may_change=False)
        assert np.array_equal(tok.generate_observations().cpu().numpy(), case["tokens"][t])


def test_full_size_properties_cfg2():
    """BASELINE cfg 2 size (256 instances x 64 agents): size-independent properties -- every token < 67, window
    centre token is 20 (distance to self = 0), slot 0 is the agent itself (rel pos 20,20), pads are 66,
    instances with identical inputs give identical rows, a second call is idempotent."""
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    from mapf_gpt_amd.runner import make_instances
    grid, s_ok, g_ok = maps.load_named("validation-mazes-seed-000")
    n_inst, n = 256, 64
    pos, goal = make_instances(grid, n_inst, n, 0, s_ok, g_ok)
    pos[200:] = pos[:56]
    goal[200:] = goal[:56]
    tok = BatchedTokenizer(grid, n_inst, n)
    dp, dg = pos.cuda(), goal.cuda()
    tok.create_agents(dp, dg)
    tok.update_agents(dp, dg, torch.full((n_inst, n), -1, dtype=torch.int32).cuda(), goals_may_change=False)
    t1 = tok.generate_observations().cpu().numpy().reshape(n_inst, n, 256)
    t2 = tok.generate_observations().cpu().numpy().reshape(n_inst, n, 256)
    assert np.array_equal(t1, t2)
    assert t1.max() <= 66
    assert (t1[:, :, 60] == 20).all()
    assert (t1[:, :, 121] == 20).all() and (t1[:, :, 122] == 20).all()
    assert (t1[:, :, 125:130] == 44).all()                     # own history is "n" x5 on the first step
    assert (t1[:, :, 251:] == 66).all()
    assert np.array_equal(t1[200:], t1[:56])
    for i in (0, 101, 255):                                    # spot-check 3 instances against the oracle
        # a fresh generator per instance: like the reference's (cpp:391-410), create_agents leaves the previous
        # agents' occupancy marks behind, so a generator is never re-used across agent sets (inference.py:133-139)
        o = orc.OracleGenerator(grid)
        o.create_agents(pos[i].numpy(), goal[i].numpy())
        o.update_agents(pos[i].numpy(), goal[i].numpy(), np.full(n, -1, np.int32))
        assert np.array_equal(o.generate_observations(), t1[i])


def _compare_with_oracle(grid, pos, goal, steps=3, seed=0, params=None, grid_step=64):
    """params = (limit, num_agents, previous actions, obs radius, agents radius) of InputParameters; None: the defaults"""
    from mapf_gpt_amd.observation_generator import BatchedTokenizer, InputParameters
    rng = np.random.Generator(np.random.PCG64(seed))
    n = pos.shape[0]
    gen = orc.OracleGenerator(grid, grid_step=grid_step, params=params)
    cfg = InputParameters(grid_step=grid_step) if params is None else InputParameters(params[0], params[1], params[2], 256, params[3], params[4], grid_step, False)
    tok = BatchedTokenizer(grid, 1, n, cfg)
    last = np.full((n,), -1, np.int32)
    pos = pos.copy()
    for t in range(steps):
        dp, dg, da = _dev(pos[None], torch.int16), _dev(goal[None], torch.int16), _dev(last[None], torch.int32)
        if t == 0:
            tok.create_agents(dp, dg)
            gen.create_agents(pos, goal)
        tok.update_agents(dp, dg, da, goals_may_change=False)
        gen.update_agents(pos, goal, last)
        got = tok.generate_observations().cpu().numpy().reshape(n, 256)
        assert np.array_equal(got, gen.generate_observations()), f"step {t}"
        last = rng.integers(0, 5, (n,)).astype(np.int32)
        pos, _ = orc.env_step(grid, pos, goal, last)
    return tok


@pytest.mark.parametrize("n_agents", [150, 190])
def test_crowded_window_more_than_64_neighbours(n_agents):
    """An open 14x14 room with 150/190 agents: windows near the middle hold > 64 agents, which takes the exact
    slow path of the multi-pass (n_agents > 64) kernel; sparse rows of the same launch take the compacted path."""
    grid = maps.pad(np.zeros((14, 14), np.uint8))
    free = np.argwhere(grid == 0)
    rng = np.random.Generator(np.random.PCG64(n_agents))
    pos = free[rng.permutation(len(free))[:n_agents]].astype(np.int32)
    goal = free[rng.permutation(len(free))[:n_agents]].astype(np.int32)
    _compare_with_oracle(grid, pos, goal, steps=3, seed=n_agents)


@pytest.mark.parametrize("params", [(20, 16, 2, 3, 5), (12, 13, 5, 5, 4), (7, 6, 0, 1, 5), (40, 9, 4, 4, 1)])
@pytest.mark.parametrize("case", ["crowded", "corridor16", "passes8", "sparse"])
def test_input_parameters_on_the_hard_cases_vs_oracle(case, params):
    """InputParameters beyond the defaults on the paths the goldens do not reach, against the C oracle (pinned for these parameter
    sets by tests/golden/tokp_*.npz): windows holding more than 64 agents (exact slow path; up to 16 record slots), distance fields
    that need 16 bits (another sentinel in the same packed arithmetic), eight candidate passes (compaction area of 64 ids with
    the overflow test), a sparse map (rows with one or two neighbours: empty record slots stay "!")."""
    rng = np.random.Generator(np.random.PCG64(sum(params) * 7 + len(case)))
    if case == "crowded":
        grid, n = maps.pad(np.zeros((14, 14), np.uint8)), 170
    elif case == "corridor16":
        h, w = 41, 40
        g = np.zeros((h, w), np.uint8)
        for r in range(1, h, 2):
            g[r, :] = 1
            g[r, (w - 1) if (r // 2) % 2 == 0 else 0] = 0
        grid, n = maps.pad(g), 48
    elif case == "passes8":
        grid, n = maps.pad(maps.random_map(40, 44, 0.1, 3)), 500
    else:
        grid, n = maps.pad(maps.random_map(60, 30, 0.2, 4)), 9
    free = np.argwhere(maps.largest_component(grid == 0))
    pos = free[rng.permutation(len(free))[:n]].astype(np.int32)
    goal = free[rng.permutation(len(free))[:n]].astype(np.int32)
    tok = _compare_with_oracle(grid, pos, goal, steps=3, seed=11, params=params)
    if case == "corridor16":
        d = tok.distance_fields()
        assert ((d > 253) & (d < 65535)).any()


@pytest.mark.parametrize("params,grid_step", [((20, 13, 5, 3, 5), 16), ((15, 10, 3, 4, 2), 32), ((20, 13, 5, 5, 5), 16)])
def test_unseeded_corner_with_other_radii_and_grid_steps(params, grid_step):
    """The reference's unseeded (right, bottom) corner of a cached partial window (observation_generator.cpp:178-198) is the window's
    LAST cell for whatever obs_radius (the window origin divides by grid_step after subtracting obs_radius, cpp:181-183, 204-207):
    agents parked so that pos + obs_radius lands on (left + 2 grid_step, top + 2 grid_step), goals beyond the corner."""
    R = params[3]
    rng = np.random.Generator(np.random.PCG64(grid_step * 31 + R))
    grid = maps.pad((rng.random((70, 75)) < 0.05).astype(np.uint8))
    H, W = grid.shape
    sites = [(cr, cc) for cr in (2 * grid_step, 3 * grid_step) for cc in (2 * grid_step, 3 * grid_step) if cr <= H - 6 and cc <= W - 6]
    for cr, cc in sites:
        grid[cr - 2 * R - 1: cr + 2, cc - 2 * R - 1: cc + 2] = 0
    free = np.argwhere(maps.largest_component(grid == 0))
    n = 24
    pos = np.zeros((n, 2), np.int32)
    taken = set()
    for a in range(n):
        cr, cc = sites[a % len(sites)]
        cand = [(cr - R, cc - R)] if a < len(sites) else [(int(cr - R - rng.integers(0, R + 1)), int(cc - R - rng.integers(0, R + 1))) for _ in range(200)]
        for q in cand:
            if grid[q] == 0 and q not in taken:
                taken.add(q); pos[a] = q
                break
        else:
            k = next(k for k in rng.permutation(len(free)) if tuple(free[k]) not in taken)
            taken.add(tuple(free[k])); pos[a] = free[k]
    goal = free[np.argsort(-(free.sum(1)))[:n]].astype(np.int32)              # beyond every corner
    _compare_with_oracle(grid, pos, goal, steps=4, seed=5, params=params, grid_step=grid_step)


def test_long_corridor_needs_16bit_fields():
    """A serpentine corridor: shortest paths far beyond 253 cells, so the one-byte fields are invalid and the
    kernel must read the 16-bit ones (a saturated byte would change the window tokens)."""
    h, w = 41, 40
    g = np.zeros((h, w), np.uint8)
    for r in range(1, h, 2):
        g[r, :] = 1
        g[r, (w - 1) if (r // 2) % 2 == 0 else 0] = 0
    grid = maps.pad(g)
    free = np.argwhere(grid == 0)
    rng = np.random.Generator(np.random.PCG64(5))
    n = 48
    pos = free[rng.permutation(len(free))[:n]].astype(np.int32)
    goal = free[rng.permutation(len(free))[:n]].astype(np.int32)
    tok = _compare_with_oracle(grid, pos, goal, steps=3, seed=9)
    d = tok.distance_fields()
    assert ((d > 253) & (d < 65535)).any()


@pytest.mark.parametrize("h,w,central_goals,static_goals", [(140, 140, True, True), (140, 150, True, False), (220, 230, False, False)])
def test_partial_window_corner_vs_oracle(h, w, central_goals, static_goals):
    """Maps larger than 128 in both dimensions: the reference's cached partial window leaves its (right, bottom) corner
    unseeded (observation_generator.cpp:178-198), visible from (left + 123, top + 123) -- the per-agent window origin and the
    patched cell must follow the pinned oracle (tests/golden/tok_corner_*.npz) on the one-byte fields (central goals: all
    distances < 254), the 16-bit fields, and both update paths (static goals / goal checks)."""
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    rng = np.random.Generator(np.random.PCG64(h * 1000 + w))
    n_inst, n = 2, 20
    grid = maps.pad((rng.random((h, w)) < 0.08).astype(np.uint8))
    H, W = grid.shape
    sites = [(cr, cc) for cr in (128, 192) for cc in (128, 192) if cr <= H - 1 and cc <= W - 1]
    for cr, cc in sites:
        grid[cr - 9: min(cr + 3, H - 5), cc - 9: min(cc + 3, W - 5)] = 0
    comp = maps.largest_component(grid == 0)
    free = np.argwhere(comp)
    pos = np.zeros((n_inst, n, 2), np.int32)
    for i in range(n_inst):
        taken = set()
        for a in range(n):
            cr, cc = sites[a % len(sites)]
            if a < len(sites):                                               # one agent exactly on every corner-view spot
                taken.add((cr - 5, cc - 5))
                pos[i, a] = (cr - 5, cc - 5)
                continue
            for _ in range(10000):                                           # bounded: never spin on a full neighbourhood
                p = (int(cr - 5 - rng.integers(0, 6)), int(cc - 5 - rng.integers(0, 6)))
                if comp[p] and p not in taken:
                    taken.add(p)
                    pos[i, a] = p
                    break
            else:
                raise AssertionError("no free start cell near the corner site")
    if central_goals:
        mid = free[np.abs(free - np.array([H // 2, W // 2])).sum(1) < 30]
        goal = mid[rng.integers(0, len(mid), (n_inst, n))].astype(np.int32)
    else:
        goal = free[rng.integers(0, len(free), (n_inst, n))].astype(np.int32)
        goal[:, :6] = free[np.argsort(-(free.sum(1)))[:6]]                      # beyond every corner
    first = np.maximum(pos - 60, 5)                                           # created where the window origin is the site's
    for i in range(n_inst):
        used = set()
        for a in range(n):                                                    # nearest free cell nobody else took (the env never
            for k in np.argsort(np.abs(free - first[i, a]).sum(1)):           # puts two agents on one cell)
                if tuple(free[k]) not in used:
                    used.add(tuple(free[k]))
                    first[i, a] = free[k]
                    break
    gens = [orc.OracleGenerator(grid) for _ in range(n_inst)]
    tok = BatchedTokenizer(grid, n_inst, n)
    p, last = first.copy(), np.full((n_inst, n), -1, np.int32)
    hits = patched = 0
    for t in range(24):
        if t == 1:
            p = pos.copy()
        if t == 12 and not static_goals:
            goal[:, 10:] = free[rng.integers(0, len(free), (n_inst, n - 10))]
        dp, dg, da = _dev(p, torch.int16), _dev(goal, torch.int16), _dev(last, torch.int32)
        if t == 0:
            tok.create_agents(dp, dg)
            for i in range(n_inst):
                gens[i].create_agents(p[i], goal[i])
        tok.update_agents(dp, dg, da, goals_may_change=not static_goals)
        got = tok.generate_observations().cpu().numpy().reshape(n_inst, n, 256)
        for i in range(n_inst):
            gens[i].update_agents(p[i], goal[i], last[i])
            exp = gens[i].generate_observations()
            assert np.array_equal(got[i], exp), f"step {t} instance {i} rows {np.argwhere((got[i] != exp).any(1)).ravel().tolist()}"
            d = gens[i].dist()
            for a in range(n):
                if any(p[i, a, 0] + 5 == cr and p[i, a, 1] + 5 == cc for cr, cc in sites):
                    hits += 1
                    v, m = int(d[a][p[i, a, 0] + 5, p[i, a, 1] + 5]), int(d[a][p[i, a, 0], p[i, a, 1]])
                    plain = 41 if v == 65535 else (43 if v - m > 20 else 42 if v - m < -20 else v - m + 20)
                    patched += int(exp[a, 120] != plain)
        last = rng.integers(0, 5, (n_inst, n)).astype(np.int32)
        bias = rng.random((n_inst, n)) < 0.5
        last[bias] = rng.choice([2, 4], int(bias.sum()))
        for i in range(n_inst):
            p[i], _ = orc.env_step(grid, p[i], goal[i], last[i])
    assert hits > 0
    if not central_goals:
        assert patched > 0, "the walk never produced a row where the reference differs from the plain BFS distance"
    fields = tok.distance_fields()
    longest = int(fields[fields != 65535].max())                              # which of the two field widths the kernel read
    assert longest <= 253 if static_goals else (central_goals or longest > 253)


def test_vocabulary_size_and_the_pairing_with_a_policy():
    """Encoder's vocabulary has 2 * cost2go_value_limit + 27 tokens (cpp:321-350); a policy with 67 embedding rows cannot take the rows of a tokenizer with a
    larger limit (the reference: IndexError in nn.Embedding): mgpt_step_create and the adapter refuse the pairing, smaller vocabularies pass."""
    import ctypes
    from mapf_gpt_amd import _lib
    from mapf_gpt_amd.env import BatchedEnv
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.observation_generator import BatchedTokenizer, InputParameters
    grid, s_ok, g_ok = maps.load_named("validation-random-seed-000")
    net = build_model("tiny", max_rows=8)
    env = BatchedEnv(grid, 1, 8, 16)
    L = _lib.lib()
    for limit, vocab in ((20, 67), (10, 47), (30, 87)):
        tok = BatchedTokenizer(grid, 1, 8, InputParameters(limit, 13, 5, 256, 5, 5, 64, False))
        assert tok.vocab_size == vocab
        h = ctypes.c_void_p()
        rc = L.mgpt_step_create(ctypes.byref(h), tok._h, net._h, env._h, 8, 0, 0, 0, 0)
        if vocab <= 67:
            assert rc == _lib.OK
            L.mgpt_step_destroy(h)
        else:
            assert rc == _lib.ERR_UNSUPPORTED and b"87 tokens" in L.mgpt_last_error()
    with pytest.raises(ValueError, match="vocabulary of 87"):
        MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", cost2go_value_limit=30, agents_radius=5))
    assert MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", cost2go_value_limit=10)).input_parameters.cost2go_value_limit == 10
