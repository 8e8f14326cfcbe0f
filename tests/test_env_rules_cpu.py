"""The env spec's switchable collision rules on the oracle (CPU): every mask keeps the hard invariants -- no vertex conflict,
no obstacle entry, no edge swap, at most one cell per step -- and each switch means what the header says.
PARITY UNPINNED either way (POGEMA absent): tests/test_env_pogema_fixtures.py pins the mask once fixtures exist."""
import numpy as np
import pytest

from mapf_gpt_amd import maps
from oracle import oracle as orc


@pytest.mark.parametrize("rules", [0, 1, 2, 3])
def test_invariants_under_every_mask(rules):
    grid, s_ok, g_ok = maps.load_named("validation-mazes-seed-000")
    rng = np.random.Generator(np.random.PCG64(100 + rules))
    p, g = maps.place_agents(grid, 80, 3)
    p, g = p.astype(np.int32), g.astype(np.int32)
    for t in range(60):
        act = rng.integers(0, 5, 80).astype(np.int32)
        if t % 4 == 0:
            act[:] = rng.integers(1, 5)
        new, _ = orc.env_step(grid, p, g, act, rules=rules)
        assert len({tuple(x) for x in new}) == 80
        assert (grid[new[:, 0], new[:, 1]] == 0).all()
        assert (np.abs(new - p).sum(1) <= 1).all()
        old = {tuple(x): a for a, x in enumerate(p)}
        for a in range(80):
            b = old.get(tuple(new[a]))
            if b is not None and b != a:
                assert tuple(new[b]) != tuple(p[a]), "edge swap"
                if rules & orc.RULE_NO_FOLLOW:
                    raise AssertionError("an agent entered a cell that was occupied at the start of the step")
        p = new


def test_what_each_switch_means():
    grid = np.zeros((5, 7), np.uint8)
    goal = np.zeros((3, 2), np.int32)
    # a chain: agent 0 moves right into the cell agent 1 leaves (also moving right)
    pos = np.array([[2, 1], [2, 2], [4, 6]], np.int32)
    act = np.array([4, 4, 0], np.int32)
    assert orc.env_step(grid, pos, goal, act)[0].tolist() == [[2, 2], [2, 3], [4, 6]]                      # following allowed
    assert orc.env_step(grid, pos, goal, act, rules=orc.RULE_NO_FOLLOW)[0].tolist() == [[2, 1], [2, 3], [4, 6]]
    # two agents claim the empty cell (2, 3)
    pos = np.array([[2, 2], [2, 4], [4, 6]], np.int32)
    act = np.array([4, 3, 0], np.int32)
    assert orc.env_step(grid, pos, goal, act)[0].tolist() == [[2, 2], [2, 4], [4, 6]]                      # nobody gets it
    assert orc.env_step(grid, pos, goal, act, rules=orc.RULE_LOWEST_WINS)[0].tolist() == [[2, 3], [2, 4], [4, 6]]
    # a staying agent always keeps its cell
    pos = np.array([[2, 2], [2, 3], [4, 6]], np.int32)
    act = np.array([4, 0, 0], np.int32)
    for r in (0, 1, 2, 3):
        assert orc.env_step(grid, pos, goal, act, rules=r)[0].tolist() == [[2, 2], [2, 3], [4, 6]]
