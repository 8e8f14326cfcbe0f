"""Direct oracle-vs-reference comparison; runs only where oracle/_ref was built (build container,
and the GPU box, to which the built .so travels).  Never reads /root/reference."""
import glob
import os
import sys

import numpy as np
import pytest

from mapf_gpt_amd import maps
from oracle import oracle as orc

REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
have_ref = bool(glob.glob(os.path.join(REF_DIR, "observation_generator*.so")))
pytestmark = pytest.mark.skipif(not have_ref, reason="oracle/_ref not built (needs /root/reference at build time)")


def _ref():
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import observation_generator as og
    return og


@pytest.mark.parametrize("h,w,dens,n,seed", [(20, 21, 0.15, 24, 1), (30, 90, 0.25, 30, 2), (70, 66, 0.3, 20, 3),
                                             (130, 40, 0.1, 16, 4), (64, 64, 0.35, 40, 5)])
def test_random_maps_bit_exact(h, w, dens, n, seed):
    og = _ref()
    grid = maps.pad(maps.random_map(h, w, dens, seed))
    pos, goal = maps.place_agents(grid, n, seed)
    ref = og.ObservationGenerator(grid.astype(int).tolist(), og.InputParameters(20, 13, 5, 256, 5, 5, 64, False))
    mine = orc.OracleGenerator(grid)
    rng = np.random.Generator(np.random.PCG64(seed))
    p, g, last = pos.astype(np.int32), goal.astype(np.int32), np.full(n, -1, np.int32)
    for t in range(10):
        pl, gl = [tuple(map(int, x)) for x in p], [tuple(map(int, x)) for x in g]
        if t == 0:
            ref.create_agents(pl, gl)
            mine.create_agents(p, g)
        ref.update_agents(pl, gl, [int(x) for x in last])
        mine.update_agents(p, g, last)
        assert np.array_equal(np.array(ref.generate_observations(), dtype=np.uint8), mine.generate_observations())
        last = rng.integers(0, 5, n).astype(np.int32)
        p, _ = orc.env_step(grid, p, g, last)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_partial_window_corner_walks(seed):
    """Maps larger than 128 in both dimensions: agents wander around the (left + 123, top + 123) spots of their cached
    partial windows, where the reference's unseeded window corner (cpp:178-198) is in view -- recompute rule and corner
    value must follow the reference step by step (plain BFS differs there)."""
    og = _ref()
    rng = np.random.Generator(np.random.PCG64(500 + seed))
    grid = maps.pad((rng.random((220, 230)) < 0.1).astype(np.uint8))
    sites = [(128, 128), (128, 192), (192, 128), (192, 192)]
    for cr, cc in sites:
        grid[cr - 9: cr + 3, cc - 9: cc + 3] = (rng.random((12, 12)) < 0.08)
    comp = maps.largest_component(grid == 0)
    n = 24
    starts = []
    for k in range(n):                                         # starts inside the windows whose corner is the site
        cr, cc = sites[k % 4]
        for _ in range(10000):
            p = (int(cr - 5 - rng.integers(0, 4)), int(cc - 5 - rng.integers(0, 4)))
            if comp[p] and p not in starts:
                starts.append(p)
                break
        else:
            raise AssertionError("no free start cell near the corner site")
    free = np.argwhere(comp)
    goal = free[rng.integers(0, len(free), n)].astype(np.int32)
    goal[:8] = free[np.argsort(-(free.sum(1)))[:8]]            # some goals beyond every corner
    first = np.array([(p[0] - 60, p[1] - 60) for p in starts], np.int32)     # created with the origin the site belongs to
    for k in range(n):
        if not comp[tuple(first[k])]:
            d = np.abs(free - first[k]).sum(1)
            first[k] = free[d.argmin()]
    ref = og.ObservationGenerator(grid.astype(int).tolist(), og.InputParameters(20, 13, 5, 256, 5, 5, 64, False))
    mine = orc.OracleGenerator(grid)
    p, last = first.copy(), np.full(n, -1, np.int32)
    hits = 0
    for t in range(40):
        if t == 1:
            p = np.array(starts, np.int32)                     # positions are inputs: a jump into the corner region
        if t == 20:
            goal[8:16] = free[rng.integers(0, len(free), 8)]   # goal changes reset the window origin (cpp:464-468)
        pl, gl = [tuple(map(int, x)) for x in p], [tuple(map(int, x)) for x in goal]
        if t == 0:
            ref.create_agents(pl, gl)
            mine.create_agents(p, goal)
        ref.update_agents(pl, gl, [int(x) for x in last])
        mine.update_agents(p, goal, last)
        want = np.array(ref.generate_observations(), dtype=np.uint8)
        got = mine.generate_observations()
        assert np.array_equal(want, got), f"step {t}: rows {np.argwhere((want != got).any(1)).ravel().tolist()}"
        hits += sum(1 for a in range(n) if any(p[a][0] + 5 == cr and p[a][1] + 5 == cc for cr, cc in sites))
        last = rng.integers(0, 5, n).astype(np.int32)
        bias = rng.random(n) < 0.5                             # drift towards the spot: down / right
        last[bias] = rng.choice([2, 4], bias.sum())
        p, _ = orc.env_step(grid, p, goal, last)
    assert hits > 0, "no agent ever stood on a corner-view spot"
