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
