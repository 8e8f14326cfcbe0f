"""Dataset-side bulk tokenizer on the device vs the goldens made by the reference's own dataset tokenizer and vs the
restatement (oracle/dataset_oracle.py) on larger synthetic logs; through the C ABI (mgpt_dataset_*)."""
import glob
import os

import numpy as np
import pytest

from mapf_gpt_amd import maps
from oracle import dataset_oracle as dso
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ds_*.npz")))


def _to_str(obst):
    return "\n".join("".join("#" if v else "." for v in row) for row in obst)


@pytest.mark.parametrize("name", CASES)
def test_reference_goldens_bit_exact(name):
    from mapf_gpt_amd.dataset_tokenizer import InputParameters, ObservationGenerator
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    obst = g["grid"][5:-5, 5:-5]
    first = {"CSR": 1.0, "made_actions": g["made_actions"].tolist(), "init_positions": g["init_positions"].tolist()}
    if "lifelong_targets" in g:                                             # lifelong log (generate_observations.py:55-60)
        first["global_lifelong_targets_xy"] = g["lifelong_targets"].tolist()
    data = [{"metrics": first, "env_grid_search": {"map_name": "m"}},
            {"metrics": {"CSR": 0.0, "made_actions": g["made_actions"].tolist(), "init_positions": g["init_positions"].tolist()},
             "env_grid_search": {"map_name": "m"}}]                         # skipped: not solved
    gen = ObservationGenerator({"m": _to_str(obst)}, data, InputParameters(mask_cost2go="mask_cost2go" in g))
    inputs, gts = gen.generate_observations(0, 2)
    assert np.array_equal(np.stack(inputs), g["inputs"])
    assert np.array_equal(np.array(gts), g["gt_actions"])


@pytest.mark.parametrize("n_agents,steps,h,w,seed", [(40, 12, 21, 21, 0), (150, 6, 30, 34, 1), (3, 1, 9, 9, 2)])
def test_random_logs_vs_restatement(n_agents, steps, h, w, seed):
    """Crowded windows (150 agents), one-step episodes and ragged sizes; checker = the pinned restatement."""
    from mapf_gpt_amd.dataset_tokenizer import MapTable, agent_paths
    obst = maps.random_map(h, w, 0.12, 70 + seed)
    grid = maps.pad(obst)
    pos, goal = maps.place_agents(grid, n_agents, seed)
    pos, goal = pos.astype(np.int32), goal.astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(seed))
    init, made = pos.copy(), [[] for _ in range(n_agents)]
    for _ in range(steps):
        new, _ = orc.env_step(grid, pos, goal, rng.integers(0, 5, n_agents).astype(np.int32))
        for a in range(n_agents):
            made[a].append({(0, 0): 0, (-1, 0): 1, (1, 0): 2, (0, -1): 3, (0, 1): 4}[(int(new[a][0] - pos[a][0]), int(new[a][1] - pos[a][1]))])
        pos = new
    exp, _ = dso.generate_observations(grid, init, made)
    got = MapTable(grid).tokenize(agent_paths(init, made)).cpu().numpy().astype(np.int8).reshape(-1, 256)
    assert np.array_equal(got, exp)
