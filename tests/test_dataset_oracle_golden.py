"""The dataset-side tokenizer restatement (oracle/dataset_oracle.py) against golden rows produced by the reference's own
dataset/tokenizer code (tests/golden/make_golden_dataset.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import dataset_oracle as dso

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ds_*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_dataset_oracle_matches_reference_rows(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rows, gts = dso.generate_observations(g["grid"], g["init_positions"], g["made_actions"].tolist())
    assert rows.shape == g["inputs"].shape
    assert np.array_equal(rows, g["inputs"])
    assert np.array_equal(gts, g["gt_actions"])
    assert (g["gt_actions"] == 5).any() or name == "ds_short"        # "wait in goal" labels are exercised


def test_history_padding_rules():
    """Episode start pads with 'n' (44), the last step with one 'w' (45): generate_observations.py:206-228."""
    grid = np.ones((16, 16), np.uint8)
    grid[5:11, 5:11] = 0
    rows, gts = dso.generate_observations(grid, [[5, 5]], [[4, 4, 2, 0, 3, 3, 1]])
    own = rows[:, 121 + 4: 121 + 9]                                    # r r d w l l u = 49 49 47 45 48 48 46
    assert own[0].tolist() == [44, 44, 44, 44, 44]
    assert own[1].tolist() == [44, 44, 44, 44, 49]
    assert own[5].tolist() == [49, 49, 47, 45, 48]
    assert own[6].tolist() == [49, 47, 45, 48, 48]
    assert own[7].tolist() == [47, 45, 48, 48, 45]                     # last step: the newest slot is the 'w' pad
    assert gts.tolist() == [4, 4, 2, 0, 3, 3, 1, 5]                    # the appended step is "wait in goal"


def test_host_encoder_round_trip_on_reference_rows():
    """decode -> encode reproduces the reference's rows (empty slots decode to '!' records and are dropped again)."""
    from mapf_gpt_amd.dataset_tokenizer import Encoder
    enc = Encoder()
    g = np.load(os.path.join(GOLDEN, "ds_random.npz"))
    for row in g["inputs"][::17]:
        obs = enc.decode(row)
        assert obs["cost2go"][5, 5] == 0                                     # the observer's own cell
        assert obs["agents"][0]["relative_pos"] == (0, 0)                    # slot 0 is the observer
        obs["agents"] = [a for a in obs["agents"] if a["next_action"] != "!"]
        assert enc.encode(obs) == [int(v) for v in row]
