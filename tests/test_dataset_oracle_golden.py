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
    rows, gts = dso.generate_observations(g["grid"], g["init_positions"], g["made_actions"].tolist(),
                                          lifelong_targets=g["lifelong_targets"] if "lifelong_targets" in g else None,
                                          mask_cost2go="mask_cost2go" in g)
    assert rows.shape == g["inputs"].shape
    assert np.array_equal(rows, g["inputs"])
    assert np.array_equal(gts, g["gt_actions"])
    assert (g["gt_actions"] == 5).any() or name == "ds_short"        # "wait in goal" labels are exercised
    if "lifelong_targets" in g:                                      # the goal really moves inside the log
        own_goal = rows[:, 121 + 2: 121 + 4].reshape(len(g["init_positions"]), -1, 2)
        assert any(len({tuple(v) for v in (own_goal[a] + rows.reshape(len(own_goal), -1, 256)[a, :, 121:123] * 0).tolist()}) > 2 for a in range(len(own_goal)))
    if "mask_cost2go" in g:
        assert set(np.unique(rows[:, :121]).tolist()) == {20, 21}


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
    from oracle.dataset_encoder import Encoder
    enc = Encoder()
    g = np.load(os.path.join(GOLDEN, "ds_random.npz"))
    for row in g["inputs"][::17]:
        obs = enc.decode(row)
        assert obs["cost2go"][5, 5] == 0                                     # the observer's own cell
        assert obs["agents"][0]["relative_pos"] == (0, 0)                    # slot 0 is the observer
        obs["agents"] = [a for a in obs["agents"] if a["next_action"] != "!"]
        assert enc.encode(obs) == [int(v) for v in row]


def test_host_encoder_masks_against_reference_outputs():
    """Encoder.mask vs outputs of the reference's python Encoder.mask (tests/golden/enc_masks.npz, made by make_golden_dataset.py)."""
    from mapf_gpt_amd.dataset_tokenizer import InputParameters
    from oracle.dataset_encoder import Encoder
    g = np.load(os.path.join(GOLDEN, "enc_masks.npz"))
    flags = ("mask_actions_history", "mask_goal", "mask_greed_action", "mask_cost2go")
    for key in flags + ("all",):
        enc = Encoder(InputParameters(**({f: True for f in flags} if key == "all" else {key: True})))
        for row, exp in zip(g["rows"], g[key]):
            assert enc.mask([int(v) for v in row]) == [int(v) for v in exp], key


def test_host_encoder_masks_known_answers():
    """Encoder.mask = tokenizer.py:104-138 (known answers worked out from that code on one golden row)."""
    from mapf_gpt_amd.dataset_tokenizer import InputParameters
    from oracle.dataset_encoder import Encoder
    row = [int(v) for v in np.load(os.path.join(GOLDEN, "ds_random.npz"))["inputs"][40]]
    base = Encoder().decode(row)
    for flag in ("mask_actions_history", "mask_goal", "mask_greed_action", "mask_cost2go"):
        enc = Encoder(InputParameters(**{flag: True}))
        got = enc.mask(list(row))
        exp = list(row)
        for i in range(13):
            o = 121 + 10 * i
            if flag == "mask_actions_history":
                exp[o + 4: o + 9] = [66] * 5
            if flag == "mask_goal":
                exp[o + 2] = exp[o + 3] = 66
            if flag == "mask_greed_action":
                exp[o + 9] = 66
        if flag == "mask_cost2go":
            exp[:121] = [41 if v == 41 else 20 for v in row[:121]]
        assert got == exp, flag
        obs = enc.decode(row)                                              # decode masks first (tokenizer.py:141-149)
        if flag == "mask_goal":
            assert all(a["relative_goal"] == ("!", "!") for a in obs["agents"])
        if flag == "mask_cost2go":
            assert set(np.unique(obs["cost2go"]).tolist()) <= {0, -80}
        assert obs["agents"][0]["relative_pos"] == base["agents"][0]["relative_pos"]
