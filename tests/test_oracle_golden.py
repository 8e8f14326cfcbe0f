"""The oracle is only trustworthy once pinned: C restatement vs the reference's golden vectors."""
import os

import numpy as np
import pytest

from mapf_gpt_amd import weights
from oracle import gpt_oracle
from oracle import oracle as orc
from tests.helpers import GOLDEN, load_tok, load_tokp, replay_oracle, sha_rows, tok_cases, tokp_cases


def test_known_answer_cpp_main():
    """observation_generator.cpp:530-544 (the reference's only known-answer scenario)."""
    g = np.load(os.path.join(GOLDEN, "tok_known_answer.npz"))
    gen = orc.OracleGenerator(np.zeros((256, 256), np.uint8))
    gen.create_agents([(120, 120)], [(20, 200)])
    gen.update_agents([(120, 120)], [(20, 200)], [0])
    row = gen.generate_observations()
    assert np.array_equal(row, g["tokens"])
    assert sha_rows(row[0]) == "896eb85aa89a369759917e5903f237dc28387e6b7d431fbd6703f302a97585e1"
    blk = row[0, :121].reshape(11, 11).astype(int)
    i, j = np.meshgrid(np.arange(11), np.arange(11), indexing="ij")
    assert np.array_equal(blk, 20 - j + i)
    assert row[0, 121:131].tolist() == [20, 20, 0, 40, 44, 44, 44, 44, 45, 59]
    assert (row[0, 131:] == 66).all()


@pytest.mark.parametrize("name", tokp_cases())
def test_tokenizer_oracle_matches_reference_vectors_of_other_input_parameters(name):
    """struct InputParameters (observation_generator.h:22-40) with values inference.py never passes: limit, record slots, history
    length and the two radii change the vocabulary and the row layout (cpp:321-389, 487-512)."""
    case = load_tokp(name)
    got = replay_oracle(case)
    assert sha_rows(got) == str(case["sha256_all_rows"])
    assert np.array_equal(got[:, case["keep"]], case["tokens"])


@pytest.mark.parametrize("name", tok_cases())
def test_tokenizer_oracle_matches_reference_vectors(name):
    case = load_tok(name)
    got = replay_oracle(case)
    assert sha_rows(got) == str(case["sha256_all_rows"]), "full-trajectory checksum differs from the reference"
    assert np.array_equal(got[:, case["keep"]], case["tokens"])


@pytest.mark.parametrize("tag,tol", [("tiny_s1", 2e-6), ("tiny_s4", 5e-6), ("2M_s1", 2e-6), ("2M_s4", 1e-5),
                                     ("6M_s1", 5e-6), ("85M_s1", 1e-5)])
def test_gpt_oracle_matches_reference_logits(tag, tol):
    """fp32 port vs the real model.py; tolerance = north_star's 1e-5 or tighter."""
    g = np.load(os.path.join(GOLDEN, f"gpt_{tag}.npz"))
    name = tag.split("_")[0]
    args = weights.model_args(name)
    sd = weights.synthetic_state_dict(name, seed=int(g["seed"]), scale=float(g["scale"]))
    logits = gpt_oracle.forward_logits(sd, args, g["tokens"]).numpy()
    assert np.abs(logits - g["logits"]).max() <= tol
    assert np.array_equal(gpt_oracle.act_greedy(gpt_oracle.forward_logits(sd, args, g["tokens"])).numpy(), g["greedy"])


@pytest.mark.parametrize("name", ["tiny", "2M", "6M"])
def test_gpt_oracle_with_bias_vectors_vs_reference(name):
    """GPTConfig.bias = True: the port's bias terms against the real model.py (tests/golden/make_golden_bias.py)."""
    g = np.load(os.path.join(GOLDEN, f"gptbias_{name}_s1.npz"))
    args = dict(weights.model_args(name), bias=True)
    sd = weights.synthetic_state_dict(args, seed=int(g["seed"]), scale=float(g["scale"]))
    logits = gpt_oracle.forward_logits(sd, args, g["tokens"]).numpy()
    assert np.abs(logits - g["logits"]).max() <= 1e-5
    no_bias = {k: v for k, v in sd.items() if not k.endswith(".bias")}
    assert np.abs(gpt_oracle.forward_logits(no_bias, args, g["tokens"]).numpy() - g["logits"]).max() > 0.1


def test_gpt_oracle_on_rows_shorter_than_256_tokens_vs_reference():
    """GPT.forward for T <= block_size (model.py:167-175) and a block_size = 100 model: the port against the real model.py
    (tests/golden/make_golden_short.py)."""
    g = np.load(os.path.join(GOLDEN, "gptshort.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert len(keys) == 14
    for k in keys:
        name, rest = k.split("_b")
        block, T = (int(x) for x in rest.split("_t"))
        if name == "85M":
            continue                                                   # (seconds of CPU per row; the GPU test covers it)
        args = dict(weights.model_args(name), block_size=block)
        sd = weights.synthetic_state_dict(args, seed=0, scale=1.0)
        assert g[k + "_tokens"].shape[1] == T
        logits = gpt_oracle.forward_logits(sd, args, g[k + "_tokens"]).numpy()
        assert np.abs(logits - g[k + "_logits"]).max() <= 1e-5, k


def test_gpt_oracle_layers_tiny():
    g = np.load(os.path.join(GOLDEN, "gpt_tiny_s1.npz"))
    args = weights.model_args("tiny")
    sd = weights.synthetic_state_dict("tiny", seed=0, scale=1.0)
    _, layers = gpt_oracle.forward_logits(sd, args, g["tokens"][:1], return_layers=True)
    got = np.stack([l[0].numpy() for l in layers])
    assert np.abs(got - g["layers"]).max() <= 2e-6
