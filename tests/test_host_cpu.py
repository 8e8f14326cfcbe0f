"""Host-side logic that needs no GPU: config surface, maps, placement, sampler restatement, sharding."""
import os

import numpy as np
import pytest

from mapf_gpt_amd import maps, sampling, weights
from mapf_gpt_amd.runner import shard_range


def test_config_mirrors_reference_fields_and_forbids_extras():
    from pydantic import ValidationError
    from mapf_gpt_amd.inference import MAPFGPTInferenceConfig
    c = MAPFGPTInferenceConfig()
    # defaults of mapf_gpt/inference.py:13-31
    assert (c.name, c.num_agents, c.num_previous_actions, c.cost2go_value_limit, c.agents_radius, c.cost2go_radius) == \
        ("MAPF-GPT", 13, 5, 20, 5, 5)
    assert (c.path_to_weights, c.device, c.context_size, c.repo_id, c.grid_step, c.save_cost2go, c.batch_size, c.num_process) == \
        ("weights/MAPF-GPT-2M.pt", None, 256, "aandreychuk/MAPF-GPT", 64, False, 2048, 8)
    assert not (c.mask_actions_history or c.mask_goal or c.mask_cost2go or c.mask_greed_action)
    MAPFGPTInferenceConfig(parallel_backend="balanced_dask", num_process=4)     # 01-random.yaml:147-148
    with pytest.raises(ValidationError):
        MAPFGPTInferenceConfig(not_a_field=1)                                   # extra=forbid, inference.py:13
    # round 5: the adapter's default is the split-fp16 mode UNDER the library's envelope guard (a checkpoint outside the validated
    # range is served by the exact-fp32 kernels); "f32" stays selectable (ADVICE r03 / VERDICT r04 weak item 3)
    assert c.precision == "f16x3" and c.envelope == "fallback" and MAPFGPTInferenceConfig(precision="f32").precision == "f32"
    with pytest.raises(ValidationError):
        MAPFGPTInferenceConfig(precision="fp8")


def test_named_maps_and_padding():
    for name, shape in {"validation-random-seed-000": (30, 31), "validation-mazes-seed-000": (31, 31),
                        "wfi_warehouse": (43, 56), "Berlin_1_256_00": (74, 74), "puzzle-00": (15, 15)}.items():
        g, s, t = maps.load_named(name)
        assert g.shape == shape and g[:5].all() and g[-5:].all() and g[:, :5].all() and g[:, -5:].all()
        assert not (s & (g != 0)).any() and not (t & (g != 0)).any()
    g, s, t = maps.load_named("wfi_warehouse")
    assert s.sum() < (g == 0).sum() and t.sum() < (g == 0).sum()      # '@' / '$' restrict starts / goals


def test_placement_distinct_connected_and_seeded():
    g, s, t = maps.load_named("validation-mazes-seed-000")
    p1, q1 = maps.place_agents(g, 64, 3, s, t)
    p2, q2 = maps.place_agents(g, 64, 3, s, t)
    assert np.array_equal(p1, p2) and np.array_equal(q1, q2)
    assert len({tuple(x) for x in p1}) == 64 and len({tuple(x) for x in q1}) == 64
    comp = maps.largest_component(g == 0)
    assert comp[p1[:, 0], p1[:, 1]].all() and comp[q1[:, 0], q1[:, 1]].all()
    with pytest.raises(ValueError):
        maps.place_agents(maps.pad(np.zeros((3, 3), np.uint8)), 10, 0)


def test_synthetic_generators_shapes():
    m = maps.maze_map(21, 21, 5)
    assert m.shape == (21, 21) and 0.15 < m.mean() < 0.5
    r = maps.random_map(20, 21, 0.15, 1)
    assert r.shape == (20, 21) and 0.05 < r.mean() < 0.3


def test_weights_layout_matches_reference_key_list():
    sd = weights.synthetic_state_dict("2M", seed=0)
    assert sum(v.size for k, v in sd.items() if k != "lm_head.weight") == 1_589_440
    assert sd["lm_head.weight"] is sd["transformer.wte.weight"]
    assert sd["transformer.h.4.attn.c_attn.weight"].shape == (480, 160)
    assert sum(v.size for k, v in weights.synthetic_state_dict("6M").items() if k != "lm_head.weight") == 6_378_496
    again = weights.synthetic_state_dict("2M", seed=0)
    assert all(np.array_equal(sd[k], again[k]) for k in sd)


def test_sampler_restatement_statistics():
    logits = np.tile(np.log(np.array([0.5, 0.2, 0.1, 0.1, 0.1], np.float32)), (200000, 1))
    act, _ = sampling.sample(logits, seed=7, step=3)
    freq = np.bincount(act, minlength=5) / len(act)
    assert np.abs(freq - [0.5, 0.2, 0.1, 0.1, 0.1]).max() < 5e-3
    a2, _ = sampling.sample(logits[:100], seed=7, step=3)
    assert np.array_equal(a2, act[:100])                         # counter-based: prefix-stable
    a3, _ = sampling.sample(logits[50:100], seed=7, step=3, row0=50)
    assert np.array_equal(a3, act[50:100])                       # keyed by global row id
    g, _ = sampling.sample(logits[:4], 0, 0, do_sample=False)
    assert (g == 0).all()


def test_shard_range_partitions():
    for n, w in [(256, 8), (257, 8), (5, 8), (4096, 3)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_bench_flop_accounting_matches_survey():
    """bench.py's per-row flops are SURVEY 8d's F_ref = L (24 C^2 T + 4 T^2 C) + 2 C V: 0.996 / 3.758 / 45.90 GFLOP."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from mapf_gpt_amd import weights
    for name, gflop in (("2M", 0.996), ("6M", 3.758), ("85M", 45.90)):
        total, per_layer = bench.flops_per_row(weights.model_args(name))
        assert abs(total / 1e9 - gflop) < 0.005
        a = weights.model_args(name)
        fused = per_layer["gpt_gemm_qkv"] + per_layer["gpt_attention"] + per_layer["gpt_gemm_attn_proj"] + per_layer["gpt_mlp_fused"]
        assert a["n_layer"] * fused + 2 * a["n_embd"] * 67 == total
        assert per_layer["gpt_gemm_mlp_fc"] + per_layer["gpt_gemm_mlp_proj"] == per_layer["gpt_mlp_fused"]
    assert set(bench.WORKLOADS) == {"cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "env6M"} and bench.TOKENIZER_BYTES_PER_ROW == 694


def test_gpt_envelope_policy_names():
    """GPT(..., envelope=...) accepts the three policies of include/mapf_gpt_amd.h (MGPT_ENVELOPE_*) and nothing else; the context is
    only created on first use, so this needs no device."""
    from mapf_gpt_amd.model import GPT, GPTConfig
    cfg = GPTConfig(block_size=256)
    assert GPT.ENVELOPE_POLICIES == {"fallback": 0, "refuse": 1, "ignore": 2}
    for name in GPT.ENVELOPE_POLICIES:
        assert GPT(cfg, envelope=name).envelope_policy == name
    with pytest.raises(ValueError):
        GPT(cfg, envelope="sometimes")
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mapf_gpt_amd.h")).read()
    for name, value in (("FALLBACK", 0), ("REFUSE", 1), ("IGNORE", 2)):
        assert f"#define MGPT_ENVELOPE_{name} {value}" in header


def test_gpt_config_flags_accepted_at_construction():
    """model.py:33-34,59,82,129: every Dropout of the reference is the identity once inference.py:85 has called net.eval(); this inference-only
    implementation therefore accepts any dropout in [0, 1) and ignores it.  bias=True (Linear / LayerNorm biases, model.py:17,29-31,79-81) and the
    reference's default block_size (161, model.py:109) are accepted too (tests/test_gpu_gpt.py has the parity side) -- no device needed here; rows longer
    than block_size are refused in the reference's words (model.py:170)."""
    import torch
    from mapf_gpt_amd.model import GPT, GPTConfig
    assert GPT(GPTConfig(block_size=256, dropout=0.1)).config.dropout == 0.1
    assert GPT(GPTConfig(block_size=256, bias=True)).config.bias is True
    net = GPT(GPTConfig())
    assert net.config.block_size == 161
    with pytest.raises(ValueError, match="Cannot forward sequence of length 162, block size is only 161"):
        net._tokens_u8(torch.zeros((1, 162), dtype=torch.int64))
    with pytest.raises(ValueError, match="dropout"):
        GPT(GPTConfig(block_size=256, dropout=1.5))


def test_animation_writer_svg(tmp_path):
    """mapf_gpt_amd/animation.py (the role of env.save_animation, example.py:66-70): well-formed SVG, one disc + one ring per agent, key frames = recorded
    frames, the padded border cropped, a standing agent not animated, a goal change moves the ring."""
    import xml.etree.ElementTree as ET
    from mapf_gpt_amd import animation
    grid = maps.pad(np.array([[0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 0, 0]], np.uint8))
    frames = [np.array([[5, 5], [7, 8]]), np.array([[6, 5], [7, 8]]), np.array([[6, 6], [7, 8]])]
    goals = [np.array([[7, 7], [5, 7]]), np.array([[7, 7], [5, 7]]), np.array([[7, 7], [6, 5]])]
    path = animation.write_svg(str(tmp_path / "sub" / "ep.svg"), grid, frames, goals, seconds_per_step=0.5)
    root = ET.parse(path).getroot()
    ns = "{http://www.w3.org/2000/svg}"
    assert root.attrib["viewBox"] == "0 0 80 60"                                # 4 x 3 cells of 20 px: border cropped
    rects = root.findall(ns + "rect")
    assert len(rects) == 1 + 2                                                   # background + two obstacles
    assert {(r.attrib["x"], r.attrib["y"]) for r in rects[1:]} == {("20", "0"), ("60", "20")}
    circles = root.findall(ns + "circle")
    assert len(circles) == 4
    rings, discs = circles[:2], circles[2:]
    assert all(c.attrib["fill"] == "none" for c in rings) and rings[0].attrib["stroke"] == discs[0].attrib["fill"] != discs[1].attrib["fill"]
    a0 = discs[0].findall(ns + "animate")
    assert len(a0) == 2 and a0[0].attrib["values"] == "10;10;30" and a0[1].attrib["values"] == "10;30;30" and a0[0].attrib["dur"] == "1s"
    assert a0[0].attrib["keyTimes"] == "0.00000;0.50000;1.00000"
    assert discs[1].findall(ns + "animate") == []                               # agent 1 never moved
    assert rings[0].findall(ns + "animate") == [] and len(rings[1].findall(ns + "animate")) == 2    # only agent 1's goal changed
    assert rings[1].findall(ns + "animate")[0].attrib["calcMode"] == "discrete"
    with pytest.raises(ValueError):
        animation.write_svg(str(tmp_path / "none.svg"), grid, [], goals)
