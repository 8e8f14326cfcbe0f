"""Evaluation harness on the device: the smoke config end to end, result schema, determinism, metric sanity."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_smoke_config_end_to_end(tmp_path):
    from mapf_gpt_amd import evaluation as ev
    cfg = ev.load_yaml(os.path.join(ROOT, "eval_configs", "00-smoke", "00-smoke.yaml"))
    cfg["algorithms"]["MAPF-GPT-2M"]["path_to_weights"] = "synthetic:tiny"
    lines = []
    res = ev.evaluation(cfg, eval_dir=str(tmp_path), print_fn=lines.append)
    assert len(res) == 16
    assert os.path.exists(tmp_path / "MAPF-GPT-2M.json")
    for r in res:
        assert set(r) == {"metrics", "env_grid_search", "algorithm"}
        assert set(r["env_grid_search"]) == {"seed", "num_agents", "map_name"}
        m = r["metrics"]
        assert set(m) == {"CSR", "ISR", "SoC", "makespan", "ep_length", "avg_agents_density", "runtime"}
        assert 0.0 < m["avg_agents_density"] <= 1.0
        assert 0.0 <= m["ISR"] <= 1.0 and m["CSR"] in (0.0, 1.0) and 0 < m["ep_length"] <= 64
        assert m["makespan"] <= m["ep_length"] and m["SoC"] <= r["env_grid_search"]["num_agents"] * m["ep_length"]
    assert lines[0] == "== TabularView1" and lines[1].split()[:2] == ["num_agents", "algorithm"]
    res2 = ev.evaluation(cfg, print_fn=lambda *_: None)
    key = lambda r: tuple(sorted(r["env_grid_search"].items()))
    a = {key(r): [r["metrics"][k] for k in ev.METRIC_KEYS] for r in res}
    b = {key(r): [r["metrics"][k] for k in ev.METRIC_KEYS] for r in res2}
    assert a == b                                              # same seeds, same device sampler -> same episodes


def test_batch_composition_does_not_change_an_episode():
    """An instance's episode depends on its own (map, seed, agents) only: running it inside a bigger batch (other maps
    padded to a larger common frame) gives the same metrics as running it alone."""
    from mapf_gpt_amd import evaluation as ev
    base = {"environment": {"name": "Environment", "on_target": "nothing", "max_episode_steps": 32, "seed": 3, "num_agents": 8,
                            "map_name": {"grid_search": ["validation-random-seed-000"]}},
            "algorithms": {"A": {"name": "MAPF-GPT", "path_to_weights": "synthetic:tiny", "precision": "f16x3"}}}
    alone = ev.evaluation(base, print_fn=lambda *_: None)
    base["environment"]["map_name"]["grid_search"] = ["wfi_warehouse", "validation-random-seed-000"]
    both = ev.evaluation(base, print_fn=lambda *_: None)
    m0 = [alone[0]["metrics"][k] for k in ("CSR", "ISR", "SoC", "makespan", "ep_length")]
    m1 = [both[1]["metrics"][k] for k in ("CSR", "ISR", "SoC", "makespan", "ep_length")]
    assert both[1]["env_grid_search"]["map_name"] == "validation-random-seed-000"
    assert m0 == m1


def test_lifelong_config_reports_throughput():
    from mapf_gpt_amd import evaluation as ev
    cfg = {"environment": {"name": "Environment", "on_target": "restart", "max_episode_steps": 32, "num_agents": 8,
                           "seed": {"grid_search": [0, 1]}, "map_name": "validation-random-seed-000"},
           "algorithms": {"A": {"name": "MAPF-GPT", "path_to_weights": "synthetic:tiny", "precision": "f16x3"}},
           "results_views": {"T": {"type": "tabular", "drop_keys": ["seed"], "print_results": True}}}
    lines = []
    res = ev.evaluation(cfg, print_fn=lines.append)
    assert len(res) == 2
    for r in res:
        assert set(r["metrics"]) == {"avg_throughput", "ep_length", "runtime"}
        assert r["metrics"]["ep_length"] == 32 and r["metrics"]["avg_throughput"] >= 0.0
    assert "avg_throughput" in lines[1]
