"""Host logic of the evaluation harness (mapf_gpt_amd/evaluation.py): config expansion, framing, views."""
import os

import numpy as np

from mapf_gpt_amd import evaluation as ev

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grid_search_is_the_product_in_key_order():
    env = {"name": "Environment", "max_episode_steps": 128, "seed": {"grid_search": [0, 1, 2]},
           "num_agents": {"grid_search": [8, 16]}, "map_name": "m"}
    runs = ev.expand_grid_search(env)
    assert len(runs) == 6
    assert [p for _, p in runs][:3] == [{"seed": 0, "num_agents": 8}, {"seed": 0, "num_agents": 16}, {"seed": 1, "num_agents": 8}]
    cfg, point = runs[-1]
    assert cfg["map_name"] == "m" and cfg["seed"] == 2 and cfg["num_agents"] == 16 and cfg["max_episode_steps"] == 128
    assert ev.expand_grid_search({"a": 1}) == [({"a": 1}, {})]


def test_smoke_config_parses_and_groups():
    cfg = ev.load_yaml(os.path.join(ROOT, "eval_configs", "00-smoke", "00-smoke.yaml"))
    runs = ev.expand_grid_search(cfg["environment"])
    assert len(runs) == 4 * 2 * 2
    groups = ev.group_runs(runs)
    assert list(groups) == [(8, 64, "nothing"), (16, 64, "nothing")]
    assert all(len(v) == 8 for v in groups.values())
    from mapf_gpt_amd.inference import MAPFGPTInferenceConfig
    for algo in cfg["algorithms"].values():
        MAPFGPTInferenceConfig(**algo)                     # the reference's YAML keys are accepted, unknown ones raise


def test_common_frame_pads_with_obstacles_and_keeps_cells():
    reg = ev.MapRegistry()
    reg.register_maps({"tiny": "..#\n...", "wide": ".....\n.@.$."})
    parsed = [reg.get("tiny"), reg.get("wide")]
    frames = ev.common_frame(parsed)
    assert all(f[0].shape == (2 + 10, 5 + 10) for f in frames)
    g0, s0, t0 = frames[0]
    assert g0[5:7, 5:8].tolist() == [[0, 0, 1], [0, 0, 0]] and g0[:, 8:].all() and g0[:5].all() and g0[7:].all()
    g1, s1, t1 = frames[1]
    assert s1.sum() == 1 and s1[6, 6] and t1.sum() == 1 and t1[6, 8]       # '@' start-only, '$' goal-only cells
    assert not (s0 & (g0 != 0)).any()


def test_tabular_view_drops_and_averages():
    res = []
    for n in (8, 16):
        for seed in (0, 1):
            res.append({"metrics": {"CSR": float(seed), "ISR": 0.5, "SoC": 10.0 * n, "makespan": 20.0, "ep_length": 64.0, "runtime": 0.1},
                        "env_grid_search": {"seed": seed, "num_agents": n}, "algorithm": "A"})
    lines = []
    table = ev.tabular_view(res, {"type": "tabular", "drop_keys": ["seed", "runtime"], "print_results": True, "round_digits": 3},
                            print_fn=lines.append)
    assert [t["num_agents"] for t in table] == [8, 16]
    assert table[0]["CSR"] == 0.5 and table[1]["SoC"] == 160.0 and "runtime" not in table[0]
    assert len(lines) == 3 and lines[0].split()[:2] == ["num_agents", "algorithm"]


def test_plot_view_series_and_file(tmp_path):
    """`type: plot` (eval_configs/01-random/01-random.yaml:162-186): mean of y per x and algorithm over seeds/maps."""
    res = []
    for algo, base in (("A", 10.0), ("B", 20.0)):
        for n in (8, 16, 32):
            for seed in (0, 1, 2, 3):
                res.append({"metrics": {"SoC": base * n + seed, "CSR": 1.0}, "env_grid_search": {"seed": seed, "num_agents": n, "map_name": "m"},
                            "algorithm": algo})
    view = {"type": "plot", "x": "num_agents", "y": "SoC", "width": 3.0, "height": 2.5, "line_width": 2, "use_log_scale_x": True,
            "legend_font_size": 8, "font_size": 8, "name": "Random / Mazes", "ticks": [8, 16, 32]}
    series = ev.plot_series(res, view)
    assert list(series) == ["A", "B"] and [p[0] for p in series["A"]] == [8, 16, 32]
    x, mean, lo, hi, n = series["B"][1]
    assert n == 4 and mean == 20.0 * 16 + 1.5 and lo < mean < hi
    assert np.isclose(hi - mean, 1.96 * np.std([0, 1, 2, 3], ddof=1) / 2)
    out = tmp_path / "v.pdf"
    assert ev.plot_view(res, view, str(out)) == series
    assert out.stat().st_size > 1000
