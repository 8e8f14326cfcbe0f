"""Evaluation harness: the role of the reference's `benchmark.py` + `pogema_toolbox.evaluator.evaluation`
(benchmark.py:20-50) for configs in the reference's YAML format (eval_configs/*/*.yaml + maps.yaml).

What the reference does per folder: register the maps, expand every `grid_search` list of the
`environment:` block into the cartesian product of runs, run every algorithm of `algorithms:` on every
run (one episode per worker process, dask pool of `num_process`), collect the episode metrics
(create_env.py:15-19: infos[0]["metrics"]) and render `results_views`.

Here all runs of one (algorithm, num_agents) group become instances of ONE device-resident batch
(`BatchedRunner`): maps are padded to a common frame (extra obstacle cells change nothing for the
agents), starts/goals are drawn per (map, seed), and the whole group steps together on the GPU.
With `torch.distributed` initialised the instances of a group are sharded over the ranks and the
metric records are gathered once (runner.gather_metrics).

Result records keep the toolbox's shape: {"metrics": {...}, "env_grid_search": {...}, "algorithm": name}.
Metric keys: CSR, ISR, SoC, makespan, ep_length (env spec: DESIGN.md section 4; `avg_throughput` for
on_target="restart" runs) and `runtime`
(seconds; the group's wall time divided by its instances -- a batched run has no per-episode clock).
Placement is this repo's seeded generator, not POGEMA's (absent offline): numbers are comparable to
the paper's only in distribution, not seed for seed.
"""
import itertools
import json
import os
import time
from collections import OrderedDict

import numpy as np

from . import maps as _maps

METRIC_KEYS = ("CSR", "ISR", "SoC", "makespan", "ep_length", "avg_agents_density")
LIFELONG_QUEUE = 64          # goals pre-generated per agent in on_target="restart" runs (the queue wraps)


# ---- config handling (pure python, testable without a GPU) -------------------------------------
def expand_grid_search(env_cfg):
    """`environment:` block -> list of (full_cfg, grid_point) in the product order of the keys as written."""
    keys = [k for k, v in env_cfg.items() if isinstance(v, dict) and "grid_search" in v]
    lists = [list(env_cfg[k]["grid_search"]) for k in keys]
    base = {k: v for k, v in env_cfg.items() if k not in keys}
    out = []
    for combo in itertools.product(*lists):
        point = OrderedDict(zip(keys, combo))
        cfg = dict(base)
        cfg.update(point)
        out.append((cfg, dict(point)))
    return out


def load_yaml(path):
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)


class MapRegistry:
    """name -> map string; `register_maps` mirrors ToolboxRegistry.register_maps (benchmark.py:38-41)."""

    def __init__(self):
        self._maps = dict(_maps.named_maps())

    def register_maps(self, maps):
        self._maps.update(maps)

    def get(self, name):
        if name not in self._maps:
            raise KeyError(f"map '{name}' is not registered (register_maps(yaml of eval_configs/<folder>/maps.yaml))")
        return _maps.parse_map(self._maps[name])


def common_frame(parsed, r=_maps.OBS_RADIUS):
    """Pad every (obst, start_ok, goal_ok) to one (H, W): r obstacle cells on every side, then obstacle
    fill to the largest map (top-left aligned)."""
    H = max(p[0].shape[0] for p in parsed) + 2 * r
    W = max(p[0].shape[1] for p in parsed) + 2 * r
    out = []
    for obst, s_ok, g_ok in parsed:
        g = np.ones((H, W), np.uint8)
        s = np.zeros((H, W), bool)
        t = np.zeros((H, W), bool)
        h, w = obst.shape
        g[r:r + h, r:r + w] = obst
        s[r:r + h, r:r + w] = s_ok
        t[r:r + h, r:r + w] = g_ok
        out.append((g, s, t))
    return out


def group_runs(runs):
    """Runs that can share one batch: same num_agents, max_episode_steps and on_target."""
    groups = OrderedDict()
    for i, (cfg, point) in enumerate(runs):
        key = (int(cfg["num_agents"]), int(cfg.get("max_episode_steps", 128)), cfg.get("on_target", "nothing"))
        groups.setdefault(key, []).append(i)
    return groups


def tabular_view(results, view_cfg, print_fn=print):
    """`type: tabular` of results_views: drop `drop_keys`, group by the remaining grid keys + algorithm, mean."""
    drop = set(view_cfg.get("drop_keys", []))
    digits = int(view_cfg.get("round_digits", 2))
    rows = OrderedDict()
    for r in results:
        key = tuple((k, v) for k, v in r["env_grid_search"].items() if k not in drop) + (("algorithm", r["algorithm"]),)
        rows.setdefault(key, []).append(r["metrics"])
    present = set().union(*[set(r["metrics"]) for r in results]) if results else set()
    cols = [k for k in list(METRIC_KEYS) + ["avg_throughput", "runtime"] if k not in drop and k in present]
    table = []
    for key, ms in rows.items():
        rec = OrderedDict(key)
        for c in cols:
            rec[c] = round(float(np.mean([m[c] for m in ms])), digits)
        table.append(rec)
    if table and view_cfg.get("print_results", False):
        hdr = list(table[0].keys())
        wid = [max(len(str(h)), max(len(str(t[h])) for t in table)) for h in hdr]
        print_fn("  ".join(str(h).ljust(w) for h, w in zip(hdr, wid)))
        for t in table:
            print_fn("  ".join(str(t[h]).ljust(w) for h, w in zip(hdr, wid)))
    return table


def plot_series(results, view_cfg):
    """Data of a `type: plot` view (eval_configs/01-random/01-random.yaml:162-186): per algorithm, the mean of metric
    `y` at every value of grid key `x` over all other grid keys (seeds, maps), with a normal-approximation 95 % interval
    of that mean -> {algorithm: [(x, mean, lo, hi, n), ...]} sorted by x."""
    xk, yk = view_cfg["x"], view_cfg["y"]
    acc = OrderedDict()
    for r in results:
        if xk not in r["env_grid_search"] or yk not in r["metrics"]:
            continue
        acc.setdefault(r["algorithm"], OrderedDict()).setdefault(r["env_grid_search"][xk], []).append(float(r["metrics"][yk]))
    out = OrderedDict()
    for algo, by_x in acc.items():
        pts = []
        for x in sorted(by_x):
            v = np.asarray(by_x[x], dtype=np.float64)
            half = 1.96 * v.std(ddof=1) / np.sqrt(len(v)) if len(v) > 1 else 0.0
            pts.append((x, float(v.mean()), float(v.mean() - half), float(v.mean() + half), len(v)))
        out[algo] = pts
    return out


def plot_view(results, view_cfg, path):
    """`type: plot` of results_views: one line per algorithm, metric `y` against grid key `x` with the interval band;
    honours width/height (inches), line_width, use_log_scale_x, ticks, font_size, legend_font_size, name (title).
    Writes `path` (format by extension) and returns the plotted series."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    series = plot_series(results, view_cfg)
    fs = view_cfg.get("font_size", 8)
    fig, ax = plt.subplots(figsize=(float(view_cfg.get("width", 3.0)), float(view_cfg.get("height", 2.5))))
    for algo, pts in series.items():
        xs = [p[0] for p in pts]
        line, = ax.plot(xs, [p[1] for p in pts], marker="o", markersize=3, linewidth=float(view_cfg.get("line_width", 2)), label=algo)
        ax.fill_between(xs, [p[2] for p in pts], [p[3] for p in pts], alpha=0.2, color=line.get_color(), linewidth=0)
    if view_cfg.get("use_log_scale_x", False):
        ax.set_xscale("log")
    if view_cfg.get("use_log_scale_y", False):
        ax.set_yscale("log")
    if view_cfg.get("ticks"):
        ax.set_xticks(list(view_cfg["ticks"]))
        ax.set_xticklabels([str(t) for t in view_cfg["ticks"]])
        ax.minorticks_off()
    ax.set_xlabel(str(view_cfg["x"]), fontsize=fs)
    ax.set_ylabel(str(view_cfg["y"]), fontsize=fs)
    ax.tick_params(labelsize=fs)
    if view_cfg.get("name"):
        ax.set_title(str(view_cfg["name"]), fontsize=fs)
    if series:
        ax.legend(fontsize=view_cfg.get("legend_font_size", fs))
    ax.grid(True, alpha=0.3)
    fig.tight_layout()
    fig.savefig(path)
    plt.close(fig)
    return series


# ---- the batched evaluation (GPU) -----------------------------------------------------------------
def _build_algorithm(algo_cfg, max_rows):
    from .inference import MAPFGPTInference, MAPFGPTInferenceConfig
    cfg = MAPFGPTInferenceConfig(**algo_cfg)                   # unknown keys raise, as in the reference (extra=forbid)
    cfg.batch_size = max(int(cfg.batch_size), 1)
    algo = MAPFGPTInference(cfg)
    return algo, cfg


def evaluation(evaluation_config, eval_dir=None, registry=None, precision=None, max_rows_per_batch=65536, rank=0, world=1,
               print_fn=print, trace=None):
    """Run `evaluation_config` (dict in the reference's YAML schema).  Returns the list of result records (on every
    rank); writes `<eval_dir>/<algorithm>.json` and prints the tabular views on rank 0.
    `trace(kind, payload)` (tests): called with ("reset", {algorithm, runs, grids, pos, goal, max_steps}) for every batch
    this rank runs and with ("step", int32 actions [instances, agents]) after every step -- the device's sampled actions,
    from which an episode can be replayed on the host."""
    import torch
    from .runner import BatchedRunner, gather_metrics, shard_range

    registry = registry or MapRegistry()
    runs = expand_grid_search(evaluation_config["environment"])
    groups = group_runs(runs)
    results = []
    for algo_name, algo_cfg in evaluation_config["algorithms"].items():
        algo_cfg = dict(algo_cfg)
        if precision is not None:
            algo_cfg["precision"] = precision
        algo, cfg = _build_algorithm(algo_cfg, max_rows_per_batch)
        for (n_agents, max_steps, on_target), idxs in groups.items():
            if on_target not in ("nothing", "restart"):
                raise NotImplementedError(f"on_target={on_target!r}: 'nothing' and 'restart' (lifelong) are built")
            parsed = [registry.get(runs[i][0]["map_name"]) for i in idxs]
            frames = dict(zip(idxs, common_frame(parsed)))
            per_batch = max(1, max_rows_per_batch // n_agents)
            for b0 in range(0, len(idxs), per_batch):
                chunk = idxs[b0:b0 + per_batch]
                lo, hi = shard_range(len(chunk), rank, world)
                mine = chunk[lo:hi]
                local = torch.zeros((0, len(METRIC_KEYS)), dtype=torch.float32, device=cfg.device)
                t0 = time.perf_counter()
                if mine:
                    grids = np.stack([frames[i][0] for i in mine])
                    pos = np.empty((len(mine), n_agents, 2), np.int16)
                    goal = np.empty((len(mine), n_agents, 2), np.int16)
                    for k, i in enumerate(mine):
                        g, s_ok, g_ok = frames[i]
                        pos[k], goal[k] = _maps.place_agents(g, n_agents, int(runs[i][0].get("seed", 0)), s_ok, g_ok)
                    run = BatchedRunner(grids, len(mine), n_agents, algo.net, max_episode_steps=max_steps,
                                        seed=int(cfg.seed or 0), do_sample=True, precision=cfg.precision, device=cfg.device,
                                        row_offset=lo * n_agents)
                    queue = None
                    if on_target == "restart":          # lifelong: a seeded queue of further goals per agent (wraps)
                        queue = np.empty((len(mine), n_agents, LIFELONG_QUEUE, 2), np.int16)
                        for k, i in enumerate(mine):
                            g, s_ok, g_ok = frames[i]
                            cells = np.argwhere(_maps.largest_component(g == 0) & g_ok)
                            rng = np.random.Generator(np.random.PCG64([int(runs[i][0].get("seed", 0)), 0x4C4C]))
                            queue[k] = cells[rng.integers(0, len(cells), (n_agents, LIFELONG_QUEUE))]
                        queue = torch.from_numpy(queue)
                    run.reset(torch.from_numpy(pos), torch.from_numpy(goal), goal_queue=queue)
                    if trace is None:
                        run.run(max_steps)
                    else:
                        trace("reset", {"algorithm": algo_name, "runs": list(mine), "grids": grids.copy(), "pos": pos.copy(),
                                        "goal": goal.copy(), "max_steps": max_steps, "on_target": on_target})
                        for _ in range(max_steps):
                            run.step()
                            trace("step", run.actions.cpu().numpy().copy())
                    local = run.metrics().to(torch.float32)
                    if queue is not None:               # ISR column carries the throughput (arrivals per step) in lifelong runs
                        local[:, 1] = run.env.goals_reached().sum(1).to(torch.float32) / float(max_steps)
                    torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                allm = gather_metrics(local, len(chunk), rank, world).cpu().numpy()
                for k, i in enumerate(chunk):
                    m = {key: float(allm[k, j]) for j, key in enumerate(METRIC_KEYS)}
                    if on_target == "restart":
                        m = {"avg_throughput": m["ISR"], "ep_length": m["ep_length"]}
                    m["runtime"] = dt / max(1, len(chunk))
                    results.append({"metrics": m, "env_grid_search": runs[i][1], "algorithm": algo_name})
        del algo
    if rank == 0:
        if eval_dir is not None:
            os.makedirs(eval_dir, exist_ok=True)
            for algo_name in evaluation_config["algorithms"]:
                with open(os.path.join(eval_dir, f"{algo_name}.json"), "w") as f:
                    json.dump([r for r in results if r["algorithm"] == algo_name], f, indent=1)
        for view_name, view in (evaluation_config.get("results_views") or {}).items():
            if view.get("type") == "tabular":
                print_fn(f"== {view_name}")
                tabular_view(results, view, print_fn)
            elif view.get("type") == "plot" and eval_dir is not None:
                plot_view(results, view, os.path.join(eval_dir, f"{view_name}.pdf"))
    return results


def run_folder(folder, eval_root="eval_configs", **kw):
    """= one iteration of benchmark.py:37-50 on a folder laid out like the reference's eval_configs/<folder>/."""
    reg = MapRegistry()
    maps_path = os.path.join(eval_root, folder, "maps.yaml")
    if os.path.exists(maps_path):
        reg.register_maps(load_yaml(maps_path))
    cfg = load_yaml(os.path.join(eval_root, folder, f"{os.path.basename(folder)}.yaml"))
    return evaluation(cfg, eval_dir=os.path.join(eval_root, folder), registry=reg, **kw)
