#!/usr/bin/env python3
"""Golden vectors for the DATASET-side tokenizer, made by the reference's own code
(dataset/tokenizer/generate_observations.py with its two cppimport modules compiled by g++ into /tmp; nothing of the
reference is copied into this repo).  Run in the container that has /root/reference:

    python tests/golden/make_golden_dataset.py

Logs are synthetic: this repo's map generators and oracle env step produce collision-free executed paths; the
record layout is the reference's ({"metrics": {"CSR", "made_actions", "init_positions"}, "env_grid_search": {"map_name"}}).
"""
import os, subprocess, sys, sysconfig, types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/dataset/tokenizer"
TMP = "/tmp/mgpt_dsref"


def build_reference():
    os.makedirs(os.path.join(TMP, "tokenizer"), exist_ok=True)
    inc = subprocess.check_output([sys.executable, "-m", "pybind11", "--includes"], text=True).split()
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    for m in ("cost2go", "encoder"):
        src = os.path.join(TMP, m + ".cpp")
        with open(os.path.join(REF, m + ".cpp")) as f, open(src, "w") as g:      # drop the cppimport trailer
            skip = False
            for line in f:
                if line.startswith("<%"): skip = True
                if not skip: g.write(line)
                if line.startswith("%>"): skip = False
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC"] + inc + [src, "-o", os.path.join(TMP, "tokenizer", m + suffix)])
    with open(os.path.join(TMP, "tokenizer", "__init__.py"), "w") as f:
        f.write(f"__path__.append({REF!r})\n")
    sys.path.insert(0, TMP)
    sys.modules["cppimport"] = types.ModuleType("cppimport")
    sys.modules["cppimport.import_hook"] = types.ModuleType("cppimport.import_hook")


def to_str(obst):
    return "\n".join("".join("#" if v else "." for v in row) for row in obst)


def synth_log(obst, n_agents, steps, seed):
    from mapf_gpt_amd import maps
    from oracle import oracle as orc
    grid = maps.pad(obst)
    pos, goal = maps.place_agents(grid, n_agents, seed)
    pos, goal = pos.astype(np.int32), goal.astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(seed))
    init = pos.copy()
    made = [[] for _ in range(n_agents)]
    d = [orc.bfs(grid, g) for g in goal]
    for t in range(steps):
        want = np.zeros(n_agents, np.int32)
        for a in range(n_agents):                     # mostly greedy towards the goal, sometimes random / wait
            if rng.random() < 0.25:
                want[a] = rng.integers(0, 5)
            else:
                best, bd = 0, int(d[a][pos[a][0], pos[a][1]])
                for k, (dr, dc) in enumerate([(-1, 0), (1, 0), (0, -1), (0, 1)], start=1):
                    v = int(d[a][pos[a][0] + dr, pos[a][1] + dc])
                    if v < bd: best, bd = k, v
                want[a] = best
        new, _ = orc.env_step(grid, pos, goal, want)
        for a in range(n_agents):
            dr, dc = int(new[a][0] - pos[a][0]), int(new[a][1] - pos[a][1])
            made[a].append({(0, 0): 0, (-1, 0): 1, (1, 0): 2, (0, -1): 3, (0, 1): 4}[(dr, dc)])
        pos = new
    return grid, init, made


def synth_lifelong_log(obst, n_agents, steps, seed, n_targets=24):
    """A lifelong log: every agent walks greedily (with some noise) to the next target of its own list; the list is what
    LogActions stores as global_lifelong_targets_xy (create_env.py:28-32).  Padded coordinates, like init_positions."""
    from mapf_gpt_amd import maps
    from oracle import oracle as orc
    grid = maps.pad(obst)
    pos, _ = maps.place_agents(grid, n_agents, seed)
    pos = pos.astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    comp = maps.largest_component(grid == 0)
    free = np.argwhere(comp)
    targets = np.zeros((n_agents, n_targets, 2), np.int32)
    for a in range(n_agents):
        prev = pos[a]
        for k in range(n_targets):                    # near targets so that several are reached within the log
            d = np.abs(free - prev).sum(1)
            near = free[(d >= 1) & (d <= 4)]
            targets[a, k] = near[rng.integers(0, len(near))]
            prev = targets[a, k]
    cur = np.zeros(n_agents, int)
    init = pos.copy()
    made = [[] for _ in range(n_agents)]
    fields = {}
    for t in range(steps):
        want = np.zeros(n_agents, np.int32)
        for a in range(n_agents):
            g = tuple(int(v) for v in targets[a, cur[a]])
            if g not in fields:
                fields[g] = orc.bfs(grid, np.array(g))
            d = fields[g]
            if rng.random() < 0.2:
                want[a] = rng.integers(0, 5)
            else:
                best, bd = 0, int(d[pos[a][0], pos[a][1]])
                for k, (dr, dc) in enumerate([(-1, 0), (1, 0), (0, -1), (0, 1)], start=1):
                    v = int(d[pos[a][0] + dr, pos[a][1] + dc])
                    if v < bd: best, bd = k, v
                want[a] = best
        new, _ = orc.env_step(grid, pos, np.full_like(pos, -1), want)
        for a in range(n_agents):
            dr, dc = int(new[a][0] - pos[a][0]), int(new[a][1] - pos[a][1])
            made[a].append({(0, 0): 0, (-1, 0): 1, (1, 0): 2, (0, -1): 3, (0, 1): 4}[(dr, dc)])
        pos = new
        for a in range(n_agents):                     # the same rule get_goal_positions applies when it replays the path
            if tuple(pos[a]) == tuple(targets[a, cur[a]]):
                cur[a] += 1
    assert cur.max() < n_targets - 2 and cur.max() >= 2, cur
    return grid, init, made, targets


def main():
    build_reference()
    from tokenizer.generate_observations import ObservationGenerator
    from tokenizer.parameters import InputParameters
    from mapf_gpt_amd import maps
    cases = {"ds_random": (maps.random_map(18, 20, 0.15, 3), 14, 11, 1), "ds_maze": (maps.maze_map(17, 17, 4), 22, 9, 2),
             "ds_short": (maps.random_map(12, 12, 0.1, 9), 5, 3, 5)}
    for name, (obst, n, steps, seed) in cases.items():
        grid, init, made = synth_log(obst, n, steps, seed)
        data = [{"metrics": {"CSR": 1.0, "made_actions": [list(m) for m in made], "init_positions": [[int(p[0]), int(p[1])] for p in init]},
                 "env_grid_search": {"map_name": "m"}}]
        gen = ObservationGenerator({"m": to_str(obst)}, data, InputParameters())
        inputs, gts = gen.generate_observations(0, 1)
        out = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(out, grid=grid.astype(np.uint8), init_positions=init.astype(np.int16), made_actions=np.array(made, np.int8),
                            inputs=np.stack(inputs).astype(np.int8), gt_actions=np.array(gts, np.int8))
        print(name, np.stack(inputs).shape, "->", out)
    # the mask_cost2go ablation on one of the logs above (parameters.py:14, generate_observations.py:253-262)
    obst, n, steps, seed = cases["ds_random"]
    grid, init, made = synth_log(obst, n, steps, seed)
    data = [{"metrics": {"CSR": 1.0, "made_actions": [list(m) for m in made], "init_positions": [[int(p[0]), int(p[1])] for p in init]},
             "env_grid_search": {"map_name": "m"}}]
    inputs, gts = ObservationGenerator({"m": to_str(obst)}, data, InputParameters(mask_cost2go=True)).generate_observations(0, 1)
    out = os.path.join(ROOT, "tests", "golden", "ds_maskc2g.npz")
    np.savez_compressed(out, grid=grid.astype(np.uint8), init_positions=init.astype(np.int16), made_actions=np.array(made, np.int8),
                        inputs=np.stack(inputs).astype(np.int8), gt_actions=np.array(gts, np.int8), mask_cost2go=np.array(1))
    print("ds_maskc2g", np.stack(inputs).shape, "->", out)
    # the python Encoder's mask_* ablations (dataset/tokenizer/tokenizer.py:104-138) on rows of ds_random
    sys.path.insert(0, "/root/reference")
    from dataset.tokenizer.parameters import InputParameters as PyParams
    from dataset.tokenizer.tokenizer import Encoder as PyEncoder
    rows = np.load(os.path.join(ROOT, "tests", "golden", "ds_random.npz"))["inputs"][::9]
    masked = {}
    for flag in ("mask_actions_history", "mask_goal", "mask_greed_action", "mask_cost2go"):
        enc = PyEncoder(PyParams(**{flag: True}))
        masked[flag] = np.array([enc.mask([int(v) for v in r]) for r in rows], dtype=np.int8)
    enc = PyEncoder(PyParams(mask_actions_history=True, mask_goal=True, mask_greed_action=True, mask_cost2go=True))
    masked["all"] = np.array([enc.mask([int(v) for v in r]) for r in rows], dtype=np.int8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "enc_masks.npz"), rows=rows, **masked)
    print("enc_masks", rows.shape)
    # lifelong logs (generate_observations.py:55-60, 143-153)
    for name, (obst, n, steps, seed) in {"ds_lifelong": (maps.random_map(16, 18, 0.12, 21), 12, 14, 3),
                                         "ds_lifelong_maze": (maps.maze_map(15, 15, 8), 9, 12, 6)}.items():
        grid, init, made, targets = synth_lifelong_log(obst, n, steps, seed)
        data = [{"metrics": {"made_actions": [list(m) for m in made], "init_positions": [[int(p[0]), int(p[1])] for p in init],
                             "global_lifelong_targets_xy": [[[int(x), int(y)] for x, y in tg] for tg in targets]},
                 "env_grid_search": {"map_name": "m"}}]
        inputs, gts = ObservationGenerator({"m": to_str(obst)}, data, InputParameters()).generate_observations(0, 1)
        out = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(out, grid=grid.astype(np.uint8), init_positions=init.astype(np.int16), made_actions=np.array(made, np.int8),
                            inputs=np.stack(inputs).astype(np.int8), gt_actions=np.array(gts, np.int8),
                            lifelong_targets=targets.astype(np.int16))
        print(name, np.stack(inputs).shape, "->", out)


if __name__ == "__main__":
    main()
