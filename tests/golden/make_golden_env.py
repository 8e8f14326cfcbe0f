#!/usr/bin/env python3
"""Env-step fixtures captured from POGEMA itself -- runs only where `import pogema` succeeds (it does NOT in the build
container or on the GPU box: pogema is an un-vendored pip dependency of the reference, pyproject.toml:18).

What it captures, with the configuration the reference uses (experiment_setup/create_env.py:36-46, example.py:41-50:
collision_system="soft", observation_type="MAPF", obs_radius=5, on_target="nothing" and "restart"):
    grid      uint8 [H, W]      the PADDED obstacle map the env reports (obs["global_obstacles"])
    pos       int32 [T+1, n, 2] obs["global_xy"] after reset and after every step (padded coordinates)
    goal      int32 [T+1, n, 2] obs["global_target_xy"] (changes in "restart" mode)
    actions   int32 [T, n]      what was passed to env.step (seeded; every third step all agents push one way, which
                                provokes chains, swaps and contested cells -- the places where our spec is RECALLED)
    terminated / truncated  uint8 [T, n]
    metrics   the final infos[0]["metrics"] dict (CSR, ISR, SoC, makespan, ep_length, avg_agents_density) as JSON
per scenario, into tests/golden/env_pogema_<scenario>.npz.  tests/test_env_pogema_fixtures.py (skipped while these files
are absent) replays them through the oracle with every rule mask and fails unless the DEFAULT mask reproduces POGEMA --
naming the mask that does, so that pinning is `mgpt_env_set_rules(mask)` + a change of default, not a rewrite.

Run (anywhere pogema 2.x and this repo are importable):   python tests/golden/make_golden_env.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

SCENARIOS = [
    # name, map (None = pogema's own random generator), size, density, agents, steps, on_target
    ("random_dense", None, 16, 0.2, 48, 48, "nothing"),
    ("random_sparse", None, 24, 0.1, 24, 64, "nothing"),
    ("corridor", "corridor", 0, 0.0, 6, 32, "nothing"),
    ("restart", None, 16, 0.2, 24, 64, "restart"),
]
CORRIDOR = "\n".join(["#########", "#.......#", "#########"])      # one lane: every move is a chain, a swap or a contest


def capture(name, map_str, size, density, n, steps, on_target):
    from pogema import GridConfig, pogema_v0
    kw = dict(num_agents=n, obs_radius=5, max_episode_steps=steps, seed=7, collision_system="soft", observation_type="MAPF",
              on_target=on_target)
    if map_str is None:
        kw.update(size=size, density=density)
    else:
        kw.update(map=map_str)
    env = pogema_v0(grid_config=GridConfig(**kw))
    obs, _ = env.reset()
    rng = np.random.Generator(np.random.PCG64(13))
    grid = (np.asarray(obs[0]["global_obstacles"]).astype(int) != 0).astype(np.uint8)
    pos = [np.array([o["global_xy"] for o in obs], np.int32)]
    goal = [np.array([o["global_target_xy"] for o in obs], np.int32)]
    acts, term, trunc, metrics = [], [], [], {}
    for t in range(steps):
        a = rng.integers(0, 5, n).astype(np.int32)
        if t % 3 == 0:
            a[:] = rng.integers(1, 5)
        obs, _, te, tr, infos = env.step(a.tolist())
        acts.append(a); term.append(np.array(te, np.uint8)); trunc.append(np.array(tr, np.uint8))
        pos.append(np.array([o["global_xy"] for o in obs], np.int32))
        goal.append(np.array([o["global_target_xy"] for o in obs], np.int32))
        if all(te) or all(tr):
            metrics = dict(infos[0].get("metrics", {}))
            break
    np.savez_compressed(os.path.join(OUT, f"env_pogema_{name}.npz"), grid=grid, pos=np.stack(pos), goal=np.stack(goal),
                        actions=np.stack(acts), terminated=np.stack(term), truncated=np.stack(trunc), on_target=np.array(on_target),
                        metrics=np.array(json.dumps({k: float(v) for k, v in metrics.items()})))
    print(f"env_pogema_{name}: {len(acts)} steps, {n} agents, grid {grid.shape}, metrics {metrics}")


def main():
    try:
        import pogema  # noqa: F401
    except ImportError as e:
        print(f"pogema is not importable here ({e}); nothing captured -- env parity stays unpinned", file=sys.stderr)
        return 1
    for name, m, size, dens, n, steps, ot in SCENARIOS:
        capture(name, CORRIDOR if m == "corridor" else None, size, dens, n, steps, ot)
    return 0


if __name__ == "__main__":
    sys.exit(main())
