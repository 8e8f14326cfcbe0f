#!/usr/bin/env python3
"""tools/fuzz_env.py [seconds] [seed] -- soak of env_step_kernel against the C restatement of the env spec (oracle/, the checker; the spec itself is
parity-unpinned against POGEMA) on random cases the fixed test list does not enumerate: map sizes 8 .. 120, obstacle densities
0 .. 0.4, 1 .. 256 agents up to half of the free cells (corridors full of agents: chains, swaps, contested cells), every rule mask 0 .. 3, action
mixtures (uniform, everybody the same way, two opposed streams, out-of-range ids), several instances per case on one map.  Every step also checks the
invariants any MAPF step must keep: no vertex conflict, nobody on an obstacle, no edge swap, moves of at most one cell.  Not part of the test suite
(unbounded by design); a mismatch prints the case and exits 1."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mapf_gpt_amd import maps  # noqa: E402
from mapf_gpt_amd.env import BatchedEnv  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def draw_actions(rng, n_inst, n, mode):
    if mode == 0:
        return rng.integers(0, 5, (n_inst, n)).astype(np.int32)
    if mode == 1:                                            # everybody pushes the same way: chains against walls
        return np.full((n_inst, n), int(rng.integers(1, 5)), np.int32)
    if mode == 2:                                            # two opposed streams: swaps and contested cells
        a, b = ((1, 2), (3, 4))[int(rng.integers(0, 2))]
        act = np.where(rng.random((n_inst, n)) < 0.5, a, b).astype(np.int32)
        return act
    act = rng.integers(0, 5, (n_inst, n)).astype(np.int32)   # a few ids outside 0 .. 4 (treated as "wait")
    bad = rng.random((n_inst, n)) < 0.05
    act[bad] = rng.choice(np.array([-1, 5, 7, 100], np.int32), int(bad.sum()))
    return act


def one_case(rng, idx):
    h, w = int(rng.integers(8, 121)), int(rng.integers(8, 121))
    if rng.random() < 0.6:
        h, w = min(h, 32), min(w, 32)                        # small maps: crowded
    dens = float(rng.uniform(0.0, 0.4))
    kind = "maze" if rng.random() < 0.3 else "random"
    raw = maps.maze_map(h, w, int(rng.integers(1 << 30))) if kind == "maze" else maps.random_map(h, w, dens, int(rng.integers(1 << 30)))
    grid = maps.pad(raw)
    free = int(maps.largest_component(grid == 0).sum())
    n = int(min(rng.integers(1, 257), max(1, free // 2)))
    n_inst = int(rng.integers(1, 5))
    rules = int(rng.integers(0, 4))
    steps = int(rng.integers(8, 40))
    desc = f"case {idx}: {kind} {h}x{w} dens {dens:.2f} agents {n} instances {n_inst} rules {rules} steps {steps}"
    try:
        placed = [maps.place_agents(grid, n, seed=int(rng.integers(1 << 30))) for _ in range(n_inst)]
    except ValueError:
        return None
    p = np.stack([x[0] for x in placed]).astype(np.int32)
    g = np.stack([x[1] for x in placed]).astype(np.int32)
    env = BatchedEnv(grid, n_inst, n, max_episode_steps=10 ** 6)
    env.set_rules(rules)
    env.reset(torch.from_numpy(p.astype(np.int16)), torch.from_numpy(g.astype(np.int16)))
    was_done = np.zeros(n_inst, bool)
    for t in range(steps):
        act = draw_actions(rng, n_inst, n, int(rng.integers(0, 4)))
        env.step(torch.from_numpy(act).cuda())
        got, _, done = env.sync_state()
        got = got.cpu().numpy().astype(np.int32)
        done = done.cpu().numpy()
        for i in range(n_inst):
            if was_done[i]:                                  # everybody was on its goal: the instance is frozen
                if not np.array_equal(got[i], p[i]):
                    print("MOVED AFTER DONE", desc, "step", t, "instance", i)
                    return False
                continue
            exp, _ = orc.env_step(grid, p[i], g[i], act[i], rules=rules)
            if not np.array_equal(got[i], exp):
                bad = np.argwhere((got[i] != exp).any(1)).ravel()
                print("MISMATCH", desc, "step", t, "instance", i, "agents", bad[:8].tolist(), "got", got[i][bad[0]].tolist(), "want", exp[bad[0]].tolist())
                return False
            ok = len({tuple(x) for x in exp}) == n and (grid[exp[:, 0], exp[:, 1]] == 0).all() and (np.abs(exp - p[i]).sum(1) <= 1).all()
            old = {tuple(x): a for a, x in enumerate(p[i])}
            for a in range(n):
                b = old.get(tuple(exp[a]))
                if b is not None and b != a and tuple(exp[b]) == tuple(p[i][a]):
                    ok = False
            if not ok:
                print("INVARIANT BROKEN", desc, "step", t, "instance", i)
                return False
            p[i] = exp
        was_done = done != 0
    return True


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.Generator(np.random.PCG64(int(sys.argv[2]) if len(sys.argv) > 2 else 0))
    t0, done, skipped = time.time(), 0, 0
    while time.time() - t0 < budget:
        r = one_case(rng, done + skipped)
        if r is False:
            sys.exit(1)
        if r is None:
            skipped += 1
        else:
            done += 1
    print(f"fuzz_env: {done} cases (positions equal to the spec's restatement every step, invariants kept) in {time.time() - t0:.0f} s ({skipped} unplaceable draws skipped)")


if __name__ == "__main__":
    main()
