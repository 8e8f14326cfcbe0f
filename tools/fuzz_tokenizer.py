#!/usr/bin/env python3
"""tools/fuzz_tokenizer.py [seconds] [seed] -- soak of tokens_kernel against the C oracle (oracle/, the checker) on random cases the fixed test list
does not enumerate: map sizes 12 .. 150 (one-byte and 16-bit fields, grids beyond 64 and 128: the partial-window branches), obstacle densities, 1 .. 256
agents (every KP instance, ragged last chunks, > 64 neighbours), InputParameters drawn from the range mgpt_tokenizer_create accepts, grid_step 16 / 32 / 64,
goal changes mid-way, out-of-range action ids.  Not part of the test suite (unbounded by design); a mismatch prints the case and exits 1."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mapf_gpt_amd import maps  # noqa: E402
from mapf_gpt_amd.observation_generator import BatchedTokenizer, InputParameters  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def draw_params(rng):
    while True:
        L = int(rng.integers(1, 41)); S = int(rng.integers(1, 17)); Hn = int(rng.integers(0, 6)); R = int(rng.integers(1, 6))
        A = int(rng.integers(0, min(5, L) + 1))
        if (2 * R + 1) ** 2 + S * (5 + Hn) <= 256:
            return (L, S, Hn, R, A)


def one_case(rng, idx):
    h, w = int(rng.integers(12, 151)), int(rng.integers(12, 151))
    if rng.random() < 0.5:
        h, w = min(h, 48), min(w, 48)                       # small maps: crowded windows
    dens = float(rng.uniform(0.0, 0.35))
    grid = maps.pad(maps.random_map(h, w, dens, int(rng.integers(1 << 30))))
    comp = maps.largest_component(grid == 0)
    free = int(comp.sum())
    n = int(min(rng.integers(1, 257), max(1, free // 2)))
    params = None if rng.random() < 0.4 else draw_params(rng)
    grid_step = int(rng.choice([16, 32, 64]))
    try:
        pos, goal = maps.place_agents(grid, n, seed=int(rng.integers(1 << 30)))
    except ValueError:
        return None
    steps = int(rng.integers(2, 6))
    desc = f"case {idx}: {h}x{w} dens {dens:.2f} agents {n} params {params} grid_step {grid_step} steps {steps}"
    gen = orc.OracleGenerator(grid, grid_step=grid_step, params=params)
    cfg = InputParameters(grid_step=grid_step) if params is None else InputParameters(params[0], params[1], params[2], 256, params[3], params[4], grid_step, False)
    tok = BatchedTokenizer(grid, 1, n, cfg)
    last = np.full((n,), -1, np.int32)
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt).cuda()
    for t in range(steps):
        goals_changed = False
        if t > 0 and rng.random() < 0.3:                   # lifelong-style goal change for a third of the agents
            cells = np.argwhere(comp)
            who = rng.permutation(n)[: max(1, n // 3)]
            goal = goal.copy(); goal[who] = cells[rng.permutation(len(cells))[: len(who)]]
            goals_changed = True
        dp, dg, da = dev(pos[None], torch.int16), dev(goal[None], torch.int16), dev(last[None], torch.int32)
        if t == 0:
            tok.create_agents(dp, dg); gen.create_agents(pos, goal)
        tok.update_agents(dp, dg, da, goals_may_change=goals_changed)
        gen.update_agents(pos, goal, last)
        got = tok.generate_observations().cpu().numpy().reshape(n, 256)
        want = gen.generate_observations()
        if not np.array_equal(got, want):
            bad = np.argwhere(got != want)
            print("MISMATCH", desc, "step", t, "first at (row, token)", bad[0].tolist(), "got", int(got[tuple(bad[0])]), "want", int(want[tuple(bad[0])]), f"({len(bad)} tokens)")
            return False
        act = rng.integers(0, 5, (n,)).astype(np.int32)
        pos, _ = orc.env_step(grid, pos, goal, act)
        last = act.copy()
        if rng.random() < 0.2:
            last[rng.integers(0, n)] = int(rng.choice([-1, 5, 9]))      # ids outside 0..4 encode as 'n' (cpp:442-463)
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
    print(f"fuzz_tokenizer: {done} cases bit-exact against the oracle in {time.time() - t0:.0f} s ({skipped} unplaceable draws skipped)")


if __name__ == "__main__":
    main()
