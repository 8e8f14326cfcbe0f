#!/usr/bin/env python3
"""Golden vectors for NON-DEFAULT InputParameters (observation_generator.h:22-40), from the REAL reference (build container only).

The compiled reference tokenizer (oracle/_ref, `make -C oracle ref`) accepts any InputParameters (pybind ctor
observation_generator.cpp:551); inference.py only ever passes (20, 13, 5, 256, 5, 5).  Every case here fits a 256-token row
((2 obs_radius + 1)^2 + num_agents (5 + num_previous_actions) <= 256) and keeps agents_radius <= cost2go_value_limit (beyond
that the reference throws in int_vocab.at, cpp:358-359).  Inputs: the seeded walks of make_golden.py.
Run:  python tests/golden/make_golden_params.py      -> tests/golden/tokp_*.npz
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
OUT = os.path.join(ROOT, "tests", "golden")

from make_golden import import_reference, walk  # noqa: E402
from mapf_gpt_amd import maps  # noqa: E402

# tag: (map, agents, steps, goal change at, (limit, num_agents, previous actions, obs radius, agents radius), rows kept)
CASES = {
    "l10_s8_h3_r4_a4": ("validation-mazes-seed-000", 64, 16, 8, (10, 8, 3, 4, 4), None),
    "l20_s13_h5_r5_a3": ("validation-random-seed-000", 32, 16, None, (20, 13, 5, 5, 3), None),
    "l20_s16_h2_r3_a2": ("wfi_warehouse", 192, 8, None, (20, 16, 2, 3, 2), 48),                 # 49-cell window, 7-token records, 16 slots, > 64 agents
    "l30_s10_h0_r5_a5": ("validation-mazes-seed-000", 64, 10, 5, (30, 10, 0, 5, 5), None),      # no history, another vocabulary
    "l5_s4_h5_r2_a5": ("validation-random-seed-000", 32, 12, None, (5, 4, 5, 2, 5), None),      # agents radius = limit, 25-cell window
    "l20_s9_h4_r5_a5": ("Berlin_1_256_00", 256, 6, 3, (20, 9, 4, 5, 5), 48),                    # 9-token records (odd alignment), 256 agents
}


def main():
    og, _, _ = import_reference()
    for tag, (name, n, steps, gc, (L, S, Hn, R, A), nkeep) in CASES.items():
        grid, s_ok, g_ok = maps.load_named(name)
        pos, goal = maps.place_agents(grid, n, seed=3, start_ok=s_ok, goal_ok=g_ok)
        gen = og.ObservationGenerator(grid.astype(int).tolist(), og.InputParameters(L, S, Hn, 256, R, A, 64, False))
        P, G, Ac, T = [], [], [], []
        for t, (p, g, a) in enumerate(walk(grid, pos, goal, steps, 300 + len(tag), gc, g_ok)):
            pl, gl = [tuple(map(int, x)) for x in p], [tuple(map(int, x)) for x in g]
            if t == 0:
                gen.create_agents(pl, gl)
            gen.update_agents(pl, gl, [int(x) for x in a])
            tok = np.array(gen.generate_observations(), dtype=np.int64)
            assert tok.shape == (n, 256) and tok.min() >= 0 and tok.max() <= 2 * L + 26, (tok.shape, tok.min(), tok.max())
            P.append(p); G.append(g); Ac.append(a); T.append(tok.astype(np.uint8))
        P, G, Ac, T = np.array(P, np.int16), np.array(G, np.int16), np.array(Ac, np.int8), np.array(T)
        sha = hashlib.sha256(T.tobytes()).hexdigest()
        keep = np.arange(n) if nkeep is None else np.sort(np.random.Generator(np.random.PCG64(2)).permutation(n)[:nkeep])
        np.savez_compressed(os.path.join(OUT, f"tokp_{tag}.npz"), grid=grid.astype(np.uint8), pos=P, goal=G, actions=Ac,
                            tokens=T[:, keep], keep=keep, sha256_all_rows=np.array(sha),
                            params=np.array([L, S, Hn, 256, R, A], np.int32))
        print(f"tokp_{tag}: grid {grid.shape} agents {n} steps {steps} params {(L, S, Hn, R, A)} sha {sha[:16]}")


if __name__ == "__main__":
    main()
