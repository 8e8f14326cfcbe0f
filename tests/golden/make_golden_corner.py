#!/usr/bin/env python3
"""Golden rows for the one place where the reference's tiled cost-to-go is NOT the plain BFS distance (build container
only; needs /root/reference through oracle/_ref).

observation_generator.cpp:178-198 (get_cells_on_border) seeds every border cell of an agent's cached 129 x 129 partial
window with its exact distance except the (right, bottom) corner, which therefore gets min(in-window neighbours) + 1 from
the flood fill of cpp:200-286.  An agent that walks to (left + 123, top + 123) without a recompute (cpp:469-477) sees that
cell at window position (10, 10).  Cases: four corner sites of a 266 x 266 padded map with different obstacles around the
corner, several goals (beyond the corner: the reference's value is 2 too large; before it: exact; on it), and a walk
on / off the spot that exercises the recompute rule.  Inputs are positions fed to update_agents directly.
Run:  python tests/golden/make_golden_corner.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import import_reference, ref_params, tokenizer_case  # noqa: E402

SITES = [(128, 128), (128, 192), (192, 128), (192, 192)]          # corners of the windows with origin (0|64, 0|64)


def build_grid(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    g = (rng.random((256, 256)) < 0.08).astype(np.uint8)
    grid = np.ones((266, 266), np.uint8)
    grid[5:261, 5:261] = g
    for (cr, cc) in SITES:                                        # clear the neighbourhood of every site, then shape it
        grid[cr - 8: cr + 4, cc - 8: cc + 4] = 0
    grid[127, 192] = 1                                            # site 1: the upper neighbour of the corner is blocked
    grid[191, 128] = 1; grid[192, 127] = 1                        # site 2: both in-window neighbours blocked
    grid[192, 192] = 1                                            # site 3: the corner itself is blocked
    return grid


def main():
    from mapf_gpt_amd import maps
    og, _, _ = import_reference()
    grid = build_grid(5)
    starts = [(cr - 128 + 60, cc - 128 + 60) for cr, cc in SITES]     # origin = site - 128 in both coordinates
    spots = [(cr - 5, cc - 5) for cr, cc in SITES]                    # corner at window cell (10, 10)
    for s in starts + spots:
        grid[s[0], s[1]] = 0
    goal_sets = {"beyond": [(250, 250), (250, 255), (255, 250), (255, 255)], "before": [(10, 10), (10, 70), (70, 10), (70, 70)],
                 "on_corner": [SITES[0], (250, 250), (250, 250), (250, 250)]}
    free = grid == 0
    for tag, goals in goal_sets.items():
        goals = [g if free[g] else tuple(np.argwhere(free)[np.abs(np.argwhere(free) - np.array(g)).sum(1).argmin()]) for g in goals]
        seq = [starts, spots, [(r + 1, c) for r, c in spots], spots, [(r - 1, c - 1) for r, c in spots], spots]
        for q in seq:
            for p in q:
                assert free[p], (tag, p)
        gen = og.ObservationGenerator(grid.astype(int).tolist(), ref_params(og))
        P, T = [], []
        for t, q in enumerate(seq):
            pl = [tuple(map(int, x)) for x in q]
            gl = [tuple(map(int, x)) for x in goals]
            if t == 0:
                gen.create_agents(pl, gl)
            gen.update_agents(pl, gl, [0] * len(pl))
            T.append(np.array(gen.generate_observations(), dtype=np.uint8))
            P.append(np.array(q, np.int16))
        P, T = np.array(P), np.array(T)
        G = np.broadcast_to(np.array(goals, np.int16), P.shape).copy()
        # the layout of make_golden.py's trajectory files, so that the generic golden tests replay these as well
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"tok_corner_{tag}.npz"), grid=grid, pos=P, goal=G,
                            actions=np.zeros(P.shape[:2], np.int8), tokens=T, keep=np.arange(P.shape[1]),
                            sha256_all_rows=np.array(hashlib.sha256(T.tobytes()).hexdigest()))
        print(tag, "corner tokens per step:", [[int(T[t][a][120]) for a in range(4)] for t in range(len(seq))])
    # scripted corner visits with non-default grid_step: create inside the window, jump to its (left + 2 step - 5) spot
    for step, (h, w) in {16: (60, 66), 32: (100, 90), 64: (150, 140)}.items():
        rng = np.random.Generator(np.random.PCG64(900 + step))
        g2 = np.ones((h + 10, w + 10), np.uint8)
        g2[5:-5, 5:-5] = rng.random((h, w)) < 0.1
        H2, W2 = g2.shape
        sites = [(l + 2 * step, t + 2 * step) for l in range(0, H2, step) for t in range(0, W2, step)
                 if l + 2 * step <= H2 - 6 and t + 2 * step <= W2 - 6][:12]
        starts, spots = [], []
        for cr, cc in sites:
            st = (cr - 2 * step + min(10, step - 2) + 5, cc - 2 * step + min(10, step - 2) + 5)
            sp = (cr - 5, cc - 5)
            g2[cr - 7: cr + 2, cc - 7: cc + 2] = 0
            g2[cr - 1, cc] = len(spots) % 3 == 1                # some sites: upper neighbour of the corner blocked
            g2[st], g2[sp] = 0, 0
            starts.append(st); spots.append(sp)
        comp = maps.largest_component(g2 == 0)
        keep = [i for i in range(len(sites)) if comp[starts[i]] and comp[spots[i]]]
        starts, spots = [starts[i] for i in keep], [spots[i] for i in keep]
        free = np.argwhere(comp)
        far = free[np.argsort(-(free.sum(1)))]
        goals = [tuple(far[i]) if i % 2 == 0 else tuple(free[rng.integers(0, len(free))]) for i in range(len(starts))]
        seq = [starts, spots, spots, [(r, c + 1) if comp[r, c + 1] else (r, c) for r, c in spots], spots]
        gen = og.ObservationGenerator(g2.astype(int).tolist(), ref_params(og, step))
        P, T = [], []
        for t, q in enumerate(seq):
            pl, gl = [tuple(map(int, x)) for x in q], [tuple(map(int, x)) for x in goals]
            if t == 0:
                gen.create_agents(pl, gl)
            gen.update_agents(pl, gl, [0] * len(pl))
            T.append(np.array(gen.generate_observations(), dtype=np.uint8))
            P.append(np.array(q, np.int16))
        P, T = np.array(P), np.array(T)
        G = np.broadcast_to(np.array(goals, np.int16), P.shape).copy()
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"tok_cornerstep{step}.npz"), grid=g2, pos=P, goal=G,
                            actions=np.zeros(P.shape[:2], np.int8), tokens=T, keep=np.arange(P.shape[1]),
                            sha256_all_rows=np.array(hashlib.sha256(T.tobytes()).hexdigest()), grid_step=np.array(step))
        print(f"cornerstep{step}: {len(starts)} agents, grid {g2.shape}")
    # non-default InputParameters.grid_step (h:38): 33 x 33 / 65 x 65 partial windows, plain random walks with a goal change
    for step, (h, w, dens, n, seed) in {16: (52, 60, 0.12, 36, 31), 32: (90, 84, 0.1, 30, 32)}.items():
        g2 = maps.pad(maps.random_map(h, w, dens, seed))
        pos, goal = maps.place_agents(g2, n, seed=seed)
        tokenizer_case(og, f"step{step}", g2, pos, goal, 40, seed=300 + step, goal_change_at=20,
                       goal_ok=maps.largest_component(g2 == 0), grid_step=step)


if __name__ == "__main__":
    main()
