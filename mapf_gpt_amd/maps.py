"""Map data: named evaluation maps, synthetic generators, padding and agent placement.

Map strings follow the reference's data format (eval_configs/*/maps.yaml; parsed by the reference at
dataset/tokenizer/generate_observations.py:94-111): '#' = obstacle, '.' = free; the warehouse map
also uses '!' (free), '@' (free, allowed start) and '$' (free, allowed goal).  The env pads every map
with `obs_radius` (=5) obstacle cells per side; all coordinates handed to the tokenizer are
(row, col) in that padded frame (observation_generator.cpp:494 indexes +-5 unchecked).

Placement uses our own seeded numpy PCG64 stream: seed-for-seed equality with POGEMA's placement is
not attainable (POGEMA is absent from the reference tree -- SURVEY.md section 0, finding 1).
"""
import json
import os

import numpy as np

OBS_RADIUS = 5
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "named_maps.json")
_named = None


def named_maps():
    global _named
    if _named is None:
        with open(_DATA) as f:
            _named = json.load(f)
    return _named


def parse_map(rows):
    """rows: list[str] or a newline-joined str -> (obst u8[h,w], start_ok bool[h,w], goal_ok bool[h,w])."""
    if isinstance(rows, str):
        rows = [r for r in rows.split("\n") if r != ""]
    h, w = len(rows), len(rows[0])
    a = np.array([list(r.ljust(w, "#")) for r in rows])
    obst = (a == "#").astype(np.uint8)
    free = obst == 0
    start_ok = (a == "@") if (a == "@").any() else free
    goal_ok = (a == "$") if (a == "$").any() else free
    return obst, start_ok & free, goal_ok & free


def pad(arr, r=OBS_RADIUS, value=1):
    return np.pad(arr, r, mode="constant", constant_values=value)


def load_named(name, r=OBS_RADIUS):
    """-> padded (obst u8[H,W], start_ok, goal_ok)."""
    obst, s, g = parse_map(named_maps()[name])
    return pad(obst, r, 1), pad(s, r, False), pad(g, r, False)


def random_map(h, w, density, seed):
    """Bernoulli-obstacle map in the style of the reference's 01-random set (17-21 cells/side, ~0.15)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((h, w)) < density).astype(np.uint8)


def maze_map(h, w, seed, extra_open=0.08):
    """Maze in the style of the 02-mazes set: randomized-DFS corridors on the odd lattice plus a few
    knocked-out walls so that the wall density lands near 0.3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = np.ones((h, w), dtype=np.uint8)
    ch, cw = (h + 1) // 2, (w + 1) // 2
    seen = np.zeros((ch, cw), dtype=bool)
    stack = [(int(rng.integers(ch)), int(rng.integers(cw)))]
    seen[stack[0]] = True
    g[2 * stack[0][0], 2 * stack[0][1]] = 0
    while stack:
        r, c = stack[-1]
        nb = [(r + dr, c + dc) for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1))
              if 0 <= r + dr < ch and 0 <= c + dc < cw and not seen[r + dr, c + dc]]
        if not nb:
            stack.pop()
            continue
        nr, nc = nb[int(rng.integers(len(nb)))]
        seen[nr, nc] = True
        g[2 * nr, 2 * nc] = 0
        g[r + nr, c + nc] = 0
        stack.append((nr, nc))
    g[(rng.random((h, w)) < extra_open)] = 0
    return g


def largest_component(free):
    """bool[H,W] -> bool[H,W] mask of the largest 4-connected free component."""
    H, W = free.shape
    lab = np.zeros((H, W), dtype=np.int32)
    best, best_n, cur = 0, 0, 0
    for r0 in range(H):
        for c0 in range(W):
            if not free[r0, c0] or lab[r0, c0]:
                continue
            cur += 1
            lab[r0, c0] = cur
            stack, n = [(r0, c0)], 0
            while stack:
                r, c = stack.pop()
                n += 1
                for dr, dc in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < H and 0 <= cc < W and free[rr, cc] and not lab[rr, cc]:
                        lab[rr, cc] = cur
                        stack.append((rr, cc))
            if n > best_n:
                best, best_n = cur, n
    return lab == best if best else np.zeros_like(free)


def place_agents(obst, n_agents, seed, start_ok=None, goal_ok=None, component=None):
    """Distinct starts and distinct goals inside the largest free component (padded coords).
    -> (pos int16[n,2], goal int16[n,2])."""
    free = obst == 0
    comp = largest_component(free) if component is None else component
    s_ok = comp if start_ok is None else (comp & start_ok)
    g_ok = comp if goal_ok is None else (comp & goal_ok)
    s_cells, g_cells = np.argwhere(s_ok), np.argwhere(g_ok)
    if len(s_cells) < n_agents or len(g_cells) < n_agents:
        raise ValueError(f"map has {len(s_cells)} start / {len(g_cells)} goal cells, need {n_agents}")
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = s_cells[rng.permutation(len(s_cells))[:n_agents]]
    goal = g_cells[rng.permutation(len(g_cells))[:n_agents]]
    return pos.astype(np.int16), goal.astype(np.int16)
