"""CPU restatement of the reference's DATASET-side tokenizer (dataset/tokenizer/generate_observations.py +
cost2go.cpp + encoder.cpp) -- test infrastructure only, like everything under oracle/.

It differs from the inference tokenizer (observation_generator.cpp) in three ways (generate_observations.py):
  * neighbours inside the 11x11 window are ordered by the BFS distance from the OBSERVER'S CELL to theirs, then id, and
    agents in a different component are dropped (:121-141); inference orders by Manhattan distance;
  * action history comes from the logged path (executed moves), padded with "n" at the start of an episode and one "w"
    at its last step (:206-229); the goal is the path's last cell (:199);
  * all timesteps of a logged episode are emitted, with the ground-truth action next to every row (:73-90).
Lifelong logs ("global_lifelong_targets_xy", :55-60, 143-153): the goal of an agent at a timestep is the first target of its
list it has not stood on yet; the records use it (relative goal, greedy bits) while the cost-to-go window keeps the field
of the path's LAST cell (:75-78).  mask_cost2go (:253-262, cost2go.cpp:52-62): window cells become 0 / 1 (blocked).
Token ids and the row layout are the inference tokenizer's (encoder.cpp:52-127).
"""
import numpy as np

MOVES = [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)]                    # generate_observations.py:10
LETTER = {(0, 0): 45, (-1, 0): 46, (1, 0): 47, (0, -1): 48, (0, 1): 49}   # w u d l r (:11-17); 'n' = 44
TOK_N, TOK_W, TOK_BITS0, TOK_PAD = 44, 45, 50, 66


def bfs_from(grid, src):
    """= cost2go.cpp:8-31: 4-connected BFS distances from `src` over cells with grid == 0, -1 elsewhere."""
    H, W = grid.shape
    d = np.full((H, W), -1, np.int32)
    d[src] = 0
    frontier = [src]
    while frontier:
        nxt = []
        for (r, c) in frontier:
            for dr, dc in MOVES[1:]:
                nr, nc = r + dr, c + dc
                if 0 <= nr < H and 0 <= nc < W and grid[nr, nc] == 0 and d[nr, nc] < 0:
                    d[nr, nc] = d[r, c] + 1
                    nxt.append((nr, nc))
        frontier = nxt
    return d


def int_token(v, limit=20):
    """encoder.cpp:52-73: -20..20 -> 0..40, -80 -> 41, -40 -> 42, 40 -> 43 (anything else raises there)."""
    if -limit <= v <= limit:
        return v + limit
    return {-4 * limit: 2 * limit + 1, -2 * limit: 2 * limit + 2, 2 * limit: 2 * limit + 3}[v]


def agent_paths(init_positions, made_actions):
    """= get_agent_paths (:159-177) for cost2go_radius == 5: positions after every logged action."""
    paths = []
    for p0, acts in zip(init_positions, made_actions):
        cur = [int(p0[0]), int(p0[1])]
        path = [tuple(cur)]
        for a in acts:
            cur[0] += MOVES[a][0]; cur[1] += MOVES[a][1]
            path.append(tuple(cur))
        paths.append(path)
    return paths


def goal_positions(paths, targets):
    """= get_goal_positions (:143-153); running past the end of a target list raises IndexError there as well."""
    out = []
    for path, tg in zip(paths, targets):
        cur, gp = 0, []
        for pos in path:
            if tuple(pos) == (int(tg[cur][0]), int(tg[cur][1])):
                cur += 1
            gp.append((int(tg[cur][0]), int(tg[cur][1])))
        out.append(gp)
    return out


def generate_observations(grid, init_positions, made_actions, num_agents=13, npa=5, agents_radius=5, radius=5, limit=20,
                          lifelong_targets=None, mask_cost2go=False):
    """One logged instance -> (inputs int8 [n * (T+1), 256], gt_actions int64 [n * (T+1)]), rows agent-major then time,
    exactly the append order of generate_observations (:73-90)."""
    grid = np.asarray(grid)
    paths = agent_paths(init_positions, made_actions)
    n, L = len(paths), len(paths[0])
    goals = goal_positions(paths, lifelong_targets) if lifelong_targets is not None else None      # :55-60
    cache = {}

    def field(src):
        if src not in cache:
            cache[src] = bfs_from(grid, src)
        return cache[src]

    actions = [list(a) + [0] for a in made_actions]                     # :67-68
    rows, gts = [], []
    for a in range(n):
        acts = actions[a]
        goal_t = len(acts)                                              # find_last_non_zero_index (:33-37)
        for i in reversed(range(len(acts))):
            if acts[i] != 0:
                goal_t = i
                break
        dg = field(paths[a][-1])
        for t in range(L):
            me = paths[a][t]
            # --- neighbours (generate_agent_proximity, :121-141) ---
            dm = field(me)
            cand = []
            for j in range(n):
                pj = paths[j][t]
                if abs(me[0] - pj[0]) <= agents_radius and abs(me[1] - pj[1]) <= agents_radius and dm[pj] >= 0:
                    cand.append((int(dm[pj]), j))
            cand.sort()
            toks = []
            # --- window (cost2go.cpp:44-88) ---
            mid = int(dg[me])
            for i in range(2 * radius + 1):
                for jj in range(2 * radius + 1):
                    v = int(dg[me[0] - radius + i, me[1] - radius + jj])
                    if mask_cost2go:                                    # cost2go.cpp:52-62
                        v = 1 if v < 0 else 0
                    elif v >= 0:
                        v -= mid
                        v = 2 * limit if v > limit else (-2 * limit if v < -limit else v)
                    else:
                        v = -4 * limit
                    toks.append(int_token(v, limit))
            # --- agent records (get_agent_info :179-245, encoder.cpp:88-108) ---
            for _, j in cand[:num_agents]:
                pj = paths[j][min(t, L - 1)]
                gj = goals[j][min(t, L - 1)] if goals is not None else paths[j][-1]     # :194-199
                toks += [int_token(pj[0] - me[0], limit), int_token(pj[1] - me[1], limit),
                         int_token(gj[0] - me[0], limit), int_token(gj[1] - me[1], limit)]
                if t < npa:                                             # :206-214
                    hist = [TOK_N] * (npa - t) + [LETTER[(paths[j][i][0] - paths[j][i - 1][0], paths[j][i][1] - paths[j][i - 1][1])]
                                                   for i in range(1, min(t + 1, L - 1))]
                else:                                                   # :215-224
                    hist = [LETTER[(paths[j][i][0] - paths[j][i - 1][0], paths[j][i][1] - paths[j][i - 1][1])]
                            for i in range(t - npa + 1, min(t + 1, L - 1))]
                hist += [TOK_W] * (npa - len(hist))                     # :225-228
                toks += hist
                dgj = field(gj)
                bits = 0
                for m in MOVES[1:]:                                     # :230-243, order u d l r
                    q = (pj[0] + m[0], pj[1] + m[1])
                    bits = bits * 2 + (1 if (dgj[q] >= 0 and dgj[pj] > dgj[q]) else 0)
                toks.append(TOK_BITS0 + bits)
            toks += [TOK_PAD] * (256 - len(toks))                       # encoder.cpp:110-125
            rows.append(np.array(toks, dtype=np.int8))
            g = acts[t]
            if t > goal_t:
                g = 5                                                   # :88-89 "wait in goal"
            gts.append(g)
    return np.stack(rows), np.array(gts, dtype=np.int64)
