#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference (build container only).

Needs /root/reference (absent on the GPU box -- nothing at test time reads it; tests only read the
.npz files this script wrote).  Inputs are produced by repo-owned seeded generators; expected
outputs come from
  * the reference tokenizer mapf_gpt/observation_generator.cpp compiled by `make -C oracle ref`
    (imported from oracle/_ref), and
  * the reference policy mapf_gpt/model.py imported from /root/reference with a 3-line loguru stub
    (only logger.warning/debug are used: model.py:7,41,217-224).
Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from mapf_gpt_amd import maps, weights  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def import_reference():
    orc.build_ref()
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    sys.path.insert(0, "/root/reference")
    lg = types.ModuleType("loguru")
    lg.logger = type("L", (), {"__getattr__": lambda s, k: (lambda *a, **kw: None)})()
    sys.modules["loguru"] = lg
    import observation_generator as og
    from mapf_gpt.model import GPT, GPTConfig
    return og, GPT, GPTConfig


def ref_params(og, grid_step=64):
    return og.InputParameters(20, 13, 5, 256, 5, 5, grid_step, False)   # inference.py:109-118 defaults


def walk(grid, pos, goal, steps, seed, goal_change_at=None, goal_ok=None):
    """Collision-free seeded walk: random INTENDED actions (many are blocked, so the action history
    differs from the displacement), executed through our env spec.  Yields per-step tokenizer inputs:
    (pos_t, goal_t, actions_fed_t) with actions_fed_0 = -1 (inference.py:140)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(pos)
    pos, goal = pos.astype(np.int32).copy(), goal.astype(np.int32).copy()
    last = np.full(n, -1, dtype=np.int32)
    for t in range(steps):
        if goal_change_at is not None and t == goal_change_at:
            free = np.argwhere((grid == 0) if goal_ok is None else goal_ok)
            who = rng.permutation(n)[: max(1, n // 3)]
            goal[who] = free[rng.permutation(len(free))[: len(who)]]
        yield pos.copy(), goal.copy(), last.copy()
        act = rng.integers(0, 5, size=n).astype(np.int32)
        pos, _ = orc.env_step(grid, pos, goal, act)
        last = act


def tokenizer_case(og, name, grid, pos, goal, steps, seed, goal_change_at=None, goal_ok=None, keep=None, grid_step=64):
    gen = og.ObservationGenerator(grid.astype(int).tolist(), ref_params(og, grid_step))
    P, G, A, T = [], [], [], []
    for t, (p, g, a) in enumerate(walk(grid, pos, goal, steps, seed, goal_change_at, goal_ok)):
        pl, gl = [tuple(map(int, x)) for x in p], [tuple(map(int, x)) for x in g]
        if t == 0:
            gen.create_agents(pl, gl)                                   # inference.py:138
        gen.update_agents(pl, gl, [int(x) for x in a])                  # inference.py:142-144
        tok = np.array(gen.generate_observations(), dtype=np.int64)     # inference.py:145
        assert tok.min() >= 0 and tok.max() <= 66
        P.append(p); G.append(g); A.append(a); T.append(tok.astype(np.uint8))
    P, G, A, T = np.array(P, np.int16), np.array(G, np.int16), np.array(A, np.int8), np.array(T)
    sha = hashlib.sha256(T.tobytes()).hexdigest()
    if keep is not None:      # keep the full inputs (needed to rebuild state) but only a sample of output rows
        T = T[:, keep]
    np.savez_compressed(os.path.join(OUT, f"tok_{name}.npz"), grid=grid.astype(np.uint8), pos=P, goal=G,
                        actions=A, tokens=T, keep=np.arange(P.shape[1]) if keep is None else np.asarray(keep),
                        sha256_all_rows=np.array(sha), **({} if grid_step == 64 else {"grid_step": np.array(grid_step)}))
    print(f"tok_{name}: grid {grid.shape} agents {P.shape[1]} steps {steps} tokens {T.shape} sha {sha[:16]}")
    return T


def main():
    og, GPT, GPTConfig = import_reference()
    import torch

    # 1. the reference's own smoke scenario, observation_generator.cpp:530-544
    gen = og.ObservationGenerator([[0] * 256 for _ in range(256)], ref_params(og))
    gen.create_agents([(120, 120)], [(20, 200)])
    gen.update_agents([(120, 120)], [(20, 200)], [0])
    row = np.array(gen.generate_observations(), dtype=np.uint8)
    sha = hashlib.sha256(row[0].tobytes()).hexdigest()
    assert sha == "896eb85aa89a369759917e5903f237dc28387e6b7d431fbd6703f302a97585e1", sha
    np.savez_compressed(os.path.join(OUT, "tok_known_answer.npz"), tokens=row, sha256=np.array(sha))

    # 2. trajectories on the evaluation maps + synthetic large-map / goal-change cases
    rows_for_model = []
    cases = [("random000", "validation-random-seed-000", 32, 24, None),
             ("mazes000", "validation-mazes-seed-000", 64, 24, None),
             ("warehouse", "wfi_warehouse", 192, 10, None),
             ("berlin", "Berlin_1_256_00", 256, 8, None),
             ("puzzle00", "puzzle-00", 4, 16, 8)]
    for tag, name, n, steps, gc in cases:
        grid, s_ok, g_ok = maps.load_named(name)
        pos, goal = maps.place_agents(grid, n, seed=0, start_ok=s_ok, goal_ok=g_ok)
        keep = None
        if n > 64:
            keep = np.sort(np.random.Generator(np.random.PCG64(1)).permutation(n)[:48])
        T = tokenizer_case(og, tag, grid, pos, goal, steps, seed=100, goal_change_at=gc, keep=keep)
        rows_for_model.append(T[steps // 2, : min(4, T.shape[1])])
    # large rectangular synthetic maps: the reference's tiled cost2go branch (cpp:222-279) runs only when H or W > 64
    for tag, (h, w, dens, n, seed) in {"rect50x160": (50, 160, 0.2, 40, 7), "rect140x70": (140, 70, 0.3, 24, 8)}.items():
        grid = maps.pad(maps.random_map(h, w, dens, seed))
        pos, goal = maps.place_agents(grid, n, seed=seed)
        tokenizer_case(og, tag, grid, pos, goal, 12, seed=200 + seed, goal_change_at=6,
                       goal_ok=maps.largest_component(grid == 0))
    rows = np.concatenate(rows_for_model, axis=0)[:16].astype(np.uint8)

    # 3. policy logits from the real model.py on synthetic weights (released weights need network)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name, scales in {"tiny": (1.0, 4.0), "2M": (1.0, 4.0), "6M": (1.0,), "85M": (1.0,)}.items():
        args = weights.model_args(name)
        for scale in scales:
            sd = weights.synthetic_state_dict(name, seed=0, scale=scale)
            net = GPT(GPTConfig(**args)).eval()
            missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            B = 16 if name != "85M" else 4
            idx = torch.from_numpy(rows[:B].astype(np.int64))
            with torch.no_grad():
                logits, _ = net(idx)                                    # model.py:167-189
                greedy = net.act(idx, do_sample=False)                  # model.py:244-260
            out = dict(tokens=rows[:B], logits=logits[:, 0, :].numpy().astype(np.float32),
                       greedy=greedy.numpy().astype(np.int64), seed=np.array(0), scale=np.array(scale))
            if name == "tiny":   # per-layer residual stream for one row, to localise errors
                hs = []
                with torch.no_grad():
                    x = net.transformer.wte(idx[:1]) + net.transformer.wpe(torch.arange(256))
                    hs.append(x[0].numpy().copy())
                    for blk in net.transformer.h:
                        x = blk(x)
                        hs.append(x[0].numpy().copy())
                out["layers"] = np.stack(hs).astype(np.float32)
            tag = f"gpt_{name}_s{int(scale)}"
            np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
            print(tag, "logits", out["logits"].shape, "max|logit|", float(np.abs(out["logits"]).max()), missing)


if __name__ == "__main__":
    main()
