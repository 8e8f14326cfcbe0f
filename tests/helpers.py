"""Shared test helpers (CPU side)."""
import glob
import hashlib
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tok_cases():
    return sorted(os.path.basename(p)[4:-4] for p in glob.glob(os.path.join(GOLDEN, "tok_*.npz"))
                  if "known_answer" not in p)


def load_tok(name):
    return np.load(os.path.join(GOLDEN, f"tok_{name}.npz"))


def tokp_cases():
    """goldens of NON-DEFAULT InputParameters (tests/golden/make_golden_params.py, from the compiled reference)"""
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "tokp_*.npz")))


def load_tokp(name):
    return np.load(os.path.join(GOLDEN, f"tokp_{name}.npz"))


def replay_oracle(case):
    """Run the C oracle over a golden trajectory -> uint8 [S, n, 256] (all agents)."""
    grid, P, G, A = case["grid"], case["pos"], case["goal"], case["actions"]
    params = None
    if "params" in case:                   # (limit, num_agents, previous actions, context, obs radius, agents radius)
        L, S, Hn, _, R, Ar = [int(v) for v in case["params"]]
        params = (L, S, Hn, R, Ar)
    gen = orc.OracleGenerator(grid, grid_step=int(case["grid_step"]) if "grid_step" in case else 64, params=params)
    out = []
    for t in range(P.shape[0]):
        if t == 0:
            gen.create_agents(P[0], G[0])
        gen.update_agents(P[t], G[t], A[t].astype(np.int32))
        out.append(gen.generate_observations())
    return np.array(out)


def sha_rows(tokens):
    return hashlib.sha256(np.ascontiguousarray(tokens, dtype=np.uint8).tobytes()).hexdigest()


def record_parity(**entry):
    """Append one measured parity record (shape, scale, precision, errors, ...) to gpurun_out/parity_records.jsonl: the GPU run
    leaves the numbers behind instead of printing them into a -q run; the round's copy is committed
    under profiles/."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_records.jsonl"), "a") as f:
        f.write(json.dumps({k: (float(v) if hasattr(v, "__float__") and not isinstance(v, (int, str, bool)) else v) for k, v in entry.items()}) + "\n")
