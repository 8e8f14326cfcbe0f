#!/usr/bin/env python3
"""Adapter I/O fixture CAPTURED from the reference's own `mapf_gpt/inference.py` (build container only; needs
/root/reference and oracle/_ref).  SURVEY 8c item 4 / VERDICT r1 "missing" 7.

The reference adapter is imported as it is and driven through a scripted multi-environment episode; what is recorded per
`act_batch` call is everything that crosses its boundary: the observation dicts going in, the token rows it hands to the
policy (`net.act(tensor_obs, generator=...)`, inference.py:87-101: one entry per batch_size chunk) and the action lists it
returns.  The policy is a deterministic stand-in injected through the constructor's own `net=` parameter (inference.py:48,
79-80): action = (sum of the row's tokens + 3 * position of the row in its chunk) mod 5 -- chunking and row order become
visible in the outputs.  Positions evolve by this repo's env step applied to the returned actions.
What has to be faked to import the file at all, since the packages are absent from the image: `pogema_toolbox`'s `AlgoBase`
(a pydantic model holding `name`, `num_process`, `device`, `parallel_backend`) and `ToolboxRegistry` logging calls, `cppimport`
(the pybind module is the one oracle/_ref builds from the reference's own source), `loguru`.  None of it is on the recorded path.
Run:  python tests/golden/make_golden_adapter.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "adapter_io.npz")

from mapf_gpt_amd import maps, weights  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class FakeNet:
    """Deterministic policy stand-in; records the rows of every chunk it is given."""

    def __init__(self):
        self.chunks = []

    def act(self, idx, do_sample=True, generator=None):
        import torch
        rows = idx.detach().cpu().to(torch.int64)
        self.chunks.append(rows.numpy().astype(np.uint8))
        a = (rows.sum(1) + 3 * torch.arange(rows.shape[0])) % 5
        return a.reshape(-1, 1).to(idx.device)


def import_reference_adapter():
    import pydantic
    import torch
    orc.build_ref()
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    sys.path.insert(0, ref_dir)
    import observation_generator as og                       # the reference's pybind module, built from its own source
    lg = types.ModuleType("loguru")
    lg.logger = type("L", (), {"__getattr__": lambda s, k: (lambda *a, **kw: None)})()
    sys.modules["loguru"] = lg
    pt = types.ModuleType("pogema_toolbox")
    ac = types.ModuleType("pogema_toolbox.algorithm_config")

    class AlgoBase(pydantic.BaseModel):
        name: str = None
        num_process: int = 3
        device: str = "cuda"
        parallel_backend: str = "multiprocessing"
    ac.AlgoBase = AlgoBase
    rg = types.ModuleType("pogema_toolbox.registry")
    rg.ToolboxRegistry = type("R", (), {k: staticmethod(lambda *a, **kw: None) for k in ("info", "debug", "warning", "success", "error")})
    sys.modules.update({"pogema_toolbox": pt, "pogema_toolbox.algorithm_config": ac, "pogema_toolbox.registry": rg})
    ci = types.ModuleType("cppimport")
    sys.modules["cppimport"] = ci
    sys.modules["cppimport.import_hook"] = types.ModuleType("cppimport.import_hook")
    sys.path.insert(0, "/root/reference")
    import mapf_gpt
    sys.modules["mapf_gpt.observation_generator"] = og
    mapf_gpt.observation_generator = og
    from mapf_gpt.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    return MAPFGPTInference, MAPFGPTInferenceConfig, torch


def observations(grid, pos, goal):
    return [{"global_xy": (int(p[0]), int(p[1])), "global_target_xy": (int(g[0]), int(g[1])), "global_obstacles": grid.astype(np.int64)}
            for p, g in zip(pos, goal)]


def main():
    Inference, Config, torch = import_reference_adapter()
    ck = "/tmp/mgpt_adapter_ckpt.pt"
    args = weights.model_args("tiny")
    torch.save({"model": {"_orig_mod." + k: torch.from_numpy(v) for k, v in weights.synthetic_state_dict("tiny", seed=0).items()},
                "model_args": args}, ck)
    net = FakeNet()
    algo = Inference(Config(path_to_weights=ck, device="cpu", batch_size=11), net=net)
    envs = [("validation-random-seed-000", 9, 1), ("validation-mazes-seed-000", 16, 2), ("puzzle-00", 3, 3)]
    slots = [7, 2, 40]                                        # act_batch position keys (inference.py:151-157)
    state = []
    for name, n, seed in envs:
        grid, s_ok, g_ok = maps.load_named(name)
        pos, goal = maps.place_agents(grid, n, seed, s_ok, g_ok)
        state.append([grid, pos.astype(np.int32), goal.astype(np.int32)])
    rec = {"n_env": np.array(len(envs)), "slots": np.array(slots), "batch_size": np.array(11)}
    for e, (grid, pos, goal) in enumerate(state):
        rec[f"grid{e}"] = grid.astype(np.uint8)
    calls = []
    for t in range(14):
        if t == 9:
            algo.reset_states()                               # inference.py:174-177: generators and action memory dropped
            calls.append(("reset",))
        active = [0, 1, 2] if t % 4 != 3 else [1, 0]          # a call with a subset, in another order
        net.chunks = []
        obs = [observations(state[e][0], state[e][1], state[e][2]) for e in active]
        if t == 5:                                            # one environment alone through act() (inference.py:148-149)
            out = [algo.act(obs[0])]
            active, obs = active[:1], obs[:1]
        else:
            out = algo.act_batch(obs, positions=[slots[e] for e in active])
        calls.append(("act", list(active), [state[e][1].copy() for e in active], [state[e][2].copy() for e in active],
                      [c.copy() for c in net.chunks], [list(map(int, o)) for o in out]))
        for e, acts in zip(active, out):
            state[e][1], _ = orc.env_step(state[e][0], state[e][1], state[e][2], np.asarray(acts, np.int32))
    k = 0
    kinds = []
    for c in calls:
        if c[0] == "reset":
            kinds.append(-1)
            continue
        kinds.append(k)
        _, active, P, G, chunks, out = c
        rec[f"c{k}_active"] = np.array(active)
        for j, e in enumerate(active):
            rec[f"c{k}_pos{j}"] = P[j].astype(np.int16)
            rec[f"c{k}_goal{j}"] = G[j].astype(np.int16)
            rec[f"c{k}_out{j}"] = np.array(out[j], np.int8)
        rec[f"c{k}_nchunks"] = np.array(len(chunks))
        for j, ch in enumerate(chunks):
            rec[f"c{k}_chunk{j}"] = ch
        k += 1
    rec["sequence"] = np.array(kinds)
    np.savez_compressed(OUT, **rec)
    print("calls", k, "->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
