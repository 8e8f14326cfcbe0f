"""The whole step (env -> tokenizer -> forward -> sample -> env) at BASELINE.json's FULL per-GPU sizes, through BatchedRunner
(one mgpt_step_run per step, graph replay from the second step on).  At these sizes the oracle cannot follow every row, so
the checks are the size-independent ones: sampled instances teacher-forced against the oracle (tokens and env step bit for
bit), logits of sampled rows against the fp32 torch port, row invariants over ALL rows, env invariants over ALL instances,
and replicated instances (same map, starts, goals; arg-max actions) staying identical through the run."""
import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps, weights
from oracle import gpt_oracle
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _workload(name):
    import bench
    from mapf_gpt_amd.runner import make_instances
    map_name, n_agents, n_inst, model, _ = bench.WORKLOADS[name]
    if name == "cfg4":
        grids, pos, goal = bench.cfg4_instances(0, n_inst, n_agents)
        return grids, pos, goal, n_inst, n_agents, model
    grid, s_ok, g_ok = maps.load_named(map_name)
    pos, goal = make_instances(grid, n_inst, n_agents, first_seed=0, start_ok=s_ok, goal_ok=g_ok)
    return grid, pos, goal, n_inst, n_agents, model


@pytest.mark.parametrize("name,precision,tol", [("cfg2", "f16x3", 1e-5), ("cfg3", "f16x3", 1e-5), ("cfg4", "f16x3", 1e-5), ("cfg5", "bf16", 1e-1)])
def test_whole_step_at_full_size(name, precision, tol):
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner
    grids, pos, goal, n_inst, n, model = _workload(name)
    per_inst = grids.ndim == 3
    half = n_inst // 2
    if not per_inst:                                   # shared map: the second half replicates the first (arg-max policy below)
        pos[half:] = pos[:half]
        goal[half:] = goal[:half]
    # cfg2: the whole 16 384-row step is ONE forward launch, as in bench.py (that launch was never compared with anything)
    net = build_model(model, seed=0, max_rows=min(n_inst * n, 16384 if name == "cfg2" else (4096 if model != "85M" else 1024)), precision=precision)
    run = BatchedRunner(grids, n_inst, n, net, max_episode_steps=64, seed=0, do_sample=False, precision=precision)
    run.reset(pos, goal)
    sample = [0, n_inst // 3, n_inst - 1] if per_inst else [0, half // 2, half - 1]
    grid_of = (lambda i: grids[i]) if per_inst else (lambda i: grids)
    gens = {i: orc.OracleGenerator(grid_of(i)) for i in sample}
    p = {i: pos[i].numpy().astype(np.int32).copy() for i in sample}
    g = {i: goal[i].numpy().astype(np.int32) for i in sample}
    last = {i: np.full(n, -1, np.int32) for i in sample}
    for i in sample:
        gens[i].create_agents(p[i], g[i])
    prev = pos.numpy().astype(np.int32)
    free = (grids == 0)
    for t in range(3):
        run.step()
        tokens = run.tokens.cpu().numpy().reshape(n_inst, n, 256)
        actions = run.actions.cpu().numpy()
        cur = run.env.sync_state()[0].cpu().numpy().astype(np.int32)
        # ---- every row ----
        assert tokens.max() <= 66
        assert (tokens[:, :, 60] == 20).all(), "window centre = distance to self"
        assert (tokens[:, :, 121] == 20).all() and (tokens[:, :, 122] == 20).all(), "slot 0 is the agent itself"
        assert (tokens[:, :, 251:] == 66).all()
        assert actions.min() >= 0 and actions.max() <= 4
        # ---- every instance: env invariants ----
        assert (np.abs(cur - prev).sum(2) <= 1).all(), "at most one cell per step"
        cells = cur[:, :, 0].astype(np.int64) * 100000 + cur[:, :, 1]
        assert (np.sort(cells, axis=1)[:, 1:] != np.sort(cells, axis=1)[:, :-1]).all(), "vertex conflict"
        on_free = free[np.arange(n_inst)[:, None], cur[:, :, 0], cur[:, :, 1]] if per_inst else free[cur[:, :, 0], cur[:, :, 1]]
        assert on_free.all(), "agent on an obstacle"
        if not per_inst:
            assert np.array_equal(tokens[half:2 * half], tokens[:half]) and np.array_equal(cur[half:2 * half], cur[:half]), "replicas diverged"
        # ---- sampled instances against the oracle ----
        for i in sample:
            gens[i].update_agents(p[i], g[i], last[i])
            assert np.array_equal(tokens[i], gens[i].generate_observations()), f"{name}: tokens of instance {i}, step {t}"
            exp, _ = orc.env_step(grid_of(i), p[i], g[i], actions[i])
            assert np.array_equal(cur[i], exp), f"{name}: env step of instance {i}, step {t}"
            p[i], last[i] = exp, actions[i].copy()
        prev = cur
    # ---- logits of the last step's rows: the forward of ALL rows (the launches the runner itself issues), sampled rows of it
    #      against the fp32 port, the replicated half against the first half, and a small separate launch against the big one ----
    all_logits = net.logits_tokens(run.tokens).cpu().numpy().reshape(n_inst, n, 67)
    assert np.isfinite(all_logits).all()
    if not per_inst:
        assert np.array_equal(all_logits[half:2 * half], all_logits[:half]), "replicated instances: logits differ inside one forward"
    pick = [(i, a) for i in sample for a in list(range(0, n, max(1, n // 4)))[:4]]
    rows = np.stack([tokens[i][a] for i, a in pick])
    sd, args = weights.synthetic_state_dict(model, seed=0), weights.model_args(model)
    ref = gpt_oracle.forward_logits(sd, args, rows).numpy()
    got = np.stack([all_logits[i][a] for i, a in pick])
    err = float(np.abs(got - ref).max())
    assert err <= tol, f"{name} {precision}: max |dlogit| = {err:.3e}"
    small = net.logits_tokens(torch.from_numpy(np.ascontiguousarray(rows)).cuda()).cpu().numpy()
    # (the kernels are chosen per CALL: a small call keeps the 32 x 32 x 16 MFMA kernels, a large one runs on 16 x 16 x 32 -- other summation orders,
    #  the same envelope: both are within tol of the fp32 port)
    dsl = float(np.abs(small - got).max())
    assert dsl <= tol, f"a row's logits depend on the launch it rides in beyond the precision envelope: {dsl:.3e}"
    if precision != "bf16":                              # arg-max actions of those rows are the port's
        assert np.array_equal(got[:, :5].argmax(1), ref[:, :5].argmax(1))
