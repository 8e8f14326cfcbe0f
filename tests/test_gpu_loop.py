"""End-to-end hot path on the device: adapter bookkeeping (inference.py:127-172) and the batched runner,
teacher-forced against the oracle step by step."""
import os

import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps, weights
from oracle import gpt_oracle
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


class _GreedyNet:
    """inject-able `net` (inference.py:79-80) whose act() is deterministic (do_sample=False)."""

    def __init__(self, net):
        self.net = net

    def act(self, idx, generator=None):
        return self.net.act(idx, do_sample=False)


def test_adapter_act_batch_bookkeeping():
    from mapf_gpt_amd.env import GridEnv
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    from mapf_gpt_amd.model import build_model
    net = build_model("tiny", seed=0, max_rows=32)
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", batch_size=16), net=_GreedyNet(net))
    envs = [GridEnv(map_name="validation-random-seed-000", num_agents=12, seed=s, max_episode_steps=8) for s in (0, 1)]
    sd, args = weights.synthetic_state_dict("tiny", seed=0), weights.model_args("tiny")
    algo.reset_states()
    obs = [e.reset()[0] for e in envs]
    gens = [orc.OracleGenerator(e.grid) for e in envs]
    last = [np.full(12, -1, np.int32) for _ in envs]
    for t in range(6):
        acts = algo.act_batch(obs, positions=[10, 20])            # slot keys, inference.py:151-157
        assert [len(a) for a in acts] == [12, 12] and all(isinstance(x, int) for x in acts[0])
        for i, e in enumerate(envs):
            p = np.array([o["global_xy"] for o in obs[i]], np.int32)
            g = np.array([o["global_target_xy"] for o in obs[i]], np.int32)
            if t == 0:
                gens[i].create_agents(p, g)
            gens[i].update_agents(p, g, last[i])                  # the adapter must feed ITS OWN previous actions back
            rows = gens[i].generate_observations()
            logits = gpt_oracle.forward_logits(sd, args, rows).numpy()
            top2 = np.sort(logits[:, :5], axis=1)[:, -2:]
            safe = (top2[:, 1] - top2[:, 0]) > 1e-4
            exp = logits[:, :5].argmax(1)
            assert np.array_equal(np.array(acts[i])[safe], exp[safe]), f"step {t} env {i}"
            last[i] = np.array(acts[i], np.int32)
            assert algo._last_actions[[10, 20][i]] == acts[i]     # inference.py:168
        obs = [e.step(a)[0] for e, a in zip(envs, acts)]
    single = algo.act(obs[0])                                     # inference.py:148-149 -> slot 0, fresh generator
    assert len(single) == 12 and 0 in algo._obs_generators
    with pytest.raises(ValueError):                               # a slot twice in one call would lose the first entry
        algo.act_batch([obs[0], obs[1]], positions=[7, 7])
    algo.reset_states()
    assert algo._obs_generators == {} and algo._last_actions == {}


def test_adapter_truncates_fractional_obstacles_like_the_reference():
    """inference.py:135 builds the grid with .astype(int), so 0.5 is a FREE cell and 1.7 an obstacle; the rows of an
    environment whose global_obstacles carries such values must equal the rows of the truncated integer map."""
    from mapf_gpt_amd.env import GridEnv
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    from mapf_gpt_amd.model import build_model
    net = build_model("tiny", seed=0, max_rows=32)
    env = GridEnv(map_name="validation-random-seed-000", num_agents=8, seed=2, max_episode_steps=8)
    obs = env.reset()[0]
    g = np.asarray(obs[0]["global_obstacles"]).astype(np.float64)
    frac = np.where(g != 0, 1.7, 0.5)                             # same truncated map, fractional entries
    obs_frac = [dict(o, global_obstacles=frac) for o in obs]
    rows = []
    for o in (obs, obs_frac):
        algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", batch_size=16), net=_GreedyNet(net))
        rows.append(algo._prepare_inputs(0, o).cpu().numpy())
    assert np.array_equal(rows[0], rows[1])


def test_adapter_shared_context_subset_calls_vs_oracle():
    """Slots of equal shape share one tokenizer context (one instance per slot, own map each); a call may present any subset
    of them and in any order: absent slots keep their state (action history, goal field, window origin).  Five environments
    (three on one map shape with 10 agents, one on the same shape with another map, one with 7 agents) driven through
    _tokenize with changing subsets; every row is compared with that environment's own oracle generator (the reference keeps one
    ObservationGenerator per slot, inference.py:133-145), fed the same positions and the same previous actions."""
    from mapf_gpt_amd.env import GridEnv
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", batch_size=64))
    specs = [("validation-random-seed-000", 10, 0), ("validation-random-seed-000", 10, 1), ("validation-random-seed-001", 10, 2),
             ("validation-random-seed-000", 7, 3), ("validation-random-seed-000", 10, 4)]
    envs = [GridEnv(map_name=m, num_agents=n, seed=sd, max_episode_steps=64) for m, n, sd in specs]
    slots = [100, 7, 42, 5, 3]
    obs = [e.reset()[0] for e in envs]
    gens = [orc.OracleGenerator(e.grid) for e in envs]
    last = [np.full(n, -1, np.int32) for _, n, _ in specs]
    started = [False] * 5
    rng = np.random.Generator(np.random.PCG64(9))
    algo.reset_states()
    subsets = [[0, 1, 2, 3], [1], [3, 0], [0, 1, 2, 3, 4], [4, 2], [2, 1, 0], [0, 1, 2, 4, 3], [3]]
    for t, sub in enumerate(subsets):
        rows = algo._tokenize([slots[e] for e in sub], [obs[e] for e in sub]).cpu().numpy()
        off = 0
        for e in sub:
            n = specs[e][1]
            p = np.array([o["global_xy"] for o in obs[e]], np.int32)
            g = np.array([o["global_target_xy"] for o in obs[e]], np.int32)
            if not started[e]:
                gens[e].create_agents(p, g)
                started[e] = True
            gens[e].update_agents(p, g, last[e])
            assert np.array_equal(rows[off:off + n], gens[e].generate_observations()), f"call {t} env {e}"
            off += n
            act = rng.integers(0, 5, n).astype(np.int32)         # what act_batch would have stored (inference.py:168)
            algo._last_actions[slots[e]] = act.tolist()
            last[e] = act
            obs[e] = envs[e].step(act.tolist())[0]
    # contexts: slots 0 and 1 (same frame, 10 agents, first call) share one; slot 2 (another map frame), slot 3 (7 agents) and
    # slot 4 (first seen in a later call) have their own
    g = [algo._obs_generators[slots[e]] for e in range(5)]
    assert g[0] is g[1] and g[0].k == 2 and (envs[2].grid.shape != envs[0].grid.shape) == (g[2] is not g[0])
    assert g[3].n == 7 and g[3] is not g[0] and g[4].k == 1 and g[4] is not g[0]


def test_adapter_accepts_pretokenised_rows_and_chunks():
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    from tests.helpers import load_tok
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", batch_size=5))
    rows = load_tok("random000")["tokens"][2, :12].astype(int).tolist()       # inference.py:128,146 pass-through
    out = algo.act_batch([rows[:7], rows[7:]])
    assert [len(o) for o in out] == [7, 5] and all(0 <= a <= 4 for o in out for a in o)


def test_batched_runner_teacher_forced_vs_oracle():
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner, make_instances
    grid, s_ok, g_ok = maps.load_named("validation-mazes-seed-000")
    n_inst, n = 3, 20
    net = build_model("tiny", seed=0, max_rows=64)
    pos, goal = make_instances(grid, n_inst, n, 0, s_ok, g_ok)
    run = BatchedRunner(grid, n_inst, n, net, max_episode_steps=32, seed=5, do_sample=True)
    run.reset(pos, goal)
    gens = [orc.OracleGenerator(grid) for _ in range(n_inst)]
    p, g = pos.numpy().astype(np.int32).copy(), goal.numpy().astype(np.int32)
    last = np.full((n_inst, n), -1, np.int32)
    for i in range(n_inst):
        gens[i].create_agents(p[i], g[i])
    for t in range(10):
        run.step()
        tokens = run.tokens.cpu().numpy().reshape(n_inst, n, 256)
        actions = run.actions.cpu().numpy()
        newpos = run.env.sync_state()[0].cpu().numpy().astype(np.int32)
        for i in range(n_inst):
            gens[i].update_agents(p[i], g[i], last[i])
            assert np.array_equal(tokens[i], gens[i].generate_observations()), f"tokens step {t} inst {i}"
            exp_pos, _ = orc.env_step(grid, p[i], g[i], actions[i])
            assert np.array_equal(newpos[i], exp_pos), f"env step {t} inst {i}"
            p[i] = exp_pos
        assert actions.min() >= 0 and actions.max() <= 4
        last = actions.copy()
    m = run.metrics().cpu().numpy()
    assert (m[:, 4] == 10).all()


def test_lifelong_runner_against_host_restatement():
    """on_target="restart": goals advance along each agent's queue when reached, arrivals are counted, episodes never
    terminate early, and the tokenizer follows every goal change (teacher-forced: device actions replayed on the host
    with the oracle's env step + generator)."""
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner
    grid, s_ok, g_ok = maps.load_named("validation-random-seed-000")
    n_inst, n, Q, T = 3, 20, 5, 48
    rng = np.random.Generator(np.random.PCG64(11))
    comp = maps.largest_component(grid == 0)
    free = np.argwhere(comp)
    pos = np.stack([maps.place_agents(grid, n, 100 + i, component=comp)[0] for i in range(n_inst)]).astype(np.int32)
    queue = free[rng.integers(0, len(free), (n_inst, n, Q))].astype(np.int32)          # [inst, agent, Q, 2]
    for i in range(n_inst):                                   # first goals 1-3 cells away so that arrivals do happen
        for a in range(n):
            d = np.abs(free - pos[i, a]).sum(1)
            near = free[(d >= 1) & (d <= 3)]
            queue[i, a, 0] = near[rng.integers(0, len(near))]
    goal = queue[:, :, 0].copy()
    net = build_model("tiny", seed=0, max_rows=n_inst * n, precision="f32")
    run = BatchedRunner(grid, n_inst, n, net, max_episode_steps=T, seed=5, do_sample=True)
    run.reset(torch.from_numpy(pos.astype(np.int16)), torch.from_numpy(goal.astype(np.int16)),
              goal_queue=torch.from_numpy(np.roll(queue, -1, axis=2).astype(np.int16)))   # next goals: entries 1, 2, ..., 0
    nxt = np.roll(queue, -1, axis=2)
    qn = np.zeros((n_inst, n), np.int64)
    reached = np.zeros((n_inst, n), np.int64)
    gens = [orc.OracleGenerator(grid) for _ in range(n_inst)]
    for i in range(n_inst):
        gens[i].create_agents(pos[i], goal[i])
    last = np.full((n_inst, n), -1, np.int32)
    for t in range(T):
        run.step()
        toks = run.tokens.cpu().numpy().reshape(n_inst, n, 256)
        acts = run.actions.cpu().numpy()
        for i in range(n_inst):
            gens[i].update_agents(pos[i], goal[i], last[i])
            assert np.array_equal(toks[i], gens[i].generate_observations()), f"tokens, step {t} instance {i}"
            pos[i], _ = orc.env_step(grid, pos[i], goal[i], acts[i])
            on = (pos[i] == goal[i]).all(1)
            for a in np.nonzero(on)[0]:
                reached[i, a] += 1
                goal[i, a] = nxt[i, a, qn[i, a]]
                qn[i, a] = (qn[i, a] + 1) % Q
        last = acts.copy()
        p_dev, g_dev, done = run.env.sync_state()
        assert np.array_equal(p_dev.cpu().numpy(), pos) and np.array_equal(g_dev.cpu().numpy(), goal), f"state, step {t}"
        assert (done.cpu().numpy() == (2 if t == T - 1 else 0)).all()
    assert np.array_equal(run.env.goals_reached().cpu().numpy(), reached)
    assert reached.sum() > 0


def test_adapter_io_captured_from_the_reference():
    """tests/golden/adapter_io.npz was captured from the reference's own MAPFGPTInference (make_golden_adapter.py): the same
    observation dicts go into OUR adapter with the same deterministic stand-in policy (injected through `net=`, inference.py:48);
    the rows it hands to the policy -- chunk by chunk -- and the action lists it returns must be the reference's."""
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adapter_io.npz"))

    class FakeNet:
        def __init__(self):
            self.chunks = []

        def act(self, idx, do_sample=True, generator=None):
            rows = idx.detach().cpu().to(torch.int64)
            self.chunks.append(rows.numpy().astype(np.uint8))
            a = (rows.sum(1) + 3 * torch.arange(rows.shape[0])) % 5
            return a.reshape(-1, 1).to(idx.device)

    net = FakeNet()
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny", batch_size=int(g["batch_size"])), net=net)
    slots = g["slots"].tolist()
    grids = [g[f"grid{e}"] for e in range(int(g["n_env"]))]
    n_calls = 0
    for k in g["sequence"].tolist():
        if k < 0:
            algo.reset_states()
            continue
        active = g[f"c{k}_active"].tolist()
        obs = []
        for j, e in enumerate(active):
            P, G = g[f"c{k}_pos{j}"], g[f"c{k}_goal{j}"]
            obs.append([{"global_xy": (int(p[0]), int(p[1])), "global_target_xy": (int(q[0]), int(q[1])), "global_obstacles": grids[e].astype(np.int64)}
                        for p, q in zip(P, G)])
        net.chunks = []
        out = [algo.act(obs[0])] if (len(active) == 1 and k == 5) else algo.act_batch(obs, positions=[slots[e] for e in active])
        assert len(net.chunks) == int(g[f"c{k}_nchunks"]), f"call {k}: chunking"
        for j, ch in enumerate(net.chunks):
            assert np.array_equal(ch, g[f"c{k}_chunk{j}"]), f"call {k} chunk {j}: rows handed to the policy"
        for j in range(len(active)):
            assert [int(a) for a in out[j]] == g[f"c{k}_out{j}"].tolist(), f"call {k} env {j}: returned actions"
        n_calls += 1
    assert n_calls == 14
