"""HIP env step vs our C restatement (parity of the env is UNPINNED against POGEMA -- see DESIGN.md) + invariants."""
import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n_agents,n_inst", [("validation-random-seed-000", 32, 6), ("validation-mazes-seed-000", 64, 4),
                                                  ("puzzle-00", 4, 9), ("wfi_warehouse", 192, 2)])
def test_env_step_matches_spec_and_invariants(name, n_agents, n_inst):
    from mapf_gpt_amd.env import BatchedEnv
    from mapf_gpt_amd.runner import make_instances
    grid, s_ok, g_ok = maps.load_named(name)
    pos, goal = make_instances(grid, n_inst, n_agents, 0, s_ok, g_ok)
    env = BatchedEnv(grid, n_inst, n_agents, max_episode_steps=40)
    env.reset(pos, goal)
    p = pos.numpy().astype(np.int32).copy()
    g = goal.numpy().astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(11))
    dens = [[orc.agents_density(grid, p[i])] for i in range(n_inst)]          # the reset observation's sample
    was_done = np.zeros(n_inst, bool)
    for t in range(40):
        act = rng.integers(0, 5, (n_inst, n_agents)).astype(np.int32)
        if t % 3 == 0:       # provoke swaps and chains: everybody pushes the same way
            act[:] = rng.integers(1, 5)
        env.step(torch.from_numpy(act).cuda())
        got, _, done = env.sync_state()
        got = got.cpu().numpy().astype(np.int32)
        for i in range(n_inst):
            exp, k = orc.env_step(grid, p[i], g[i], act[i])
            assert np.array_equal(got[i], exp), f"step {t} instance {i}"
            # invariants: no vertex conflict, never on an obstacle, no edge swap, moves of at most one cell
            assert len({tuple(x) for x in exp}) == n_agents
            assert (grid[exp[:, 0], exp[:, 1]] == 0).all()
            assert (np.abs(exp - p[i]).sum(1) <= 1).all()
            old = {tuple(x): a for a, x in enumerate(p[i])}
            for a in range(n_agents):
                b = old.get(tuple(exp[a]))
                if b is not None and b != a:
                    assert tuple(exp[b]) != tuple(p[i][a]), "edge swap"
            if not was_done[i]:                                # the step that ends the episode still samples
                dens[i].append(orc.agents_density(grid, exp))
            p[i] = exp
        was_done = done.cpu().numpy() != 0
    m = env.metrics().cpu().numpy()
    assert np.allclose(m[:, 5], [np.mean(d) for d in dens], rtol=1e-6, atol=0), "avg_agents_density"
    assert (m[:, 4] == 40).all() or (env.done.cpu().numpy() == 1).any()
    on = (p == g).all(2)
    assert np.allclose(m[:, 1], on.mean(1))
    assert (env.done.cpu().numpy() != 0).all()          # truncated (2) or terminated (1) after 40 steps
    frozen = env.sync_state()[0].cpu().numpy().copy()   # done instances ignore further actions
    env.step(torch.from_numpy(rng.integers(0, 5, (n_inst, n_agents)).astype(np.int32)).cuda())
    assert np.array_equal(env.sync_state()[0].cpu().numpy(), frozen)


def test_grid_env_list_api_shape():
    """create_env.py:14-25 call shape: reset -> (obs, info); step -> 5-tuple of per-agent lists; metrics at the end."""
    from mapf_gpt_amd.env import GridEnv
    env = GridEnv(map_name="validation-random-seed-000", num_agents=8, seed=0, max_episode_steps=5)
    obs, info = env.reset()
    assert len(obs) == 8 and set(obs[0]) == {"global_xy", "global_target_xy", "global_obstacles"}
    assert obs[0]["global_obstacles"].shape == (30, 31)
    for t in range(5):
        obs, rew, term, trunc, infos = env.step([0] * 8)
        assert len(rew) == len(term) == len(trunc) == len(infos) == 8
    assert all(trunc) and set(infos[0]["metrics"]) == {"CSR", "ISR", "SoC", "makespan", "ep_length", "avg_agents_density"}
    assert infos[0]["metrics"]["ep_length"] == 5


def test_aec_facade_matches_parallel_api():
    """Agents acting in turn through AECEnv give the same trajectory as one GridEnv.step per cycle."""
    from mapf_gpt_amd.env import AECEnv, GridEnv
    kw = dict(map_name="validation-random-seed-000", num_agents=6, seed=4, max_episode_steps=12)
    ref = GridEnv(**kw)
    obs_ref, _ = ref.reset()
    aec = AECEnv(**kw)
    aec.reset()
    rng = np.random.Generator(np.random.PCG64(1))
    plan = rng.integers(0, 5, (12, 6))
    t, acted, steps_seen = 0, 0, 0
    for agent in aec.agent_iter():
        obs, reward, terminated, truncated, info = aec.last()
        i = aec.possible_agents.index(agent)
        if terminated or truncated:
            aec.step(None)
            continue
        assert obs["global_xy"] == obs_ref[i]["global_xy"] and obs["global_target_xy"] == obs_ref[i]["global_target_xy"]
        aec.step(int(plan[t, i]))
        acted += 1
        if acted == 6:
            obs_ref, _, term, trunc, infos = ref.step(plan[t].tolist())
            acted, t = 0, t + 1
            steps_seen += 1
    assert steps_seen == 12 and not aec.agents
    assert "metrics" in aec.infos["agent_0"] and aec.infos["agent_0"]["metrics"]["ep_length"] == 12


@pytest.mark.parametrize("rules", [1, 2, 3])
def test_rule_switches_match_the_oracle(rules):
    """The two switchable (RECALLED) collision rules -- no following of a leaving agent (1), lowest id wins a contested cell (2)
    -- on the device against the oracle with the same mask, crowded instances, pushes in one direction every third step."""
    from mapf_gpt_amd.env import BatchedEnv
    from mapf_gpt_amd.runner import make_instances
    grid, s_ok, g_ok = maps.load_named("validation-mazes-seed-000")
    n_inst, n_agents = 3, 70
    pos, goal = make_instances(grid, n_inst, n_agents, 5, s_ok, g_ok)
    env = BatchedEnv(grid, n_inst, n_agents, max_episode_steps=64)
    env.set_rules(rules)
    env.reset(pos, goal)
    p, g = pos.numpy().astype(np.int32).copy(), goal.numpy().astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(rules))
    differs = False
    for t in range(30):
        act = rng.integers(0, 5, (n_inst, n_agents)).astype(np.int32)
        if t % 3 == 0:
            act[:] = rng.integers(1, 5)
        env.step(torch.from_numpy(act).cuda())
        got = env.sync_state()[0].cpu().numpy().astype(np.int32)
        for i in range(n_inst):
            exp, _ = orc.env_step(grid, p[i], g[i], act[i], rules=rules)
            assert np.array_equal(got[i], exp), f"rules {rules} step {t} instance {i}"
            differs |= not np.array_equal(exp, orc.env_step(grid, p[i], g[i], act[i])[0])
            p[i] = exp
    assert differs, "the mask must change at least one outcome on this scenario"
    with pytest.raises(RuntimeError):
        env.set_rules(8)


def test_grid_env_records_and_saves_an_episode_animation(tmp_path):
    """enable_animation() / save_animation() of the list-API env (example.py:59,66-70): frames = reset + every step, positions as handed out."""
    import xml.etree.ElementTree as ET
    from mapf_gpt_amd.env import GridEnv
    env = GridEnv(map_name="validation-random-seed-000", num_agents=8, seed=1, max_episode_steps=16)
    with pytest.raises(RuntimeError):
        env.save_animation(str(tmp_path / "x.svg"))
    env.enable_animation()
    obs, _ = env.reset()
    rng = np.random.default_rng(0)
    seen = [[o["global_xy"] for o in obs]]
    for _ in range(5):
        obs, *_ = env.step(rng.integers(0, 5, size=8).tolist())
        seen.append([o["global_xy"] for o in obs])
    assert len(env._frames) == 6 and all(np.array_equal(np.asarray(s), f) for s, f in zip(seen, env._frames))
    root = ET.parse(env.save_animation(str(tmp_path / "ep.svg"))).getroot()
    ns = "{http://www.w3.org/2000/svg}"
    assert len(root.findall(ns + "circle")) == 16
    moved = [c for c in root.findall(ns + "circle") if c.findall(ns + "animate")]
    assert moved and all(len(a.attrib["values"].split(";")) == 6 for c in moved for a in c.findall(ns + "animate"))
    env.reset()
    assert len(env._frames) == 1                                                # a recording covers one episode
