"""mgpt_step_run: the whole env step behind one C-ABI call, replayed as a hipGraph.  The graph path must be
indistinguishable from the eager launches (tokens, sampled actions, positions, metrics) -- the episode tests against
the oracle (test_gpu_loop.py) run through the graph path by default, this file pins graph == eager, the re-capture on
reset / buffer change, and the launch-bound speed-up the graph exists for (cfg1)."""
import time

import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps

pytestmark = pytest.mark.gpu


def _pair(model, n_inst, n, precision="f16x3", max_steps=64, map_name="validation-random-seed-000", seed=7):
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner, make_instances
    grid, s_ok, g_ok = maps.load_named(map_name)
    net = build_model(model, seed=0, max_rows=max(64, n_inst * n), precision=precision)
    pos, goal = make_instances(grid, n_inst, n, 0, s_ok, g_ok)
    a = BatchedRunner(grid, n_inst, n, net, max_episode_steps=max_steps, seed=seed, do_sample=True, use_graph=True)
    b = BatchedRunner(grid, n_inst, n, net, max_episode_steps=max_steps, seed=seed, do_sample=True, use_graph=False)
    return a, b, pos, goal


@pytest.mark.parametrize("model,n_inst,n", [("tiny", 3, 20), ("2M", 1, 32), ("6M", 2, 64)])
def test_graph_steps_equal_eager_steps(model, n_inst, n):
    a, b, pos, goal = _pair(model, n_inst, n)
    for episode in range(2):                       # the second episode re-captures after reset
        a.reset(pos, goal)
        b.reset(pos, goal)
        for t in range(14):
            a.step()
            b.step()
            assert torch.equal(a.tokens, b.tokens), f"tokens, episode {episode} step {t}"
            assert torch.equal(a.actions, b.actions), f"actions, episode {episode} step {t}"
            # the device-side step counter draws what the host-argument entry draws with step = t
            want = a.net.act_tokens(a.tokens, do_sample=True, seed=7, step=t, row0=0)
            assert torch.equal(a.actions.view(-1), want), f"device step counter, episode {episode} step {t}"
            assert torch.equal(a.env.sync_state()[0], b.env.sync_state()[0]), f"positions, episode {episode} step {t}"
        assert torch.equal(a.metrics(), b.metrics())
    acts = a.actions.cpu().numpy()
    assert acts.min() >= 0 and acts.max() <= 4 and len(np.unique(acts)) > 1        # sampling is live (not a frozen step counter)


def test_graph_draws_change_with_the_device_step_counter():
    """A frozen RNG step would repeat the same draw for an unchanged observation; with all agents boxed in (every move is
    a wait) the observations repeat, so the sampled actions must still vary from step to step."""
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner
    grid = np.ones((12, 12), np.uint8)
    cells = [(2, 2), (2, 6), (6, 2), (6, 6), (9, 9), (9, 4)]
    for r, c in cells:
        grid[r, c] = 0
    pos = torch.tensor([cells], dtype=torch.int16)
    goal = torch.roll(pos, 1, dims=1)              # unreachable goals: nobody ever finishes, nobody can move
    net = build_model("tiny", seed=0, max_rows=64)
    run = BatchedRunner(grid, 1, len(cells), net, max_episode_steps=64, seed=3, do_sample=True, use_graph=True)
    ref = BatchedRunner(grid, 1, len(cells), net, max_episode_steps=64, seed=3, do_sample=True, use_graph=False)
    run.reset(pos, goal)
    ref.reset(pos, goal)
    seen = []
    for _ in range(12):
        run.step()
        ref.step()
        assert torch.equal(run.actions, ref.actions)
        seen.append(run.actions.cpu().numpy().copy())
    assert len({s.tobytes() for s in seen}) > 3


def test_graph_follows_a_new_token_buffer():
    a, b, pos, goal = _pair("tiny", 2, 16)
    a.reset(pos, goal)
    b.reset(pos, goal)
    for t in range(4):
        a.step()
        b.step()
    a.tokens = torch.empty_like(a.tokens)          # a different device buffer: the recorded graph must not be replayed into the old one
    for t in range(4):
        a.step()
        b.step()
        assert torch.equal(a.tokens, b.tokens) and torch.equal(a.actions, b.actions)


def test_cfg1_graph_replay_timing_is_recorded():
    """cfg1 = one 32-agent instance on the 2M model.  Measured since round 2 (profiles/r02_cfg1_step_trace.txt, BENCH_r04): the step is
    a chain of dependent launches that keeps the GPU 97-99 % busy (32 rows = a few dozen workgroups per kernel: latency of one
    workgroup, not launch cost), so the graph replays at the speed of the eager launches.  Since round 5 the replay is an OPTION
    (BatchedRunner(use_graph=True)), eager is the default; this test records the two timings and only rejects a pathological
    replay (2 x slower): wall-clock gates on a shared box flake, the functional gate is graph == eager above."""
    a, b, pos, goal = _pair("2M", 1, 32, max_steps=100000)
    res = {"graph": [], "eager": []}
    for rep in range(5):                                   # median of alternating repeats: both run at the same speed,
        for tag, run in (("graph", a), ("eager", b)):      # so a single pair would only compare noise
            run.reset(pos, goal)
            run.run(10)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.run(100)
            torch.cuda.synchronize()
            res[tag].append((time.perf_counter() - t0) / 100 * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"cfg1 ms/step (median of 5): graph {med['graph']:.3f}  eager {med['eager']:.3f}  speedup {med['eager'] / med['graph']:.2f}x")
    assert med["graph"] < 2.0 * med["eager"], res          # medians of 5 alternating repeats
