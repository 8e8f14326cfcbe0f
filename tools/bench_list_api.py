#!/usr/bin/env python3
"""tools/bench_list_api.py -- the reference-shaped list API end to end (host lists in, host lists out, PCIe + Python
conversion inside the timing): GridEnv.step -> MAPFGPTInference.act_batch per step, like example.py:60-66."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mapf_gpt_amd.env import GridEnv
from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig

def run(map_name, n_agents, n_envs, steps=24, precision="f16x3", freeze=False):
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:2M", batch_size=4096, precision=precision))
    envs = [GridEnv(map_name=map_name, num_agents=n_agents, seed=s, max_episode_steps=10 ** 6) for s in range(n_envs)]
    algo.reset_states()
    obs = [e.reset()[0] for e in envs]
    for _ in range(3):
        acts = algo.act_batch(obs)
        obs = [e.step(a)[0] for e, a in zip(envs, acts)]
    if freeze:          # what a long-running evaluation process does after start-up: the objects alive now (model, envs, torch) leave the
        import gc       # collector's generations, so the ~12 k short-lived dicts / tuples of a step stop triggering full scans of them
        gc.collect(); gc.freeze()
    torch.cuda.synchronize(); t0 = time.perf_counter(); t_act = 0.0
    for _ in range(steps):
        ta = time.perf_counter()
        acts = algo.act_batch(obs)
        t_act += time.perf_counter() - ta
        obs = [e.step(a)[0] for e, a in zip(envs, acts)]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows = n_envs * n_agents * steps
    print(f"{map_name} {n_envs} envs x {n_agents} agents{' (gc.freeze() after start-up)' if freeze else ''}: {rows / dt:9.0f} agent-steps/s end to end "
          f"({rows / t_act:9.0f} in act_batch alone; host env step {1e3 * (dt - t_act) / steps:.2f} ms/step)", flush=True)

def run_workers(map_name, n_agents, n_envs, n_workers=2, steps=24, precision="f16x3"):
    """The reference evaluates with `num_process` workers, each owning a MAPFGPTInference and a share of the environments
    (inference.py:30-31, eval_configs/*.yaml `num_process`).  Here the workers are threads of one process sharing the GPU: while one
    waits for its launch (the library calls and torch's synchronisations drop the GIL) the other steps its environments on the
    host.  Aggregate agent-steps/s over all workers, every worker running the plain act_batch / step loop of run()."""
    import threading, gc
    per = n_envs // n_workers
    algos = [MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:2M", batch_size=4096, precision=precision)) for _ in range(n_workers)]
    envs = [[GridEnv(map_name=map_name, num_agents=n_agents, seed=w * per + s, max_episode_steps=10 ** 6) for s in range(per)] for w in range(n_workers)]
    obs = []
    for w in range(n_workers):
        algos[w].reset_states()
        o = [e.reset()[0] for e in envs[w]]
        for _ in range(3):
            acts = algos[w].act_batch(o)
            o = [e.step(a)[0] for e, a in zip(envs[w], acts)]
        obs.append(o)
    gc.collect(); gc.freeze()
    start = threading.Barrier(n_workers + 1)

    def loop(w):
        o = obs[w]
        torch.cuda.set_device(0)
        start.wait()
        for _ in range(steps):
            acts = algos[w].act_batch(o)
            o = [e.step(a)[0] for e, a in zip(envs[w], acts)]

    threads = [threading.Thread(target=loop, args=(w,)) for w in range(n_workers)]
    for t in threads: t.start()
    torch.cuda.synchronize(); start.wait(); t0 = time.perf_counter()
    for t in threads: t.join()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows = n_workers * per * n_agents * steps
    print(f"{map_name} {n_workers} workers x {per} envs x {n_agents} agents (gc.freeze() after start-up): {rows / dt:9.0f} agent-steps/s end to end, all workers", flush=True)

if __name__ == "__main__":
    run("validation-random-seed-000", 32, 1)
    run("validation-mazes-seed-000", 64, 16)
    run("validation-mazes-seed-000", 64, 64)
    run("validation-mazes-seed-000", 64, 64, freeze=True)
    run_workers("validation-mazes-seed-000", 64, 64, n_workers=2)
    run_workers("validation-mazes-seed-000", 64, 128, n_workers=2)
    run_workers("validation-mazes-seed-000", 64, 128, n_workers=4)
