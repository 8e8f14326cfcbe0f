#!/usr/bin/env python3
"""tools/time_gridenv.py -- where a GridEnv.step of the list API spends its time (64 agents, maze map): the library call
(H2D actions, step kernel, D2H state, sync), the list-of-dict observations, the rest."""
import time
import numpy as np
from mapf_gpt_amd.env import GridEnv

env = GridEnv(map_name="validation-mazes-seed-000", num_agents=64, seed=0, max_episode_steps=100000)
obs, _ = env.reset()
acts = [1] * 64
for _ in range(50):
    env.step(acts)
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    env.step(acts)
t_step = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    pos, goal, done = env._pull(acts)
t_pull = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    env._obs(pos, goal)
t_obs = (time.perf_counter() - t0) / N
a = np.asarray(acts, dtype=np.int32)
t0 = time.perf_counter()
for _ in range(N):
    pos, goal, done = env._pull(a)
t_pull_np = (time.perf_counter() - t0) / N
print("GridEnv.step %.1f us = library call %.1f us (%.1f with a ready int32 array) + observations %.1f us + rest %.1f us" %
      (t_step * 1e6, t_pull * 1e6, t_pull_np * 1e6, t_obs * 1e6, (t_step - t_pull - t_obs) * 1e6))
