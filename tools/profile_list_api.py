import sys, cProfile, pstats, io, time
sys.path.insert(0, '/root/repo')
import torch
from mapf_gpt_amd.env import GridEnv
from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:2M", batch_size=4096, precision="f16x3"))
env = GridEnv(map_name="validation-random-seed-000", num_agents=32, seed=0, max_episode_steps=10**6)
algo.reset_states(); obs = env.reset()[0]
for _ in range(20):
    obs = env.step(algo.act(obs))[0]
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(300):
    obs = env.step(algo.act(obs))[0]
dt = time.perf_counter() - t0
pr.disable()
print("per step ms", dt / 300 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
