#!/usr/bin/env python3
"""One episode through the reference-shaped list API, in the role of the reference's example.py (example.py:14-75):
env (GridEnv, POGEMA's list API) -> MAPFGPTInference.act -> env.step until the episode ends, then the metrics.

    python example.py --map_name validation-random-seed-000 --num_agents 32 --model 2M [--weights weights/MAPF-GPT-2M.pt]

Without released weights (unreachable offline) `--weights synthetic:2M` (the default) runs a randomly initialised policy.
`--animation` writes the episode as svg/<map>-<model>-seed-<seed>.svg (example.py:59,66-70; own writer, mapf_gpt_amd/animation.py).
"""
import argparse


def run_episode(env, algo):
    """= pogema_toolbox.run_episode as the reference uses it (create_env.py:14-19): act until every agent is terminated or
    truncated, return infos[0]["metrics"]."""
    algo.reset_states()
    obs, _ = env.reset()
    while True:
        obs, rewards, terminated, truncated, infos = env.step(algo.act(obs))
        if all(terminated) or all(truncated):
            return infos[0]["metrics"]


def main():
    ap = argparse.ArgumentParser(description="MAPF-GPT inference on MI355X (list API)")
    ap.add_argument("--num_agents", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--map_name", type=str, default="validation-random-seed-000")
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--max_episode_steps", type=int, default=128)
    ap.add_argument("--show_map_names", action="store_true")
    ap.add_argument("--model", type=str, choices=["2M", "6M", "85M", "DDG-2M"], default="2M")
    ap.add_argument("--weights", type=str, default=None, help="checkpoint path (default: synthetic:<model>)")
    ap.add_argument("--precision", type=str, default="f16x3", choices=["f32", "f16x3", "bf16"])
    ap.add_argument("--animation", action="store_true", help="save the episode as an animated SVG (example.py:30,66-70)")
    a = ap.parse_args()

    from mapf_gpt_amd import maps
    if a.show_map_names:
        print("\n".join(sorted(maps.named_maps())))
        return
    from mapf_gpt_amd.env import GridEnv
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    env = GridEnv(map_name=a.map_name, num_agents=a.num_agents, seed=a.seed, max_episode_steps=a.max_episode_steps,
                  obs_radius=5, on_target="nothing", collision_system="soft")
    shape = "2M" if a.model == "DDG-2M" else a.model
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights=a.weights or f"synthetic:{shape}", device=a.device,
                                                   precision=a.precision))
    if a.animation:
        env.enable_animation()                                                     # example.py:59
    print(run_episode(env, algo))
    if a.animation:
        print("Saved animation to:", env.save_animation(f"svg/{a.map_name}-{a.model}-seed-{a.seed}.svg"))   # example.py:66-70


if __name__ == "__main__":
    main()
