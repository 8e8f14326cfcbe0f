"""Drop-in for the reference's algorithm adapter (mapf_gpt/inference.py): same class names, same
config fields and defaults, same methods -- `act`, `act_batch`, `reset_states`, `build` -- with the
tokenizer and the policy running as HIP kernels on one MI355X.

What the evaluation harness calls (benchmark.py:22-25, example.py:63-65):
    ToolboxRegistry.register_algorithm("MAPF-GPT", MAPFGPTInference, MAPFGPTInferenceConfig)
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights=..., device=...))
    algo.reset_states(); actions = algo.act(observations)
Differences from the reference, on purpose:
  * no CPU fallback (inference.py:58-67 falls back to cpu/mps): without a HIP device construction raises;
  * weights are read from `path_to_weights` only; the hub download of inference.py:53-56 is attempted
    only if `huggingface_hub` can reach the network, and "synthetic:<shape>[:seed]" builds seeded weights;
  * token rows travel as uint8 device tensors between tokenizer and policy (never Python lists).
"""
from pathlib import Path
from typing import List, Literal, Optional

import numpy as np
import torch
from pydantic import BaseModel, ConfigDict

from . import _lib, weights
from .model import GPT, GPTConfig
from .observation_generator import BatchedTokenizer, InputParameters


class MAPFGPTInferenceConfig(BaseModel):
    """= MAPFGPTInferenceConfig(AlgoBase, extra=forbid), inference.py:13-31 (unknown keys raise, as there).
    `parallel_backend` / `num_process` / `seed` / `preprocessing` stand in for the AlgoBase fields the
    reference's YAML configs set (eval_configs/01-random/01-random.yaml:145-148)."""
    model_config = ConfigDict(extra="forbid")
    name: Literal["MAPF-GPT"] = "MAPF-GPT"
    num_agents: int = 13
    num_previous_actions: int = 5
    cost2go_value_limit: int = 20
    agents_radius: int = 5
    cost2go_radius: int = 5
    path_to_weights: Optional[str] = "weights/MAPF-GPT-2M.pt"
    device: Optional[str] = None
    context_size: int = 256
    mask_actions_history: bool = False
    mask_goal: bool = False
    mask_cost2go: bool = False
    mask_greed_action: bool = False
    repo_id: str = "aandreychuk/MAPF-GPT"
    grid_step: int = 64
    save_cost2go: bool = False
    batch_size: int = 2048
    num_process: int = 8
    # AlgoBase stand-ins
    parallel_backend: Optional[str] = None
    seed: Optional[int] = 0
    preprocessing: Optional[str] = None
    # extension: arithmetic of the policy forward ("f32" exact, "f16x3" split-fp16, "bf16")
    precision: str = "f16x3"     # default = the 1e-5 mode that runs at MFMA speed ("f32" stays selectable)


def strip_prefix_from_state_dict(state_dict, prefix="_orig_mod."):
    """= inference.py:33-44."""
    return weights.strip_prefix(state_dict, prefix)


class MAPFGPTInference:
    def __init__(self, cfg: MAPFGPTInferenceConfig, net=None):
        self.cfg = cfg
        self._obs_generators = {}     # env slot -> BatchedTokenizer (one instance)
        self._last_actions = {}       # env slot -> list[int]
        if self.cfg.device is None:
            self.cfg.device = "cuda"
        if "cuda" not in self.cfg.device:
            raise RuntimeError(f"device '{self.cfg.device}': mapf_gpt_amd runs on a HIP device only (no CPU path)")
        _lib.require_gpu()
        self.torch_generator = torch.Generator(device=self.cfg.device)      # inference.py:69-70
        self.torch_generator.manual_seed(0)
        if net is not None:                                                  # inference.py:79-80
            self.net = net
        else:
            args, sd = self._load_weights()
            self.net = GPT(GPTConfig(**args), max_rows=self.cfg.batch_size, precision=self.cfg.precision,
                           device=self.cfg.device)
            self.net.load_state_dict(sd, strict=False)                       # inference.py:83
            self.net.eval()
        self.input_parameters = InputParameters(                              # inference.py:109-118
            self.cfg.cost2go_value_limit, self.cfg.num_agents, self.cfg.num_previous_actions, self.cfg.context_size,
            self.cfg.cost2go_radius, self.cfg.agents_radius, self.cfg.grid_step, self.cfg.save_cost2go)

    def _load_weights(self):
        p = str(self.cfg.path_to_weights)
        if p.startswith("synthetic:"):
            parts = p.split(":")
            args = weights.model_args(parts[1])
            return args, weights.synthetic_state_dict(args, seed=int(parts[2]) if len(parts) > 2 else 0)
        path = Path(p)
        if not path.exists() and path.name in ("MAPF-GPT-2M.pt", "MAPF-GPT-6M.pt", "MAPF-GPT-85M.pt", "MAPF-GPT-DDG-2M.pt"):
            try:                                                             # inference.py:53-56
                from huggingface_hub import hf_hub_download
                hf_hub_download(repo_id=self.cfg.repo_id, filename=path.name, local_dir=path.parent)
            except Exception as e:  # offline
                raise FileNotFoundError(f"{path} not found and the hub download failed ({e}); "
                                        "place the checkpoint there or use path_to_weights='synthetic:2M'") from e
        return weights.load_checkpoint(path)                                 # inference.py:72-78

    @staticmethod
    def build():
        """= inference.py:121-125 ("pre-build the extension before parallel execution"): make sure the HIP
        library is compiled and loadable."""
        from . import build as _build
        _build.build()
        _lib.lib()

    # ---- per-step path ---------------------------------------------------------------------
    def _prepare_inputs(self, pos, observations):
        """= inference.py:127-146 -> uint8 device tensor [n_agents, 256]."""
        if isinstance(observations[0], dict):
            n = len(observations)
            agent_positions = np.asarray([obs["global_xy"] for obs in observations], dtype=np.int16).reshape(1, n, 2)
            goals = np.asarray([obs["global_target_xy"] for obs in observations], dtype=np.int16).reshape(1, n, 2)
            d_pos = torch.from_numpy(agent_positions).to(self.cfg.device)
            d_goal = torch.from_numpy(goals).to(self.cfg.device)
            if pos not in self._obs_generators:
                grid = np.asarray(observations[0]["global_obstacles"]).copy().astype(int)     # inference.py:135
                gen = BatchedTokenizer(grid, 1, n, self.input_parameters, device=self.cfg.device)
                gen.create_agents(d_pos, d_goal)                                             # inference.py:138
                self._obs_generators[pos] = gen
                self._last_actions[pos] = [-1] * n                                           # inference.py:140
            gen = self._obs_generators[pos]
            act = torch.as_tensor(np.asarray(self._last_actions[pos], dtype=np.int32).reshape(1, n)).to(self.cfg.device)
            gen.update_agents(d_pos, d_goal, act, goals_may_change=True)                     # inference.py:142-144
            return gen.generate_observations()                                               # inference.py:145
        rows = torch.as_tensor(np.asarray(observations, dtype=np.int64))                     # inference.py:146 (pre-tokenised)
        return rows.to(torch.uint8).to(self.cfg.device)

    def _forward_batch(self, inputs):
        """= inference.py:87-101: chunk at batch_size, sample with the adapter's generator."""
        actions: List[int] = []
        for i in range(0, inputs.shape[0], self.cfg.batch_size):
            chunk = inputs[i:i + self.cfg.batch_size]
            out = self.net.act(chunk, generator=self.torch_generator)
            out = torch.atleast_1d(torch.squeeze(out)).tolist()      # an injected net may return [B, 1] like model.py:259
            actions.extend(int(a) for a in out)
        return actions

    def act(self, observations):
        return self.act_batch([observations])[0]                                             # inference.py:148-149

    def act_batch(self, observations_list, positions=None):
        """= inference.py:151-172: rows of many envs -> one forward -> split back per env.
        Host side of the reference-shaped call: the positions, goals and fed-back actions of ALL environments of the call
        travel in ONE host-to-device copy, and every environment's tokenizer writes its rows straight into its slice of one
        [total_rows, 256] tensor (no per-environment copies, no concatenation)."""
        if positions is None:
            positions = list(range(len(observations_list)))
        counts = [len(o) for o in observations_list]
        total = sum(counts)
        tokens = torch.empty((total, 256), dtype=torch.uint8, device=self.cfg.device)
        is_env = [len(o) > 0 and isinstance(o[0], dict) for o in observations_list]
        n_dict = sum(c for c, e in zip(counts, is_env) if e)
        if n_dict:
            # staging buffer: [actions int32 | positions int16 x2 | goals int16 x2] of every dict-observation env, in call order
            buf = np.empty(12 * n_dict, dtype=np.uint8)
            acts, xy, gxy = buf[:4 * n_dict].view(np.int32), buf[4 * n_dict:8 * n_dict].view(np.int16), buf[8 * n_dict:].view(np.int16)
            o = 0
            for pos, observations, n, env in zip(positions, observations_list, counts, is_env):
                if not env:
                    continue
                xy[2 * o:2 * (o + n)] = np.asarray([obs["global_xy"] for obs in observations], dtype=np.int16).reshape(-1)
                gxy[2 * o:2 * (o + n)] = np.asarray([obs["global_target_xy"] for obs in observations], dtype=np.int16).reshape(-1)
                acts[o:o + n] = self._last_actions[pos] if pos in self._obs_generators else -1      # inference.py:140
                o += n
            dev = torch.from_numpy(buf).to(self.cfg.device)
            d_act = dev[:4 * n_dict].view(torch.int32)
            d_xy = dev[4 * n_dict:8 * n_dict].view(torch.int16).view(n_dict, 2)
            d_gxy = dev[8 * n_dict:].view(torch.int16).view(n_dict, 2)
        row, o = 0, 0
        for pos, observations, n, env in zip(positions, observations_list, counts, is_env):
            out = tokens[row:row + n]
            if env:
                p, g, a = d_xy[o:o + n].view(1, n, 2), d_gxy[o:o + n].view(1, n, 2), d_act[o:o + n].view(1, n)
                if pos not in self._obs_generators:                                               # inference.py:133-140
                    grid = np.asarray(observations[0]["global_obstacles"]).copy().astype(int)
                    gen = BatchedTokenizer(grid, 1, n, self.input_parameters, device=self.cfg.device)
                    gen.create_agents(p, g)
                    self._obs_generators[pos] = gen
                    self._last_actions[pos] = [-1] * n
                gen = self._obs_generators[pos]
                gen.update_agents(p, g, a, goals_may_change=True)                                 # inference.py:142-144
                gen.generate_observations(out)                                                    # inference.py:145
                o += n
            elif n:
                out.copy_(self._prepare_inputs(pos, observations))                                # inference.py:146 (pre-tokenised)
            row += n
        all_actions = self._forward_batch(tokens)
        results, offset = [], 0
        for pos, count in zip(positions, counts):
            env_actions = all_actions[offset:offset + count]
            self._last_actions[pos] = list(env_actions)                                      # inference.py:168
            results.append(env_actions)
            offset += count
        return results

    def reset_states(self):
        """= inference.py:174-177."""
        self._obs_generators = {}
        self._last_actions = {}
        self.torch_generator.manual_seed(0)
