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
    # extension: arithmetic of the policy forward.  Default "f16x3" (split-fp16 MFMA passes, fp32 accumulate: within 1e-5 of the fp32
    # forward, 2-3x faster) UNDER THE ENVELOPE GUARD of the library: a checkpoint whose weight statistics or whose probe rows fall
    # outside the range on which that bar was established is served by the exact-fp32 kernels instead, with one line on stderr
    # (include/mapf_gpt_amd.h: MGPT_ENVELOPE_*; `envelope` below picks "refuse" or "ignore" instead).  Rounds 3-4 defaulted to "f32"
    # because nothing checked the loaded checkpoint.  "f32" = the reference's own arithmetic
    # unconditionally; "bf16" = the reference's autocast class.
    precision: Literal["f32", "f16x3", "bf16"] = "f16x3"
    envelope: Literal["fallback", "refuse", "ignore"] = "fallback"


def strip_prefix_from_state_dict(state_dict, prefix="_orig_mod."):
    """= inference.py:33-44."""
    return weights.strip_prefix(state_dict, prefix)


class _SlotGroup:
    """One tokenizer context for environment slots of equal shape: instance i = slot slots[i], with its own map."""

    def __init__(self, slots, grids, n, params, device):
        self.slots = list(slots)
        self.index = {p: i for i, p in enumerate(self.slots)}
        self.k, self.n = len(self.slots), int(n)
        self.tok = BatchedTokenizer(grids, self.k, self.n, params, device=device)
        self.rows = torch.empty((self.k * self.n, 256), dtype=torch.uint8, device=device)
        self.created = False


class MAPFGPTInference:
    def __init__(self, cfg: MAPFGPTInferenceConfig, net=None):
        self.cfg = cfg
        self._obs_generators = {}     # env slot -> the tokenizer context (shared _SlotGroup) that holds its instance
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
                           device=self.cfg.device, envelope=self.cfg.envelope)
            self.net.load_state_dict(sd, strict=False)                       # inference.py:83
            self.net.eval()
        # (the reference builds Encoder's vocabulary from cost2go_value_limit, cpp:321-350, and its nn.Embedding raises IndexError on ids the
        #  model was not built for; here the ids would index past the embedding table on the device, so the pairing is checked up front)
        vocab = getattr(getattr(self.net, "config", None), "vocab_size", 67)      # (an injected net may be any callable with .act, inference.py:79-80)
        if 2 * self.cfg.cost2go_value_limit + 27 > vocab:
            raise ValueError(f"cost2go_value_limit={self.cfg.cost2go_value_limit} gives a vocabulary of {2 * self.cfg.cost2go_value_limit + 27} tokens; "
                             f"the policy's embedding has {vocab} (model.py:110,126)")
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
        """= inference.py:127-146 -> uint8 device tensor [n_agents, 256] (one environment; act_batch handles many at once)."""
        if isinstance(observations[0], dict):
            return self._tokenize([pos], [observations])
        rows = torch.as_tensor(np.asarray(observations, dtype=np.int64))                     # inference.py:146 (pre-tokenised)
        return rows.to(torch.uint8).to(self.cfg.device)

    def _tokenize(self, positions, observations_list, out=None, offsets=None):
        """Token rows of dict-observation environments -> `out` (uint8 [total, 256], environment e at rows offsets[e] ..).

        Environment slots of equal shape (map frame H x W, agent count) that first appear in the same call share ONE tokenizer
        context (`_SlotGroup`: one instance per slot, every instance its own map), so a call costs one host-to-device copy, one
        update launch and one tokens launch per group instead of two launches and a dozen small copies per environment
        (inference.py:133-145 keeps one ObservationGenerator per slot: same state, kept per instance here).  A later call may
        present any subset of a group's slots: absent instances are masked out of the update and keep their state."""
        dev = self.cfg.device
        if len(set(positions)) != len(positions):
            # the reference would update one generator twice; here the second entry would overwrite the first one's staging
            raise ValueError("act_batch: every environment slot (positions entry) may appear once per call")
        counts = [len(o) for o in observations_list]
        if offsets is None:
            offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).tolist()
        if out is None:
            out = torch.empty((sum(counts), 256), dtype=torch.uint8, device=dev)
        # new slots: one group per (H, W, n) among the slots this call introduces
        fresh = {}
        for pos, obs in zip(positions, observations_list):
            if pos not in self._obs_generators:
                grid = np.asarray(obs[0]["global_obstacles"])                                # inference.py:135
                # .astype(int) first: the reference truncates (0.5 -> free cell) before the generator tests for non-zero
                fresh.setdefault((grid.shape[0], grid.shape[1], len(obs)), []).append((pos, (grid.astype(int) != 0).astype(np.uint8)))
        for (H, W, n), members in fresh.items():
            grp = _SlotGroup([p for p, _ in members], np.stack([g for _, g in members]), n, self.input_parameters, dev)
            for p in grp.slots:
                self._obs_generators[p] = grp
                self._last_actions[p] = [-1] * n                                             # inference.py:140
        # group the call's environments by context
        touched = {}
        for e, pos in enumerate(positions):
            touched.setdefault(id(self._obs_generators[pos]), (self._obs_generators[pos], []))[1].append(e)
        # one staging buffer for the whole call: per group [actions int32 k*n | positions int16 k*n*2 | goals int16 k*n*2 | active u8 k (+pad)]
        plan, nbytes = [], 0
        for grp, envs in touched.values():
            kn = grp.k * grp.n
            plan.append((grp, envs, nbytes))
            nbytes += 12 * kn + ((grp.k + 15) & ~15)
        buf = np.zeros(nbytes, dtype=np.uint8)
        for grp, envs, base in plan:
            kn, n = grp.k * grp.n, grp.n
            acts = buf[base:base + 4 * kn].view(np.int32)
            xy = buf[base + 4 * kn:base + 8 * kn].view(np.int16)
            gxy = buf[base + 8 * kn:base + 12 * kn].view(np.int16)
            active = buf[base + 12 * kn:base + 12 * kn + grp.k]
            for e in envs:
                i, obs = grp.index[positions[e]], observations_list[e]
                xy[2 * i * n:2 * (i + 1) * n] = np.asarray([o["global_xy"] for o in obs], dtype=np.int16).reshape(-1)       # inference.py:130-131
                gxy[2 * i * n:2 * (i + 1) * n] = np.asarray([o["global_target_xy"] for o in obs], dtype=np.int16).reshape(-1)
                acts[i * n:(i + 1) * n] = self._last_actions[positions[e]]
                active[i] = 1
        d = torch.from_numpy(buf).to(dev)
        for grp, envs, base in plan:
            kn, n, k = grp.k * grp.n, grp.n, grp.k
            d_act = d[base:base + 4 * kn].view(torch.int32).view(k, n)
            d_xy = d[base + 4 * kn:base + 8 * kn].view(torch.int16).view(k, n, 2)
            d_gxy = d[base + 8 * kn:base + 12 * kn].view(torch.int16).view(k, n, 2)
            everyone = len(envs) == k
            if not grp.created:                                                              # inference.py:138 (all of a new group's
                grp.tok.create_agents(d_xy, d_gxy)                                           #  slots are in the call that creates it)
                grp.created = True
            grp.tok.update_agents(d_xy, d_gxy, d_act, goals_may_change=True,                 # inference.py:142-144
                                  active=None if everyone else d[base + 12 * kn:base + 12 * kn + k])
            inst = [grp.index[positions[e]] for e in envs]
            row0 = offsets[envs[0]]
            if everyone and inst == list(range(k)) and all(offsets[e] == row0 + j * n for j, e in enumerate(envs)):
                grp.tok.generate_observations(out[row0:row0 + kn])                           # inference.py:145, straight into place
            else:
                rows = grp.tok.generate_observations(grp.rows)
                src = torch.as_tensor(np.concatenate([np.arange(i * n, (i + 1) * n) for i in inst]), device=dev)
                dst = torch.as_tensor(np.concatenate([np.arange(offsets[e], offsets[e] + n) for e in envs]), device=dev)
                out[dst] = rows[src]
        return out

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
        """= inference.py:151-172: rows of many envs -> one forward -> split back per env."""
        if positions is None:
            positions = list(range(len(observations_list)))
        counts = [len(o) for o in observations_list]
        offsets = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64).tolist() if counts else []
        tokens = torch.empty((sum(counts), 256), dtype=torch.uint8, device=self.cfg.device)
        is_env = [len(o) > 0 and isinstance(o[0], dict) for o in observations_list]
        env_ids = [e for e, f in enumerate(is_env) if f]
        if env_ids:
            self._tokenize([positions[e] for e in env_ids], [observations_list[e] for e in env_ids], out=tokens,
                           offsets=[offsets[e] for e in env_ids])
        for e, (pos, observations, n) in enumerate(zip(positions, observations_list, counts)):
            if n and not is_env[e]:
                tokens[offsets[e]:offsets[e] + n].copy_(self._prepare_inputs(pos, observations))   # inference.py:146 (pre-tokenised)
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
