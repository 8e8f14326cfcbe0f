"""Dataset-side bulk tokenizer on the device: host mirror of the reference's
`dataset/tokenizer/generate_observations.py:ObservationGenerator` (same constructor arguments, same
`generate_observations(start_range, end_range) -> (inputs, gt_actions)`), over `mgpt_dataset_*`.

    gen = ObservationGenerator(maps, data, cfg)          # maps: {name: map string}, data: list of logged instances
    inputs, gt_actions = gen.generate_observations(0, len(data))

`data[i]` = {"metrics": {"CSR", "made_actions", "init_positions"}, "env_grid_search": {"map_name"}} (the
toolbox's result records, generate_observations.py:43-66).  Instances with CSR < 1 are skipped (:44-45); the
all-pairs distance table of a map is built once and reused while consecutive instances share the map (:46-54).
Lifelong logs ("global_lifelong_targets_xy", :55-60, 143-153) and the `mask_cost2go` ablation (:253-262) are handled on the
device like the rest; the other three mask_* fields are carried by `InputParameters` but, as in the reference's dataset
pipeline (whose C++ encoder ignores them), only the python `Encoder.mask` applies them (tokenizer.py:104-138; restated for the tests in oracle/dataset_encoder.py).
Not supported: cost2go_radius != 5.
"""
import ctypes

import numpy as np
import torch

from . import _lib, maps as _maps

MOVES = np.array([[0, 0], [-1, 0], [1, 0], [0, -1], [0, 1]], dtype=np.int32)      # generate_observations.py:10


class InputParameters:
    """= dataset/tokenizer/parameters.py (field names and defaults)."""

    def __init__(self, num_agents=13, num_previous_actions=5, agents_radius=5, cost2go_value_limit=20, cost2go_radius=5,
                 context_size=256, mask_greed_action=False, mask_actions_history=False, mask_goal=False, mask_cost2go=False):
        if (num_agents, num_previous_actions, agents_radius, cost2go_value_limit, cost2go_radius, context_size) != (13, 5, 5, 20, 5, 256):
            raise NotImplementedError("only the reference's defaults (13, 5, 5, 20, 5, 256) are implemented")
        self.mask_greed_action, self.mask_actions_history = bool(mask_greed_action), bool(mask_actions_history)
        self.mask_goal, self.mask_cost2go = bool(mask_goal), bool(mask_cost2go)
        self.num_agents, self.num_previous_actions, self.agents_radius = num_agents, num_previous_actions, agents_radius
        self.cost2go_value_limit, self.cost2go_radius, self.context_size = cost2go_value_limit, cost2go_radius, context_size


class MapTable:
    """All-pairs BFS table of one padded map on the device (the reference's cost2go_data, cost2go.cpp:33-42)."""

    def __init__(self, grid_padded, device="cuda"):
        _lib.require_gpu()
        self.device = torch.device(device)
        g = torch.as_tensor(np.ascontiguousarray(np.asarray(grid_padded) != 0, dtype=np.uint8)).to(self.device)
        self.H, self.W = g.shape
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_dataset_create(ctypes.byref(self._h), _lib.ptr(g), self.H, self.W, _lib.stream_ptr()))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mgpt_dataset_destroy(h)
            except Exception:
                pass
            self._h = None

    def tokenize(self, paths, goals=None, mask_cost2go=False):
        """paths int16 [n_agents, n_steps, 2] (padded coords) -> uint8 device tensor [n_agents, n_steps, 256].
        goals (lifelong logs): int16 [n_agents, n_steps, 2], the goal pursued at every timestep; mask_cost2go: the
        window shows blocked / free only (cost2go.cpp:52-62)."""
        p = torch.as_tensor(np.ascontiguousarray(paths, dtype=np.int16)).to(self.device)
        n, T1 = int(p.shape[0]), int(p.shape[1])
        g = None
        if goals is not None:
            g = torch.as_tensor(np.ascontiguousarray(goals, dtype=np.int16)).to(self.device)
            assert tuple(g.shape) == (n, T1, 2), "goals must have the shape of paths"
        out = torch.empty((n, T1, 256), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_dataset_tokenize_ex(self._h, n, T1, _lib.ptr(p), _lib.ptr(g) if g is not None else None,
                                                           1 if mask_cost2go else 0, _lib.ptr(out), _lib.stream_ptr()))
        return out


def agent_paths(init_positions, made_actions):
    """= get_agent_paths (:159-177): [n][T+1][2] cells after every logged action."""
    acts = np.asarray(made_actions, dtype=np.int64)
    steps = MOVES[acts]                                                 # [n, T, 2]
    p0 = np.asarray(init_positions, dtype=np.int32)[:, None, :]
    return np.concatenate([p0, p0 + np.cumsum(steps, axis=1)], axis=1)


def goal_positions(paths, targets):
    """= get_goal_positions (:143-153): the goal at every path cell = the first target of the agent's list it has not stood
    on yet (standing on it advances the list before the cell's goal is read).  -> int32 [n, T+1, 2].
    Like the reference, a path that exhausts its list raises IndexError."""
    out = np.empty_like(np.asarray(paths, dtype=np.int32))
    for a, (path, tg) in enumerate(zip(paths, targets)):
        cur = 0
        for t, pos in enumerate(path):
            if int(pos[0]) == int(tg[cur][0]) and int(pos[1]) == int(tg[cur][1]):
                cur += 1
            out[a, t] = (int(tg[cur][0]), int(tg[cur][1]))
    return out


def gt_actions(made_actions):
    """Labels of generate_observations.py:67-90: the logged action, one appended wait, 5 = "wait in goal" after the last move."""
    out = []
    for acts in made_actions:
        a = list(acts) + [0]
        nz = [i for i, v in enumerate(a) if v != 0]
        goal_t = nz[-1] if nz else len(a)
        out.append([5 if t > goal_t else a[t] for t in range(len(a))])
    return out


class ObservationGenerator:
    def __init__(self, maps, data, cfg=None, device="cuda"):
        self.cfg = cfg or InputParameters()
        self.maps, self.data, self.device = maps, data, device
        self.inputs, self.gt_actions = [], []
        self._table, self._table_name = None, None

    def get_grid_map(self, map_name):
        """= :105-119: '.'/'#' string -> padded 0/1 array."""
        rows = [r for r in self.maps[map_name].split() if r]
        for i, r in enumerate(rows):
            bad = set(r) - {".", "#"}
            if bad:
                raise KeyError(f"Unsupported symbol '{sorted(bad)[0]}' at line {i}")            # :101
        obst = np.array([[1 if ch == "#" else 0 for ch in r] for r in rows], dtype=np.uint8)
        return _maps.pad(obst, self.cfg.cost2go_radius, 1)

    def generate_observations(self, start_range, end_range):
        self.inputs, self.gt_actions = [], []
        for instance_id in range(start_range, end_range):
            rec = self.data[instance_id]
            if rec["metrics"].get("CSR", 1) < 1:                        # :44-45
                continue
            name = rec["env_grid_search"]["map_name"]
            if name != self._table_name:                                # :46-54
                self._table, self._table_name = MapTable(self.get_grid_map(name), self.device), name
            paths = agent_paths(rec["metrics"]["init_positions"], rec["metrics"]["made_actions"])
            goals = None
            if "global_lifelong_targets_xy" in rec["metrics"]:          # :55-60
                goals = goal_positions(paths, rec["metrics"]["global_lifelong_targets_xy"])
            toks = self._table.tokenize(paths, goals, self.cfg.mask_cost2go).cpu().numpy().astype(np.int8).reshape(-1, 256)
            self.inputs.extend(list(toks))
            for g in gt_actions(rec["metrics"]["made_actions"]):
                self.gt_actions.extend(g)
        return self.inputs, self.gt_actions
