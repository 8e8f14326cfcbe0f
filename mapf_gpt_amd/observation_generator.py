"""Host mirror of the reference's pybind module `observation_generator`
(mapf_gpt/observation_generator.cpp:546-563) over the HIP tokenizer.

Two surfaces:
  * `InputParameters` / `ObservationGenerator` -- the reference's per-env, list-in / list-out classes,
    same constructor arguments, same three methods, same return shape (n rows of 256 Python ints),
    so mapf_gpt/inference.py:127-146 works unchanged against this module;
  * `BatchedTokenizer` -- the device-resident API the batched runner uses: many instances, int16
    tensors in HBM, uint8 token rows out, zero host synchronisation.
Both drive the same kernels through the C ABI (include/mapf_gpt_amd.h); neither has a CPU path.
"""
import ctypes

import numpy as np
import torch

from . import _lib


class InputParameters:
    """= struct InputParameters (observation_generator.h:22-40; pybind ctor cpp:551: 8 positional args)."""

    def __init__(self, cost2go_value_limit=20, num_agents=13, num_previous_actions=5, context_size=256,
                 obs_radius=5, agents_radius=5, grid_step=64, save_cost2go=False):
        self.cost2go_value_limit = int(cost2go_value_limit)
        self.num_agents = int(num_agents)
        self.num_previous_actions = int(num_previous_actions)
        self.context_size = int(context_size)
        self.obs_radius = int(obs_radius)
        self.agents_radius = int(agents_radius)
        self.grid_step = int(grid_step)
        self.save_cost2go = bool(save_cost2go)

    def _struct(self):
        return _lib.InputParametersStruct(self.cost2go_value_limit, self.num_agents, self.num_previous_actions,
                                          self.context_size, self.obs_radius, self.agents_radius, self.grid_step,
                                          int(self.save_cost2go))


class BatchedTokenizer:
    """n_inst env instances x n_agents agents on one padded H x W frame, state resident in HBM.

    grids: uint8 [n_grids, H, W] (non-zero = blocked); instance i uses map i % n_grids.
    All tensors handed in must live on `device`; pos/goal int16 [n_inst, n_agents, 2] (row, col),
    actions int32 [n_inst, n_agents]."""

    def __init__(self, grids, n_inst, n_agents, cfg=None, device="cuda"):
        _lib.require_gpu()
        self.device = torch.device(device)
        cfg = cfg or InputParameters()
        grids = torch.as_tensor(np.ascontiguousarray(grids) if isinstance(grids, np.ndarray) else grids)
        if grids.dim() == 2:
            grids = grids[None]
        self.grids = (grids != 0).to(torch.uint8).contiguous().to(self.device)
        self.n_grids, self.H, self.W = self.grids.shape
        self.n_inst, self.n_agents = int(n_inst), int(n_agents)
        self._h = ctypes.c_void_p()
        st = cfg._struct()
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_create(ctypes.byref(self._h), ctypes.byref(st), self.n_inst,
                                                        self.n_agents, self.H, self.W, self.n_grids))
            _lib.check(_lib.lib().mgpt_tokenizer_set_grids(self._h, _lib.ptr(self.grids), _lib.stream_ptr()))
        self.rows = self.n_inst * self.n_agents

    @property
    def vocab_size(self):
        """Tokens of the Encoder's vocabulary for this configuration (cpp:321-350): 2 * cost2go_value_limit + 27; 67 with the reference's limit."""
        v = ctypes.c_int(0)
        _lib.check(_lib.lib().mgpt_tokenizer_vocab_size(self._h, ctypes.byref(v)))
        return v.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mgpt_tokenizer_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    def _chk(self, t, dtype, shape):
        assert t.dtype == dtype and t.is_cuda and tuple(t.shape) == tuple(shape), (t.dtype, t.shape, shape)
        return t.contiguous()

    def create_agents(self, pos, goal):
        shp = (self.n_inst, self.n_agents, 2)
        pos, goal = self._chk(pos, torch.int16, shp), self._chk(goal, torch.int16, shp)
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_create_agents(self._h, _lib.ptr(pos), _lib.ptr(goal), _lib.stream_ptr()))

    def update_agents(self, pos, goal, actions, goals_may_change=True, active=None):
        """active: optional uint8 [n_inst] device tensor; instances with 0 keep their state (not presented in this call)."""
        shp = (self.n_inst, self.n_agents, 2)
        pos = self._chk(pos, torch.int16, shp)
        goal = self._chk(goal, torch.int16, shp)
        actions = self._chk(actions, torch.int32, (self.n_inst, self.n_agents))
        if active is not None:
            active = self._chk(active, torch.uint8, (self.n_inst,))
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_update_agents_masked(self._h, _lib.ptr(pos), _lib.ptr(goal), _lib.ptr(actions),
                                                                      _lib.ptr(active) if active is not None else None,
                                                                      1 if goals_may_change else 0, _lib.stream_ptr()))

    def generate_observations(self, out=None):
        """-> uint8 [n_inst * n_agents, 256] on the device."""
        if out is None:
            out = torch.empty((self.rows, 256), dtype=torch.uint8, device=self.device)
        else:
            out = self._chk(out, torch.uint8, (self.rows, 256))
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_generate_observations(self._h, _lib.ptr(out), _lib.stream_ptr()))
        return out

    def distance_fields(self):
        """debug/test read-back: uint16 [n_inst, n_agents, H, W] (numpy, host)."""
        out = torch.empty(self.rows * self.H * self.W, dtype=torch.int16, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_copy_state(self._h, _lib.ptr(out), None, _lib.stream_ptr()))
        return out.cpu().numpy().view(np.uint16).reshape(self.n_inst, self.n_agents, self.H, self.W)

    def records(self):
        """debug/test read-back of the 16-byte agent records -> dict of numpy arrays [n_inst, n_agents, ...]."""
        out = torch.empty(self.rows * 16, dtype=torch.uint8, device=self.device)
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_tokenizer_copy_state(self._h, None, _lib.ptr(out), _lib.stream_ptr()))
        raw = out.cpu().numpy().reshape(self.n_inst, self.n_agents, 16)
        xy = raw[..., :8].copy().view(np.int16)
        return {"pos": xy[..., 0:2], "goal": xy[..., 2:4], "hist": raw[..., 8:13], "next": raw[..., 13]}


class ObservationGenerator:
    """= class ObservationGenerator (pybind surface cpp:558-562), one env instance, lists in / lists out.

    grid: list[list[int]] (0 free, non-zero blocked), already padded by the env (inference.py:135);
    positions/goals: list of (row, col); actions: list[int] (previous intended actions, -1 at start).
    Arguments are copied (as pybind's STL casters do); returns fresh Python lists."""

    def __init__(self, grid, cfg):
        self.cfg = cfg
        self._grid = np.ascontiguousarray(np.asarray(grid) != 0, dtype=np.uint8)
        if self._grid.ndim != 2:
            raise ValueError("grid must be 2-D")
        self._tok = None
        self._n = 0

    def _dev(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to("cuda")

    def create_agents(self, positions, goals):
        n = len(positions)
        if len(goals) != n:
            raise ValueError("positions and goals differ in length")
        if self._tok is None or self._n != n:
            self._tok = BatchedTokenizer(self._grid, 1, n, self.cfg)
            self._n = n
        pos = self._dev(np.asarray(positions, dtype=np.int16).reshape(1, n, 2), torch.int16)
        goal = self._dev(np.asarray(goals, dtype=np.int16).reshape(1, n, 2), torch.int16)
        self._tok.create_agents(pos, goal)

    def update_agents(self, positions, goals, actions):
        n = self._n
        if self._tok is None:
            raise RuntimeError("create_agents must be called first")
        if not (len(positions) == len(goals) == len(actions) == n):
            raise ValueError("argument lengths differ from the number of agents")
        pos = self._dev(np.asarray(positions, dtype=np.int16).reshape(1, n, 2), torch.int16)
        goal = self._dev(np.asarray(goals, dtype=np.int16).reshape(1, n, 2), torch.int16)
        act = self._dev(np.asarray(actions, dtype=np.int32).reshape(1, n), torch.int32)
        self._tok.update_agents(pos, goal, act, goals_may_change=True)

    def generate_observations(self):
        if self._tok is None:
            raise RuntimeError("create_agents must be called first")
        return self._tok.generate_observations().cpu().numpy().astype(np.int64).tolist()
