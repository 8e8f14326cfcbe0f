"""Environment surface (the role POGEMA plays for the reference: experiment_setup/create_env.py).

  * `BatchedEnv`   -- device-resident: n_inst instances stepped by one HIP kernel launch.
  * `GridEnv`      -- the list API the reference drives (create_env.py:14-25, example.py:60-65):
                      reset() -> (obs_list, info); step(actions) -> (obs, rewards, terminated,
                      truncated, infos); observations are dicts with the three keys the adapter reads
                      (inference.py:130-135): global_xy, global_target_xy, global_obstacles.
PARITY UNPINNED: POGEMA itself is not in the reference tree; semantics are the spec in DESIGN.md.
"""
import ctypes

import numpy as np
import torch

from . import _lib, maps

METRIC_KEYS = ("CSR", "ISR", "SoC", "makespan", "ep_length", "avg_agents_density")   # eval_configs/*/*.yaml results_views
RULE_NO_FOLLOW, RULE_LOWEST_WINS = 1, 2      # include/mapf_gpt_amd.h: MGPT_ENV_RULE_* (the two switchable, RECALLED collision rules)


class BatchedEnv:
    def __init__(self, grids, n_inst, n_agents, max_episode_steps=128, device="cuda"):
        _lib.require_gpu()
        self.device = torch.device(device)
        grids = torch.as_tensor(np.ascontiguousarray(grids) if isinstance(grids, np.ndarray) else grids)
        if grids.dim() == 2:
            grids = grids[None]
        self.grids = (grids != 0).to(torch.uint8).contiguous().to(self.device)
        self.n_grids, self.H, self.W = self.grids.shape
        self.n_inst, self.n_agents, self.max_episode_steps = int(n_inst), int(n_agents), int(max_episode_steps)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_create(ctypes.byref(self._h), self.n_inst, self.n_agents, self.H, self.W,
                                                  self.n_grids, self.max_episode_steps))
            _lib.check(_lib.lib().mgpt_env_set_grids(self._h, _lib.ptr(self.grids), _lib.stream_ptr()))
        shp = (self.n_inst, self.n_agents, 2)
        # caller-visible mirrors of the env state, refreshed by sync_state()
        self.pos = torch.empty(shp, dtype=torch.int16, device=self.device)
        self.goal = torch.empty(shp, dtype=torch.int16, device=self.device)
        self.done = torch.zeros((self.n_inst,), dtype=torch.uint8, device=self.device)
        self.lifelong = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mgpt_env_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    def reset(self, pos, goal):
        shp = (self.n_inst, self.n_agents, 2)
        pos = pos.to(self.device, torch.int16).contiguous()
        goal = goal.to(self.device, torch.int16).contiguous()
        assert tuple(pos.shape) == shp and tuple(goal.shape) == shp
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_reset(self._h, _lib.ptr(pos), _lib.ptr(goal), _lib.stream_ptr()))
        self.sync_state()

    def set_lifelong(self, goal_queue):
        """on_target="restart": goal_queue int16 [n_inst, n_agents, Q, 2] (None -> back to "nothing").  Call before reset()."""
        with torch.cuda.device(self.device):
            if goal_queue is None:
                _lib.check(_lib.lib().mgpt_env_set_lifelong(self._h, None, 0, _lib.stream_ptr()))
                self.lifelong = False
                return
            q = goal_queue.to(self.device, torch.int16).contiguous()
            assert q.dim() == 4 and tuple(q.shape[:2]) == (self.n_inst, self.n_agents) and q.shape[3] == 2
            _lib.check(_lib.lib().mgpt_env_set_lifelong(self._h, _lib.ptr(q), int(q.shape[2]), _lib.stream_ptr()))
            self.lifelong = True

    def set_rules(self, rules=0):
        """Collision-rule switches (bit mask of RULE_*; 0 = the spec of DESIGN.md section 4)."""
        _lib.check(_lib.lib().mgpt_env_set_rules(self._h, int(rules)))

    def goals_reached(self):
        """int32 [n_inst, n_agents]: arrivals so far (lifelong mode)."""
        out = torch.empty((self.n_inst, self.n_agents), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_lifelong_counts(self._h, _lib.ptr(out), _lib.stream_ptr()))
        return out

    def step(self, actions):
        assert actions.dtype == torch.int32 and actions.is_cuda and actions.numel() == self.n_inst * self.n_agents
        actions = actions.contiguous()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_step(self._h, _lib.ptr(actions), _lib.stream_ptr()))

    def state_ptrs(self):
        p, g, d = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.lib().mgpt_env_state(self._h, ctypes.byref(p), ctypes.byref(g), ctypes.byref(d)))
        return p, g, d

    def sync_state(self):
        """Refresh self.pos / self.goal / self.done (device tensors) from the library's state (async D2D)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_copy_state(self._h, _lib.ptr(self.pos), _lib.ptr(self.goal), _lib.ptr(self.done),
                                                      _lib.stream_ptr()))
        return self.pos, self.goal, self.done

    def metrics(self):
        """float32 [n_inst, 6] = CSR, ISR, SoC, makespan, ep_length, avg_agents_density (device tensor)."""
        out = torch.empty((self.n_inst, len(METRIC_KEYS)), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_env_metrics(self._h, _lib.ptr(out), _lib.stream_ptr()))
        return out


class GridEnv:
    """Single-instance env with the reference's list API (create_env.py:14-25), backed by BatchedEnv.

    cfg keys follow the reference's Environment config (example.py:41-50): map (padded uint8 grid or a
    named map), num_agents, seed, max_episode_steps, obs_radius (must be 5), on_target ("nothing"),
    collision_system ("soft")."""

    def __init__(self, map_name=None, grid=None, num_agents=32, seed=0, max_episode_steps=128, obs_radius=5,
                 on_target="nothing", collision_system="soft", device="cuda"):
        if obs_radius != 5 or on_target != "nothing" or collision_system != "soft":
            raise NotImplementedError("only obs_radius=5, on_target='nothing', collision_system='soft' are implemented")
        if grid is None:
            self.grid, self.start_ok, self.goal_ok = maps.load_named(map_name)
        else:
            self.grid = maps.pad(np.asarray(grid, dtype=np.uint8))
            self.start_ok = self.goal_ok = None
        self.num_agents, self.seed, self.max_episode_steps = num_agents, seed, max_episode_steps
        self._env = BatchedEnv(self.grid, 1, num_agents, max_episode_steps, device=device)
        self._obst = self.grid.astype(np.float32)

    def _pull(self, actions=None):
        """One library call: (optional) step with host actions, then positions / goals / done back on the host."""
        e, n = self._env, self.num_agents
        if not hasattr(self, "_host"):
            self._host = np.empty(8 * n + 8, dtype=np.uint8)
        act = None
        if actions is not None:
            act = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
            assert act.size == n
        with _lib.on_device(e.device):
            _lib.check(_lib.lib().mgpt_env_step_host(e._h, ctypes.c_void_p(act.ctypes.data) if act is not None else None,
                                                     ctypes.c_void_p(self._host.ctypes.data), _lib.stream_ptr()))
        host = self._host
        pos, goal = host[:4 * n].view(np.int16).reshape(n, 2), host[4 * n:8 * n].view(np.int16).reshape(n, 2)
        if getattr(self, "_frames", None) is not None:         # enable_animation(): keep what is handed out anyway
            self._frames.append(pos.copy())
            self._goal_frames.append(goal.copy())
        return pos, goal, int(host[8 * n])

    def enable_animation(self):
        """= env.enable_animation() of the reference's example (example.py:59): record the episode for save_animation."""
        self._frames, self._goal_frames = [], []

    def save_animation(self, path, seconds_per_step=0.25):
        """= env.save_animation(svg_path) (example.py:67-69): the episode since the last reset as one animated SVG (mapf_gpt_amd/animation.py)."""
        from . import animation
        if getattr(self, "_frames", None) is None:
            raise RuntimeError("enable_animation() first")
        return animation.write_svg(path, self.grid, self._frames, self._goal_frames, seconds_per_step)

    def _obs(self, pos, goal):
        p, g = pos.tolist(), goal.tolist()
        return [{"global_xy": (p[a][0], p[a][1]), "global_target_xy": (g[a][0], g[a][1]), "global_obstacles": self._obst}
                for a in range(self.num_agents)]

    def reset(self, seed=None, **kwargs):
        if seed is not None:
            self.seed = seed
        pos, goal = maps.place_agents(self.grid, self.num_agents, self.seed, self.start_ok, self.goal_ok)
        self._env.reset(torch.from_numpy(pos[None]), torch.from_numpy(goal[None]))
        if getattr(self, "_frames", None) is not None:
            self._frames, self._goal_frames = [], []           # a recording covers one episode
        pos, goal, _ = self._pull()
        return self._obs(pos, goal), {}

    def step(self, actions):
        pos, goal, done = self._pull(actions)
        obs = self._obs(pos, goal)
        terminated = [done == 1] * self.num_agents
        truncated = [done == 2] * self.num_agents
        rewards = (pos == goal).all(axis=1).astype(np.float64).tolist()
        infos = [{} for _ in range(self.num_agents)]
        if done:
            m = self._env.metrics().cpu().numpy()[0]
            infos[0]["metrics"] = {k: float(v) for k, v in zip(METRIC_KEYS, m)}   # create_env.py:18-20
        return obs, rewards, terminated, truncated, infos


class AECEnv:
    """Agent-environment-cycle façade over GridEnv (the PettingZoo-style surface BASELINE.json's north star names; the
    reference itself drives POGEMA through the parallel list API, create_env.py:14-15).  Agents act in turn; the underlying
    env steps once every agent of the cycle has supplied its action (all moves of a MAPF step are simultaneous).

        env = AECEnv(map_name=..., num_agents=...); env.reset(seed=0)
        for agent in env.agent_iter():
            obs, reward, terminated, truncated, info = env.last()
            env.step(None if terminated or truncated else policy(obs))
    """

    def __init__(self, **grid_env_kwargs):
        self._env = GridEnv(**grid_env_kwargs)
        self.possible_agents = [f"agent_{i}" for i in range(self._env.num_agents)]
        self.agents = []

    def reset(self, seed=None, options=None):
        self._obs, _ = self._env.reset(seed=seed)
        n = self._env.num_agents
        self.agents = list(self.possible_agents)
        self.rewards = {a: 0.0 for a in self.agents}
        self.terminations = {a: False for a in self.agents}
        self.truncations = {a: False for a in self.agents}
        self.infos = {a: {} for a in self.agents}
        self._pending = [0] * n
        self._cursor = 0
        self.agent_selection = self.agents[0]

    def observe(self, agent):
        return self._obs[self.possible_agents.index(agent)]

    def last(self):
        a = self.agent_selection
        return self.observe(a), self.rewards[a], self.terminations[a], self.truncations[a], self.infos[a]

    def agent_iter(self, max_iter=2 ** 62):
        it = 0
        while self.agents and it < max_iter:
            yield self.agent_selection
            it += 1

    def step(self, action):
        a = self.agent_selection
        i = self.possible_agents.index(a)
        if self.terminations[a] or self.truncations[a]:
            if action is not None:
                raise ValueError("a finished agent must be stepped with None")
            self.agents.remove(a)                                     # PettingZoo: dead agents leave on their None step
            if self.agents:
                self._cursor %= len(self.agents)
                self.agent_selection = self.agents[self._cursor]
            return
        self._pending[i] = 0 if action is None else int(action)
        self._cursor += 1
        if self._cursor == len(self.agents):                          # cycle complete: one simultaneous env step
            obs, rew, term, trunc, infos = self._env.step(self._pending)
            self._obs = obs
            for j, ag in enumerate(self.possible_agents):
                self.rewards[ag], self.terminations[ag], self.truncations[ag], self.infos[ag] = rew[j], term[j], trunc[j], infos[j]
            self._cursor = 0
        self.agent_selection = self.agents[self._cursor]
