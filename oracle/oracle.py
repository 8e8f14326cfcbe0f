"""ctypes wrapper over oracle/_build/libmapf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (mapf_gpt_amd/) never imports this module.

`OracleGenerator` mirrors the reference's pybind class ObservationGenerator
(mapf_gpt/observation_generator.cpp:558-562): same three calls, list/array in, uint8 rows out.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmapf_oracle.so")
_lib = None


def build(force=False):
    """gcc-compile the C restatement (seconds)."""
    src = os.path.join(_HERE, "mapf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def build_ref():
    """Compile the REAL reference tokenizer into oracle/_ref (only where /root/reference exists)."""
    if not os.path.isdir("/root/reference/mapf_gpt"):
        return None
    subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])
    return os.path.join(_HERE, "_ref")


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, u8p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
        L.orc_bfs.argtypes = [u8p, i32, i32, i32, i32, vp]
        L.orc_bfs.restype = None
        L.orc_gen_create.argtypes = [u8p, i32, i32]
        L.orc_gen_create.restype = vp
        L.orc_gen_destroy.argtypes = [vp]
        L.orc_gen_set_grid_step.argtypes = [vp, i32]
        L.orc_gen_set_grid_step.restype = None
        L.orc_gen_set_params.argtypes = [vp, i32, i32, i32, i32, i32]
        L.orc_gen_set_params.restype = i32
        L.orc_gen_create_agents.argtypes = [vp, i32, vp, vp]
        L.orc_gen_update_agents.argtypes = [vp, vp, vp, vp]
        L.orc_gen_generate_observations.argtypes = [vp, vp]
        for f in ("orc_gen_dist", "orc_gen_next", "orc_gen_hist"):
            getattr(L, f).argtypes = [vp]
            getattr(L, f).restype = vp
        L.orc_env_step.argtypes = [u8p, i32, i32, i32, vp, vp, vp]
        L.orc_env_step.restype = i32
        L.orc_env_step_rules.argtypes = [u8p, i32, i32, i32, vp, vp, vp, i32]
        L.orc_env_step_rules.restype = i32
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def bfs(grid, goal):
    g = np.ascontiguousarray(np.asarray(grid) != 0, dtype=np.uint8)
    H, W = g.shape
    out = np.empty((H, W), dtype=np.uint16)
    lib().orc_bfs(g.ctypes.data, H, W, int(goal[0]), int(goal[1]), out.ctypes.data)
    return out


class OracleGenerator:
    """One env instance; mirrors ObservationGenerator(grid, cfg).  `params` = (cost2go_value_limit, num_agents,
    num_previous_actions, obs_radius, agents_radius) of struct InputParameters (observation_generator.h:22-40); default:
    the values inference.py:15-29 passes."""

    def __init__(self, grid, grid_step=64, params=None):
        self.grid = np.ascontiguousarray(np.asarray(grid) != 0, dtype=np.uint8)
        self.H, self.W = self.grid.shape
        self._h = lib().orc_gen_create(self.grid.ctypes.data, self.H, self.W)
        lib().orc_gen_set_grid_step(self._h, int(grid_step))          # InputParameters.grid_step (inference.py:28)
        self.nhist = 5
        if params is not None:
            if lib().orc_gen_set_params(self._h, *[int(v) for v in params]) != 0:
                raise ValueError(f"InputParameters {tuple(params)} cannot be run by the reference either")
            self.nhist = int(params[2])
        self.n = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_gen_destroy(self._h)
            self._h = None

    def create_agents(self, positions, goals):
        p, g = _i32(positions).reshape(-1, 2), _i32(goals).reshape(-1, 2)
        self.n = p.shape[0]
        lib().orc_gen_create_agents(self._h, self.n, p.ctypes.data, g.ctypes.data)

    def update_agents(self, positions, goals, actions):
        p, g, a = _i32(positions).reshape(-1, 2), _i32(goals).reshape(-1, 2), _i32(actions)
        assert p.shape[0] == self.n and a.shape[0] == self.n
        lib().orc_gen_update_agents(self._h, p.ctypes.data, g.ctypes.data, a.ctypes.data)

    def generate_observations(self):
        out = np.empty((self.n, 256), dtype=np.uint8)
        lib().orc_gen_generate_observations(self._h, out.ctypes.data)
        return out

    def dist(self):
        ptr = lib().orc_gen_dist(self._h)
        buf = (ctypes.c_uint16 * (self.n * self.H * self.W)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint16).reshape(self.n, self.H, self.W).copy()

    def next_tokens(self):
        ptr = lib().orc_gen_next(self._h)
        return np.frombuffer((ctypes.c_uint8 * self.n).from_address(ptr), dtype=np.uint8).copy()

    def hist_tokens(self):
        ptr = lib().orc_gen_hist(self._h)
        return np.frombuffer((ctypes.c_uint8 * (self.n * self.nhist)).from_address(ptr), dtype=np.uint8).reshape(self.n, self.nhist).copy()


RULE_NO_FOLLOW, RULE_LOWEST_WINS = 1, 2          # mapf_oracle.c: the two switchable (RECALLED) collision rules


def env_step(grid, pos, goal, actions, rules=0):
    """Our env spec (parity unpinned; `rules` = bit mask of the switchable collision rules, 0 = the spec).
    Returns (new_pos int32[n,2], n_on_goal)."""
    g = np.ascontiguousarray(np.asarray(grid) != 0, dtype=np.uint8)
    H, W = g.shape
    p = _i32(pos).reshape(-1, 2).copy()
    gl, a = _i32(goal).reshape(-1, 2), _i32(actions)
    k = lib().orc_env_step_rules(g.ctypes.data, H, W, p.shape[0], p.ctypes.data, gl.ctypes.data, a.ctypes.data, int(rules))
    return p, int(k)


def agents_density(grid, pos, radius=5):
    """One sample of POGEMA's AgentsDensityWrapper (wired in at experiment_setup/create_env.py:38,49; the wrapper itself
    lives in the un-vendored `pogema` pip dependency, so this is a restatement of its published behaviour: PARITY
    UNPINNED).  Per agent: non-zero cells of its obs["agents"] window (itself included) / traversable cells of its
    obs["obstacles"] window ((2r+1)^2, the map padded with obstacles); the sample is the mean over agents.  The episode
    metric `avg_agents_density` is the mean of the samples taken at reset and after every step."""
    g = np.asarray(grid) != 0
    H, W = g.shape
    pad = np.ones((H + 2 * radius, W + 2 * radius), bool)
    pad[radius:radius + H, radius:radius + W] = g
    occ = np.zeros_like(pad, dtype=bool)
    p = np.asarray(pos).reshape(-1, 2).astype(np.int64)
    occ[p[:, 0] + radius, p[:, 1] + radius] = True
    vals = []
    for r, c in p:
        wo = pad[r:r + 2 * radius + 1, c:c + 2 * radius + 1]
        wa = occ[r:r + 2 * radius + 1, c:c + 2 * radius + 1]
        vals.append(np.count_nonzero(wa) / (wo.size - np.count_nonzero(wo)))
    return float(np.mean(vals))
