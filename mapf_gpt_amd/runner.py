"""Device-resident episode loop: env step -> tokenizer -> policy -> actions for many instances, with
instance sharding across the GPUs of a node and ONE collective (metrics all_gather) at the end.

This is what replaces the reference's episode-level process pool (dask workers each running
run_episode with its own model copy: eval_configs/01-random/01-random.yaml:147-148, inference.py:30-31,
121-125): instances are batched on one GPU and partitioned over ranks; no per-step traffic leaves
the GPU and nothing synchronises with the host inside the loop.

One hot-path step (order as in the reference's run_episode loop, example.py:65 / inference.py:142-168):
    tokenizer.update_agents(env.pos, env.goal, last_actions)   # history gets the previous INTENDED actions
    tokens  = tokenizer.generate_observations()                # uint8 [rows, 256]
    actions = policy.act(tokens)                               # int32 [rows]
    env.step(actions); last_actions = actions
"""
import ctypes

import numpy as np
import torch

from . import _lib, maps
from .env import BatchedEnv
from .observation_generator import BatchedTokenizer


def shard_range(n_total, rank, world):
    """Contiguous instance range [lo, hi) of `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metrics(local, n_total, rank=0, world=1, group=None, force=False):
    """all_gather of per-instance metric records float32 [n_local, 5] -> [n_total, 5] on every rank.
    The only collective of the job (RCCL over xGMI with backend "nccl"; gloo on CPU for tests);
    ~20 B x instances, latency-bound -- one call, no ring tuning.  force: take the collective at world = 1 too (a process
    group of one rank: how the RCCL branch is exercised on a one-GPU box, tests/test_gpu_bench.py)."""
    if world == 1 and not force:
        return local
    import torch.distributed as dist
    dev = local.device
    if local.is_cuda and dist.get_backend(group) == "gloo":      # CPU collectives (tests, dry runs): stage through the host
        local = local.cpu()
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    pad = max(counts)
    buf = torch.zeros((pad, local.shape[1]), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0).to(dev)


def make_instances(grid, n_inst, n_agents, first_seed=0, start_ok=None, goal_ok=None):
    """Seeded starts/goals for instances first_seed .. first_seed+n_inst-1 on one shared padded map
    (instance i uses seed i: SURVEY.md section 8d) -> int16 tensors [n_inst, n_agents, 2]."""
    comp = maps.largest_component(grid == 0)
    pos = np.empty((n_inst, n_agents, 2), dtype=np.int16)
    goal = np.empty((n_inst, n_agents, 2), dtype=np.int16)
    for i in range(n_inst):
        pos[i], goal[i] = maps.place_agents(grid, n_agents, first_seed + i, start_ok, goal_ok, component=comp)
    return torch.from_numpy(pos), torch.from_numpy(goal)


class BatchedRunner:
    def __init__(self, grids, n_inst, n_agents, net, max_episode_steps=128, seed=0, do_sample=True, precision=None,
                 device="cuda", row_offset=0, use_graph=False):
        self.device = torch.device(device)
        self.env = BatchedEnv(grids, n_inst, n_agents, max_episode_steps, device=device)
        self.tok = BatchedTokenizer(grids, n_inst, n_agents, device=device)
        self.net = net
        self.n_inst, self.n_agents = n_inst, n_agents
        self.rows = n_inst * n_agents
        self.seed, self.do_sample, self.precision = seed, do_sample, precision
        self.row_offset = row_offset          # global row id of this shard's first row: the sampler keys draws by global row
        self.tokens = torch.empty((self.rows, 256), dtype=torch.uint8, device=self.device)
        self.actions = torch.full((n_inst, n_agents), -1, dtype=torch.int32, device=self.device)
        self.t = 0
        self._pos_ptr, self._goal_ptr, _ = self.env.state_ptrs()
        # the whole step behind one library call.  use_graph=True replays it as a hipGraph after the first (eager) step: an OPTION since
        # round 5, not the default -- a cfg1 step is 19 dependent launches that keep the GPU 97 % busy, so the replay has no launch cost
        # to remove (0.279 ms against 0.272 ms eager, BENCH_r04); it stays for callers that want the host out of the loop, and
        # tests/test_gpu_step_graph.py keeps it bit-identical to the eager path
        self.use_graph = bool(use_graph)
        self._step = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_step_create(ctypes.byref(self._step), self.tok._h, self.net._h, self.env._h, self.rows,
                                                   _lib.PRECISIONS[precision or net.precision], 1 if do_sample else 0,
                                                   int(seed) & (2 ** 64 - 1), int(row_offset)))

    def __del__(self):
        h = getattr(self, "_step", None)
        if h:
            try:
                _lib.lib().mgpt_step_destroy(h)
            except Exception:      # interpreter shutdown
                pass
            self._step = None

    def reset(self, pos, goal, goal_queue=None):
        """goal_queue int16 [n_inst, n_agents, Q, 2]: lifelong mode (on_target="restart"), the tokenizer then re-checks
        goals every step (observation_generator.cpp:464-477)."""
        if goal_queue is not None or self.env.lifelong:
            self.env.set_lifelong(goal_queue)
        self.env.reset(pos, goal)
        self.tok.create_agents(self.env.pos, self.env.goal)
        self.actions.fill_(-1)                                   # inference.py:140
        self.t = 0
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_step_reset(self._step, 0, _lib.stream_ptr()))

    def step(self):
        """update_agents -> generate_observations -> act -> env.step in ONE library call (mgpt_step_run)."""
        with _lib.on_device(self.device):
            _lib.check(_lib.lib().mgpt_step_run(self._step, _lib.ptr(self.tokens), _lib.ptr(self.actions.view(-1)),
                                                1 if self.env.lifelong else 0, 1 if self.use_graph else 0, _lib.stream_ptr()))
        self.t += 1

    def run(self, steps):
        for _ in range(steps):
            self.step()

    def metrics(self):
        return self.env.metrics()
