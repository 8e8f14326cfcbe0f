"""Host mirror of the reference policy class (mapf_gpt/model.py) over the HIP forward.

`GPTConfig` / `GPT` keep the reference's names and call shapes for the inference surface:
    GPT(config); load_state_dict(sd, strict=False); to(device); eval();
    forward(idx) -> (logits [B,1,67], None)        (model.py:167-189, last position only)
    act(idx, do_sample=True, generator=None) -> LongTensor [B]   (model.py:244-260)
plus the device-resident fast path `act_tokens(tokens_u8, ...)` used by the batched runner.
Training helpers of the reference (configure_optimizers, estimate_mfu, crop_block_size) are out of
scope.  Every compute call goes through the C ABI; there is no
PyTorch forward here.
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, weights


@dataclass
class GPTConfig:          # = model.py:107-115 (same field names and defaults)
    block_size: int = 161
    vocab_size: int = 67
    n_layer: int = 8
    n_head: int = 8
    n_embd: int = 256
    dropout: float = 0.0
    bias: bool = False


class GPT:
    ENVELOPE_POLICIES = {"fallback": 0, "refuse": 1, "ignore": 2}      # include/mapf_gpt_amd.h: MGPT_ENVELOPE_*

    def __init__(self, config, max_rows=2048, precision="f32", device="cuda", envelope="fallback"):
        if config.vocab_size != 67:
            raise ValueError("vocab_size must be 67 (observation_generator.cpp:321-344)")
        # bias=True (model.py:115; no released config uses it): the checkpoint's *.bias tensors are loaded like any other and carried by the
        # exact-fp32 kernels -- precision="f16x3" then follows the envelope policy (fallback: fp32), "bf16" raises (include/mapf_gpt_amd.h)
        if not (0.0 <= config.dropout < 1.0):
            raise ValueError("dropout must be in [0, 1)")
        # dropout: accepted and ignored -- this is the inference path, and the reference serves every model in eval mode
        # (inference.py:85 net.eval(): nn.Dropout and scaled_dot_product_attention's dropout_p are the identity there, model.py:33-34,59,82,129)
        self.config = config
        self.max_rows = int(max_rows)
        self.precision = precision
        if envelope not in self.ENVELOPE_POLICIES:
            raise ValueError(f"envelope must be one of {sorted(self.ENVELOPE_POLICIES)}")
        # what happens to precision="f16x3" requests when the loaded checkpoint lies outside the range on which that mode's 1e-5
        # logit bar was established: "fallback" = serve them with the exact-fp32 kernels (one line on stderr), "refuse" = raise,
        # "ignore" = run the split path regardless (parity experiments).  See include/mapf_gpt_amd.h, mgpt_gpt_envelope.
        self.envelope_policy = envelope
        self.device = torch.device(device)
        self._h = None
        self._loaded = False
        self.training = False
        self._act_calls = 0          # advances the library RNG between generator-less act() calls

    # ---- lifecycle -------------------------------------------------------------------------
    def _ensure(self):
        if self._h is None:
            _lib.require_gpu()
            h = ctypes.c_void_p()
            c = self.config
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().mgpt_gpt_create(ctypes.byref(h), c.n_layer, c.n_head, c.n_embd, c.block_size, self.max_rows))
            self._h = h
            _lib.check(_lib.lib().mgpt_gpt_set_envelope_policy(h, self.ENVELOPE_POLICIES[self.envelope_policy]))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mgpt_gpt_destroy(h)
            except Exception:      # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("mapf_gpt_amd.GPT runs on a HIP device only (no CPU forward in this package)")
        if self._h is not None and device != self.device and device.index is not None:
            raise RuntimeError("move the model before loading weights")
        self.device = device
        return self

    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, state_dict, strict=False):
        """Accepts the reference checkpoint's state_dict (tensors or arrays, `_orig_mod.` prefixes allowed,
        inference.py:33-44).  strict=False (the reference's choice, inference.py:83) skips unknown keys."""
        self._ensure()
        sd = weights.strip_prefix(state_dict)
        unknown = []
        for k, v in sd.items():
            if k.endswith(".bias") and not self.config.bias:      # a bias=False module has no such parameter: an unexpected key (model.py:14-17,29)
                unknown.append(k)
                continue
            a = v.detach().to(torch.float32).cpu().numpy() if hasattr(v, "detach") else np.asarray(v, dtype=np.float32)
            a = np.ascontiguousarray(a)
            rc = _lib.lib().mgpt_gpt_set_param(self._h, k.encode(), ctypes.c_void_p(a.ctypes.data), a.size, 0)
            if rc == _lib.ERR_ARG and b"unknown parameter" in _lib.lib().mgpt_last_error():
                unknown.append(k)
                continue
            _lib.check(rc)
        if strict and unknown:
            raise KeyError(f"unexpected keys: {unknown}")
        _lib.check(_lib.lib().mgpt_gpt_finalize(self._h))
        self._loaded = True
        return unknown

    def envelope(self):
        """Precision envelope of the loaded checkpoint: {max_abs_w, max_rms_w (over the block matrices), probe_err (max |f16x3 - f32|
        over the probe rows' logits, None before the first f16x3 forward), state: "undecided" | "inside" | "outside",
        effective_precision: what a precision="f16x3" request runs}."""
        assert self._loaded, "load_state_dict first"
        out, st = (ctypes.c_float * 3)(), ctypes.c_int(0)
        _lib.check(_lib.lib().mgpt_gpt_envelope(self._h, out, ctypes.byref(st)))
        state = ("undecided", "inside", "outside")[st.value]
        eff = "f32" if (state == "outside" and self.envelope_policy == "fallback") else "f16x3"
        pr = (ctypes.c_float * 4)()
        _lib.check(_lib.lib().mgpt_gpt_envelope_probe(self._h, pr))
        return {"max_abs_w": float(out[0]), "max_rms_w": float(out[1]), "probe_err": (None if out[2] < 0 else float(out[2])),
                "probe_err_small_calls": (None if pr[0] < 0 else float(pr[0])), "probe_err_large_calls": (None if pr[1] < 0 else float(pr[1])),
                "probe_tol": (None if out[2] < 0 else float(pr[2])), "probe_max_logit": (None if out[2] < 0 else float(pr[3])),
                "state": state, "policy": self.envelope_policy, "effective_precision": eff}

    # ---- compute ---------------------------------------------------------------------------
    def _prec(self, precision):
        return _lib.PRECISIONS[precision or self.precision]

    def _tokens_u8(self, idx):
        # model.py:168-170: any t <= block_size.  256-token rows (what the tokenizer emits, inference.py:145) take the model's kernels of its
        # precision; shorter rows are served by the exact-fp32 kernels (mgpt_gpt_forward_t)
        if idx.dim() != 2 or idx.shape[1] < 1:
            raise ValueError(f"idx must be [B, T] token rows, got {tuple(idx.shape)}")
        if idx.shape[1] > self.config.block_size:
            raise ValueError(f"Cannot forward sequence of length {idx.shape[1]}, block size is only {self.config.block_size}")
        return idx.to(device=self.device, dtype=torch.uint8).contiguous()

    def logits_tokens_t(self, tokens_u8, out=None):
        """tokens uint8 [rows, T <= block_size] on the device -> float32 [rows, 67]: the logits of position T - 1 (model.py:186), exact-fp32 kernels."""
        assert self._loaded, "load_state_dict first"
        rows, T = tokens_u8.shape
        if out is None:
            out = torch.empty((rows, 67), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_gpt_forward_t(self._h, _lib.ptr(tokens_u8), rows, T, _lib.ptr(out), _lib.stream_ptr()))
        return out

    def _logits(self, tokens_u8):
        return self.logits_tokens(tokens_u8) if tokens_u8.shape[1] == 256 else self.logits_tokens_t(tokens_u8)

    def logits_tokens(self, tokens_u8, precision=None, out=None):
        """tokens uint8 [rows, 256] on the device -> float32 [rows, 67] (last-position logits)."""
        assert self._loaded, "load_state_dict first"
        rows = tokens_u8.shape[0]
        if out is None:
            out = torch.empty((rows, 67), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_gpt_forward(self._h, _lib.ptr(tokens_u8), rows, _lib.ptr(out), self._prec(precision),
                                                   _lib.stream_ptr()))
        return out

    def act_tokens(self, tokens_u8, do_sample=True, seed=0, step=0, precision=None, out=None, logits_out=None, row0=0):
        """Fused forward + 5-way masked softmax + sampling on the device (library RNG keyed by
        (seed, step, row0 + row); row0 = global id of the first row when the caller holds a shard) -> int32 [rows]."""
        assert self._loaded, "load_state_dict first"
        rows = tokens_u8.shape[0]
        if out is None:
            out = torch.empty((rows,), dtype=torch.int32, device=self.device)
        lp = _lib.ptr(logits_out) if logits_out is not None else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mgpt_gpt_act(self._h, _lib.ptr(tokens_u8), rows, _lib.ptr(out), lp, 1 if do_sample else 0,
                                               int(seed) & (2 ** 64 - 1), int(step), int(row0), self._prec(precision), _lib.stream_ptr()))
        return out

    def forward(self, idx, targets=None):
        if targets is not None:
            raise NotImplementedError("training loss is out of scope (inference path only)")
        logits = self._logits(self._tokens_u8(idx))
        return logits[:, None, :], None

    __call__ = forward

    @torch.no_grad()
    def act(self, idx, do_sample=True, generator=None):
        """= GPT.act (model.py:244-260).  With a torch `generator` the draw is torch.multinomial on the
        device from OUR logits (same masking/softmax as the reference); without one the library's fused
        sampler runs, keyed by (seed drawn once from torch's global RNG, call counter, row) so that
        successive calls draw fresh uniforms like the reference's advancing global RNG does.
        Returns an int64 tensor [B] (0-d for B == 1, like the reference's .squeeze())."""
        tokens = self._tokens_u8(idx)
        if tokens.shape[1] != 256:                 # short rows (never the hot path): logits from the fp32 kernels, then the reference's own call shape
            logits = self.logits_tokens_t(tokens)
            if do_sample:
                return torch.multinomial(torch.softmax(logits[:, :5], dim=-1), num_samples=1, generator=generator).squeeze()   # model.py:250-257
            return torch.argmax(logits[:, :5], dim=-1).squeeze()                                                               # model.py:258-259
        if do_sample and generator is not None:
            logits = self.logits_tokens(tokens)
            probs = torch.softmax(logits[:, :5], dim=-1)                       # model.py:250-254
            nxt = torch.multinomial(probs, num_samples=1, generator=generator)  # model.py:257
            return nxt.squeeze()
        if do_sample:
            # The library sampler's seed is drawn from torch's global (CPU) RNG at the first SAMPLED call (a greedy first call
            # must not pin seed 0) and drawn again -- call counter back to 0 -- whenever that RNG was touched since our draw:
            # torch.manual_seed(s), also with the SAME s as before (the usual reproducibility pattern: the second run then
            # replays the first one's draws, as the reference's multinomial would), or any other consumer of the global stream.
            # (A seed pinned with reset_sampler(seed) is kept until reset_sampler() is called again: no look at the global RNG --
            #  another thread's act() or any torch.manual_seed silently dropped the pin, and the 5-KB state snapshot of
            #  every call sat on the one-environment latency path.)
            pinned = getattr(self, "_act_pinned", False)
            if not pinned and (getattr(self, "_act_seed", None) is None or not torch.equal(torch.get_rng_state(), self._act_rng_after)):
                self._act_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                self._act_rng_after = torch.get_rng_state()
                self._act_calls = 0
        step = self._act_calls
        self._act_calls += 1
        return self.act_tokens(tokens, do_sample=do_sample, seed=getattr(self, "_act_seed", None) or 0, step=step).to(torch.int64).squeeze()

    def reset_sampler(self, seed=None):
        """Explicit control of the generator-less sampler: forget the drawn seed (the next sampled act() draws a new one from
        torch's global RNG) or pin `seed`; the call counter restarts either way."""
        self._act_calls = 0
        if seed is None:
            self._act_seed = None
            self._act_pinned = False
        else:
            self._act_seed = int(seed)
            self._act_pinned = True


def build_model(name_or_args, seed=0, scale=1.0, max_rows=2048, precision="f32", device="cuda", state_dict=None, envelope="fallback"):
    """Convenience: GPT with the named shape ("2M", "6M", "85M", "tiny") and synthetic or given weights."""
    args = weights.model_args(name_or_args)
    net = GPT(GPTConfig(**args), max_rows=max_rows, precision=precision, device=device, envelope=envelope)
    net.load_state_dict(state_dict if state_dict is not None else weights.synthetic_state_dict(args, seed=seed, scale=scale))
    return net.eval()
