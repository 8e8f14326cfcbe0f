"""ctypes binding of libmapf_gpt_amd.so (the C ABI declared in include/mapf_gpt_amd.h).

There is NO fallback: if the library is missing or no HIP device is visible, compute calls raise.
`import torch` happens first on purpose -- PyTorch-ROCm ships its own libamdhip64.so.7 and the
library must bind to that same runtime instance (same SONAME), so that device pointers and streams
can be shared with torch tensors.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmapf_gpt_amd.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = 0, 1, 2, 3, 4
PREC_F32, PREC_F16X3, PREC_BF16 = 0, 1, 2
PRECISIONS = {"f32": PREC_F32, "fp32": PREC_F32, "f16x3": PREC_F16X3, "bf16": PREC_BF16}


class MGPTError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmapf_gpt_amd error {code}: {msg}")
        self.code = code


class InputParametersStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "cost2go_value_limit", "num_agents", "num_previous_actions", "context_size",
        "obs_radius", "agents_radius", "grid_step", "save_cost2go")]


_lib = None

# every symbol include/mapf_gpt_amd.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
_pp = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "mgpt_last_error": (ctypes.c_char_p, []),
    "mgpt_abi_version": (_i, []),
    "mgpt_device_count": (_i, [ctypes.POINTER(_i)]),
    "mgpt_tokenizer_create": (_i, [_pp, ctypes.POINTER(InputParametersStruct), _i, _i, _i, _i, _i]),
    "mgpt_tokenizer_destroy": (_i, [_vp]),
    "mgpt_tokenizer_vocab_size": (_i, [_vp, _vp]),
    "mgpt_tokenizer_set_grids": (_i, [_vp, _vp, _vp]),
    "mgpt_tokenizer_create_agents": (_i, [_vp, _vp, _vp, _vp]),
    "mgpt_tokenizer_update_agents": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "mgpt_tokenizer_update_agents_masked": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mgpt_tokenizer_generate_observations": (_i, [_vp, _vp, _vp]),
    "mgpt_tokenizer_state": (_i, [_vp, _pp, _pp]),
    "mgpt_tokenizer_copy_state": (_i, [_vp, _vp, _vp, _vp]),
    "mgpt_env_create": (_i, [_pp, _i, _i, _i, _i, _i, _i]),
    "mgpt_env_destroy": (_i, [_vp]),
    "mgpt_env_set_grids": (_i, [_vp, _vp, _vp]),
    "mgpt_env_reset": (_i, [_vp, _vp, _vp, _vp]),
    "mgpt_env_step": (_i, [_vp, _vp, _vp]),
    "mgpt_env_state": (_i, [_vp, _pp, _pp, _pp]),
    "mgpt_env_copy_state": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mgpt_env_step_host": (_i, [_vp, _vp, _vp, _vp]),
    "mgpt_env_metrics": (_i, [_vp, _vp, _vp]),
    "mgpt_env_set_rules": (_i, [_vp, _i]),
    "mgpt_env_set_lifelong": (_i, [_vp, _vp, _i, _vp]),
    "mgpt_env_lifelong_counts": (_i, [_vp, _vp, _vp]),
    "mgpt_dataset_create": (_i, [_pp, _vp, _i, _i, _vp]),
    "mgpt_dataset_destroy": (_i, [_vp]),
    "mgpt_dataset_tokenize": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "mgpt_dataset_tokenize_ex": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "mgpt_gpt_create": (_i, [_pp, _i, _i, _i, _i, _i]),
    "mgpt_gpt_destroy": (_i, [_vp]),
    "mgpt_gpt_set_param": (_i, [_vp, ctypes.c_char_p, _vp, _i64, _i]),
    "mgpt_gpt_finalize": (_i, [_vp]),
    "mgpt_gpt_forward": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "mgpt_gpt_forward_t": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mgpt_gpt_act": (_i, [_vp, _vp, _i, _vp, _vp, _i, _u64, _u64, _u64, _i, _vp]),
    "mgpt_gpt_act_dev": (_i, [_vp, _vp, _i, _vp, _vp, _i, _u64, _vp, _u64, _i, _vp]),
    "mgpt_step_create": (_i, [_pp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64]),
    "mgpt_step_destroy": (_i, [_vp]),
    "mgpt_step_reset": (_i, [_vp, _u64, _vp]),
    "mgpt_step_run": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "mgpt_gpt_debug_copy": (_i, [_vp, _i, _vp, _i64, _vp]),
    "mgpt_gpt_debug_copy_raw": (_i, [_vp, _i, _i, _vp, _i64, _vp]),
    "mgpt_gpt_debug_counter": (_i, [_i, ctypes.POINTER(_u64), _i]),
    "mgpt_gpt_set_envelope_policy": (_i, [_vp, _i]),
    "mgpt_gpt_envelope": (_i, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(_i)]),
    "mgpt_gpt_envelope_probe": (_i, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "mgpt_sample_actions": (_i, [_vp, _i, _vp, _i, _u64, _u64, _u64, _vp]),
    "mgpt_prof_enable": (_i, [_i]),
    "mgpt_prof_reset": (_i, []),
    "mgpt_prof_read": (_i, [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                            ctypes.POINTER(_i64), ctypes.POINTER(_i)]),
}


def lib():
    """The loaded library (raises if it was never built: run `python -m mapf_gpt_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing. Build it with `python -m mapf_gpt_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise MGPTError(rc, lib().mgpt_last_error().decode("utf-8", "replace"))


def device_count():
    n = ctypes.c_int(0)
    check(lib().mgpt_device_count(ctypes.byref(n)))
    return n.value


def require_gpu():
    if not torch.cuda.is_available() or device_count() == 0:
        raise RuntimeError("mapf_gpt_amd needs a HIP device (MI355X / gfx950); none is visible and there is "
                           "no CPU fallback on the product path.")


def stream_ptr(stream=None):
    """hipStream_t of `stream` (default: torch's current stream) as an integer for the C ABI."""
    s = torch.cuda.current_stream() if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """Context that makes `device` current for the library call inside it; a no-op object when it already is (the
    torch.cuda.device context manager costs ~10 us per call, which shows on sub-millisecond steps)."""
    if device.index is None or torch.cuda.current_device() == device.index:      # "cuda" = whatever is current
        return _NO_SWITCH
    return torch.cuda.device(device)


def ptr(t):
    """Device (or host) data pointer of a contiguous tensor."""
    assert t.is_contiguous(), "tensor handed to the C ABI must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def prof_enable(on=True):
    check(lib().mgpt_prof_enable(1 if on else 0))


def prof_reset():
    check(lib().mgpt_prof_reset())


def prof_read():
    """-> {kernel_class: (total_ms, launches)}; synchronises the device."""
    cap = 32
    names = (ctypes.c_char_p * cap)()
    ms = (ctypes.c_float * cap)()
    cnt = (ctypes.c_int64 * cap)()
    n = ctypes.c_int(cap)
    check(lib().mgpt_prof_read(names, ms, cnt, ctypes.byref(n)))
    return {names[i].decode(): (float(ms[i]), int(cnt[i])) for i in range(n.value)}


def debug_counter(which=0, reset=False):
    """Event counter `which` of the policy kernels (include/mapf_gpt_amd.h: mgpt_gpt_debug_counter)."""
    v = _u64(0)
    check(lib().mgpt_gpt_debug_counter(int(which), ctypes.byref(v), 1 if reset else 0))
    return int(v.value)
