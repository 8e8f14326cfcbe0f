"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares, and
refuses to compute without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mapf_gpt_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from mapf_gpt_amd import build
    build.build()
    return _lib.lib()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "mapf_gpt_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgpt_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(L):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mapf_gpt_amd.h but not exported"
    assert set(syms) == set(_lib.SYMBOLS), set(syms) ^ set(_lib.SYMBOLS)


def test_version_and_error_string(L):
    assert L.mgpt_abi_version() == 1002
    rc = L.mgpt_gpt_create(None, 1, 1, 32, 256, 1)
    assert rc == _lib.ERR_ARG and b"NULL" in L.mgpt_last_error()


def test_argument_validation_without_gpu(L):
    h = ctypes.c_void_p()
    assert L.mgpt_gpt_create(ctypes.byref(h), 2, 2, 64, 257, 4) == _lib.ERR_UNSUPPORTED      # block_size 1 .. 256
    assert L.mgpt_gpt_create(ctypes.byref(h), 2, 2, 64, 0, 4) == _lib.ERR_UNSUPPORTED
    assert L.mgpt_gpt_create(ctypes.byref(h), 2, 3, 64, 256, 4) == _lib.ERR_ARG              # n_embd % n_head
    # InputParameters (observation_generator.h:22-40): non-default values are honoured inside the kernels' layout bounds
    # (tests/test_gpu_tokenizer.py::test_non_default_input_parameters_vs_reference_goldens); outside them the call refuses
    for bad in [(20, 17, 5, 256, 5, 5),      # more than 16 record slots
                (20, 13, 6, 256, 5, 5),      # more than five previous actions (AgentRec::hist)
                (20, 13, 5, 256, 6, 5),      # obs radius 6: 169 window cells
                (20, 13, 5, 256, 5, 6),      # agents radius 6
                (3, 13, 5, 256, 5, 5),       # agents radius above the value limit: the reference throws (cpp:358-359)
                (20, 16, 5, 256, 5, 5),      # 121 + 16 * 10 tokens do not fit a 256-token row
                (20, 13, 5, 128, 5, 5),      # rows are 256 tokens whatever context_size says (cpp:386)
                (0, 13, 5, 256, 5, 5)]:
        st = _lib.InputParametersStruct(*bad, 64, 0)
        assert L.mgpt_tokenizer_create(ctypes.byref(h), ctypes.byref(st), 1, 4, 20, 20, 1) == _lib.ERR_UNSUPPORTED, bad


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BatchedTokenizer(np.zeros((20, 20), np.uint8), 1, 2)
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    with pytest.raises(RuntimeError):
        MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights="synthetic:tiny"))
