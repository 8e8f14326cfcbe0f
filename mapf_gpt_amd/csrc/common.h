// common.h -- shared host-side plumbing of libmapf_gpt_amd.so (error strings, HIP checks,
// per-kernel-class event timing).  gfx950 only; no CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mapf_gpt_amd.h"

namespace mgpt {

void set_error(const char *fmt, ...);

// Allocation generation, PER CONTEXT: a policy / env context bumps its own counter whenever it frees or re-allocates device
// memory that a captured step graph may have baked in (weight planes and workspaces at finalize / lazy mode build, lifelong
// goal queues).  mgpt_step compares the generations of the contexts IT holds with the values it saw at its last step and, on
// a change, drops its graph and runs one eager step first; other models, envs and runners of the process are not disturbed.
// (The tokenizer context never re-allocates after create.)
uint64_t gpt_generation(const mgpt_gpt *g);
uint64_t env_generation(const mgpt_env *e);

#define MGPT_HIP(call)                                                                     \
    do {                                                                                   \
        hipError_t e__ = (call);                                                           \
        if (e__ != hipSuccess) {                                                           \
            mgpt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return MGPT_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

#define MGPT_REQUIRE(cond, code, ...)                                                      \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            mgpt::set_error(__VA_ARGS__);                                                  \
            return (code);                                                                 \
        }                                                                                  \
    } while (0)

#define MGPT_LAUNCH_CHECK()                                                                \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            mgpt::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return MGPT_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

// kernel classes for the timing hooks (names in prof.cpp)
enum ProfId {
    P_BFS = 0, P_TOK_UPDATE, P_TOK_NEXT, P_TOKENS, P_ENV_STEP, P_ENV_METRICS,
    P_EMBED, P_LAYERNORM, P_GEMM_QKV, P_ATTN, P_GEMM_PROJ, P_GEMM_FC, P_GEMM_PROJ2, P_MLP_FUSED,
    P_HEAD, P_SAMPLE, P_PACK, P_LNQKV_FUSED,
    P_ATTN_LAST, P_GEMM_PROJ_LAST, P_MLP_FUSED_LAST,    // the last layer's launches (token 255 only, model.py:186): timed apart from the full ones
    P_COUNT
};

// RAII: records start/stop events around a launch when profiling is enabled (no-op otherwise)
struct ProfScope {
    int slot;
    hipStream_t s;
    ProfScope(ProfId id, hipStream_t stream);
    ~ProfScope();
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mgpt
