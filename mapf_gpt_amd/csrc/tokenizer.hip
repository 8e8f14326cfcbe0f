// tokenizer.hip -- MI355X observation tokenizer (replaces mapf_gpt/observation_generator.{h,cpp}).
//
// State is a structure-of-arrays over (instance, agent) resident in HBM:
//   grids   uint8  [n_grids, H, W]            obstacle maps (instance i uses map i % n_grids)
//   dist    uint16 [n_inst, n_agents, H, W]   4-connected BFS distance-to-goal fields (65535 = wall/unreached)
//   recs    16 B   [n_inst, n_agents]         {pos, goal, 5 history tokens, greedy-bits token}
// Kernels (all integer, bit-exact against the reference):
//   bfs_kernel        one 256-thread workgroup per agent, monotone relaxation in LDS    (cpp:200-286)
//   create/update     one thread per agent                                              (cpp:391-410, 432-485)
//   tokens_kernel     one wavefront per agent, 4 agents in flight per workgroup, the instance's
//                     agent records staged in LDS, window gathered straight from the agent's own
//                     distance field, neighbours ranked wave-parallel                    (cpp:288-311, 487-528, 352-389)
#include "common.h"

using namespace mgpt;

namespace {

struct __attribute__((aligned(16))) AgentRec {
    int16_t pr, pc, gr, gc;
    uint8_t hist[5];
    uint8_t next;
    uint8_t pad[2];
};
static_assert(sizeof(AgentRec) == 16, "AgentRec must be 16 bytes");

constexpr int kUnreach = 65535;
constexpr int kFreeUnset = 65534;   // transient marker inside bfs_kernel only
constexpr int kR = 5;               // obs_radius == agents_radius == 5 (inference.py:18-19)
constexpr int kWin = 2 * kR + 1;    // 11
constexpr int kLimit = 20;          // cost2go_value_limit (inference.py:17)
constexpr int kSlots = 13;          // num_agents (inference.py:15)
constexpr int TOK_UNREACH = 41, TOK_NEG = 42, TOK_POS = 43, TOK_N = 44, TOK_BITS0 = 50, TOK_PAD = 66;

// ---------------------------------------------------------------------------------------------
// Distance field.  The reference's tiled / border-table / priority-queue construction (cpp:43-286)
// equals the plain 4-connected BFS distance from the goal (SURVEY.md finding 4, re-verified by
// tests/test_oracle_vs_reference.py).  Here: chaotic min-plus relaxation, d[i] <- min(d[i], min_nb+1),
// until a full sweep changes nothing.  Every value ever stored is the length of a real path, the
// update is monotone, so the fixpoint is the exact shortest distance regardless of thread order.
// `d` lives in LDS when the map fits (<= 32768 cells), else directly in the output buffer.
// ---------------------------------------------------------------------------------------------
template <bool kLds>
__global__ __launch_bounds__(256) void bfs_kernel(const uint8_t *__restrict__ grids, int n_grids, int n_agents,
                                                  int H, int W, const AgentRec *__restrict__ recs,
                                                  const uint8_t *__restrict__ dirty, uint16_t *__restrict__ dist_all)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ag = blockIdx.x;
    if (dirty != nullptr && dirty[ag] == 0) return;
    const int inst = ag / n_agents;
    const int cells = H * W;
    const uint8_t *grid = grids + (size_t)(inst % n_grids) * cells;
    uint16_t *out = dist_all + (size_t)ag * cells;
    uint16_t *d = kLds ? reinterpret_cast<uint16_t *>(smem) : out;
    const AgentRec r = recs[ag];
    const int tid = threadIdx.x;
    const bool goal_ok = r.gr >= 0 && r.gr < H && r.gc >= 0 && r.gc < W;

    for (int i = tid; i < cells; i += 256) d[i] = grid[i] ? kUnreach : kFreeUnset;
    __syncthreads();
    if (tid == 0 && goal_ok) d[r.gr * W + r.gc] = 0;   // cpp:157-159: the goal is seeded even if blocked
    __syncthreads();

    if (goal_ok) {
        for (;;) {
            int changed = 0;
            for (int i = tid; i < cells; i += 256) {
                const int cur = d[i];
                if (cur == kUnreach || cur == 0) continue;
                const int rr = i / W, cc = i - rr * W;
                int m = kUnreach;
                if (rr > 0) m = min(m, (int)d[i - W]);
                if (rr < H - 1) m = min(m, (int)d[i + W]);
                if (cc > 0) m = min(m, (int)d[i - 1]);
                if (cc < W - 1) m = min(m, (int)d[i + 1]);
                if (m < kFreeUnset && m + 1 < cur) {
                    d[i] = (uint16_t)(m + 1);
                    changed = 1;
                }
            }
            if (!__syncthreads_or(changed)) break;
        }
    }
    for (int i = tid; i < cells; i += 256) {
        int v = d[i];
        if (v == kFreeUnset) v = kUnreach;
        out[i] = (uint16_t)v;
    }
}

// greedy-direction bits, cpp:412-430: order u(-1,0) d(+1,0) l(0,-1) r(0,+1); bit = neighbour strictly closer
__device__ __forceinline__ int next_action_token(const uint16_t *__restrict__ d, int H, int W, int pr, int pc)
{
    if (pr < 0 || pr >= H || pc < 0 || pc >= W) return TOK_BITS0;
    const int cur = d[pr * W + pc];
    const int u = (pr > 0) ? (int)d[(pr - 1) * W + pc] : kUnreach;
    const int dn = (pr < H - 1) ? (int)d[(pr + 1) * W + pc] : kUnreach;
    const int l = (pc > 0) ? (int)d[pr * W + pc - 1] : kUnreach;
    const int rt = (pc < W - 1) ? (int)d[pr * W + pc + 1] : kUnreach;
    return TOK_BITS0 + 8 * (u < cur) + 4 * (dn < cur) + 2 * (l < cur) + (rt < cur);
}

// create_agents, cpp:391-410 (history <- "n" x 5)
__global__ __launch_bounds__(256) void tok_create_kernel(AgentRec *__restrict__ recs, const int16_t *__restrict__ pos,
                                                         const int16_t *__restrict__ goal, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    AgentRec r;
    r.pr = pos[2 * i]; r.pc = pos[2 * i + 1];
    r.gr = goal[2 * i]; r.gc = goal[2 * i + 1];
#pragma unroll
    for (int k = 0; k < 5; k++) r.hist[k] = TOK_N;
    r.next = TOK_BITS0;
    r.pad[0] = r.pad[1] = 0;
    recs[i] = r;
}

// update_agents, cpp:432-485: position, action history (intended action of the previous step),
// goal change detection; greedy bits here when goals are static, else in tok_next_kernel after the BFS.
__global__ __launch_bounds__(256) void tok_update_kernel(AgentRec *__restrict__ recs, const int16_t *__restrict__ pos,
                                                         const int16_t *__restrict__ goal,
                                                         const int32_t *__restrict__ actions, uint8_t *__restrict__ dirty,
                                                         int total, int check_goals, const uint16_t *__restrict__ dist,
                                                         int H, int W)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    AgentRec r = recs[i];
    r.pr = pos[2 * i]; r.pc = pos[2 * i + 1];
    const int act = actions[i];
    r.hist[0] = r.hist[1]; r.hist[1] = r.hist[2]; r.hist[2] = r.hist[3]; r.hist[3] = r.hist[4];   // cpp:463
    r.hist[4] = (uint8_t)((act >= 0 && act <= 4) ? TOK_N + 1 + act : TOK_N);                        // cpp:442-462
    if (check_goals) {
        const int16_t gr = goal[2 * i], gc = goal[2 * i + 1];
        dirty[i] = (gr != r.gr || gc != r.gc) ? 1 : 0;                                              // cpp:464-468
        r.gr = gr; r.gc = gc;
    } else {
        r.next = (uint8_t)next_action_token(dist + (size_t)i * H * W, H, W, r.pr, r.pc);            // cpp:483-484
    }
    recs[i] = r;
}

__global__ __launch_bounds__(256) void tok_next_kernel(AgentRec *__restrict__ recs, int total,
                                                       const uint16_t *__restrict__ dist, int H, int W)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pr = recs[i].pr, pc = recs[i].pc;
    recs[i].next = (uint8_t)next_action_token(dist + (size_t)i * H * W, H, W, pr, pc);
}

// ---------------------------------------------------------------------------------------------
// generate_observations, cpp:516-528.  Workgroup = 4 wavefronts = one chunk of kAgentsPerBlock
// agents of ONE instance; a wavefront owns one agent (one 256-token row) at a time.
//   LDS: the instance's agent records (16 B each), per wave a 256-B row image and a candidate list.
//   HBM reads per row: 121 x u16 window of the agent's own distance field (11 row segments of 22 B);
//   HBM writes per row: one coalesced 256-B store (64 lanes x 4 B).
// Neighbour order = (Manhattan distance, agent id) ascending, first 13 (cpp:496-506); the key
// (md << 16 | id) is unique, so rank = number of smaller keys is a permutation -- computed by all
// candidates in parallel, no serial sort.
// ---------------------------------------------------------------------------------------------
constexpr int kAgentsPerBlock = 16;
constexpr int kMaxCand = 128;

__device__ __forceinline__ int window_token(int v, int mid)
{
    if (v == kUnreach) return TOK_UNREACH;          // cpp:308-309 (-80)
    const int w = v - mid;                          // cpp:304
    return w > kLimit ? TOK_POS : (w < -kLimit ? TOK_NEG : w + kLimit);   // cpp:305-306
}

__global__ __launch_bounds__(256) void tokens_kernel(const AgentRec *__restrict__ recs, const uint16_t *__restrict__ dist,
                                                     int n_agents, int H, int W, int chunks_per_inst,
                                                     uint8_t *__restrict__ tokens)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    AgentRec *srec = reinterpret_cast<AgentRec *>(smem);                                  // [n_agents]
    uint32_t *scand = reinterpret_cast<uint32_t *>(smem + (size_t)n_agents * 16);         // [4][kMaxCand]
    uint8_t *srow = reinterpret_cast<uint8_t *>(scand + 4 * kMaxCand);                    // [4][256]

    const int inst = blockIdx.x / chunks_per_inst;
    const int chunk = blockIdx.x - inst * chunks_per_inst;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const AgentRec *grec = recs + (size_t)inst * n_agents;

    for (int i = tid; i < n_agents; i += 256) srec[i] = grec[i];      // one 16-B load/store per record
    __syncthreads();

    uint32_t *cand = scand + wave * kMaxCand;
    uint8_t *row = srow + wave * 256;
    uint32_t *row32 = reinterpret_cast<uint32_t *>(row);
    const int cells = H * W;

    const int a_begin = chunk * kAgentsPerBlock;
    // Issue the window gathers of ALL agents this wave owns before touching any of them: the kernel is bound by
    // HBM latency (two 2-byte gathers per agent), so bytes in flight per wave, not instructions, set the rate.
    constexpr int kPerWave = kAgentsPerBlock / 4;
    const int p0 = lane, i0 = p0 / kWin, j0 = p0 - i0 * kWin;
    const int p1 = lane + 64, i1 = p1 / kWin, j1 = p1 - i1 * kWin;
    int w0[kPerWave], w1[kPerWave];
#pragma unroll
    for (int q = 0; q < kPerWave; q++) {
        const int a = a_begin + wave + 4 * q;
        w0[q] = kUnreach; w1[q] = kUnreach;
        if (a < n_agents) {                                             // wave-uniform
            const int pr = srec[a].pr, pc = srec[a].pc;
            const uint16_t *d = dist + ((size_t)inst * n_agents + a) * cells;
            const int rr0 = pr - kR + i0, cc0 = pc - kR + j0;
            if (rr0 >= 0 && rr0 < H && cc0 >= 0 && cc0 < W) w0[q] = (int)d[rr0 * W + cc0];
            const int rr1 = pr - kR + i1, cc1 = pc - kR + j1;
            if (p1 < kWin * kWin && rr1 >= 0 && rr1 < H && cc1 >= 0 && cc1 < W) w1[q] = (int)d[rr1 * W + cc1];
        }
    }
#pragma unroll
    for (int q = 0; q < kPerWave; q++) {
        const int a = a_begin + wave + 4 * q;
        if (a >= n_agents) break;                                       // wave-uniform
        const AgentRec me = srec[a];
        const int pr = me.pr, pc = me.pc;
        const int v0 = w0[q], v1 = w1[q];

        row32[lane] = 0x42424242u;                                      // whole row <- "!" (66), cpp:375-376,386-387

        // --- neighbour candidates from the LDS-resident records ---
        int cnt = 0;
        for (int b0 = 0; b0 < n_agents; b0 += 64) {
            const int b = b0 + lane;
            bool in = false;
            int md = 0;
            if (b < n_agents) {
                const int dr = srec[b].pr - pr, dc = srec[b].pc - pc;
                in = (dr >= -kR && dr <= kR && dc >= -kR && dc <= kR);  // the 11x11 scan of cpp:492-495
                md = abs(dr) + abs(dc);                                  // cpp:498-499
            }
            const unsigned long long m = __ballot(in);
            if (in) {
                const int idx = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (idx < kMaxCand) cand[idx] = ((uint32_t)md << 16) | (uint32_t)b;
            }
            cnt += __popcll(m);
        }
        if (cnt > kMaxCand) cnt = kMaxCand;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // candidate list written by some lanes, read by all
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // --- window tokens ---
        const int mid = __shfl(v0, kR * kWin + kR);                     // centre cell = index 60, cpp:297
        row[lane] = (uint8_t)window_token(v0, mid);
        if (lane + 64 < kWin * kWin) row[lane + 64] = (uint8_t)window_token(v1, mid);

        // --- rank candidates, first 13 emit their 10-token record (cpp:352-373, 506-512) ---
        for (int c = lane; c < cnt; c += 64) {
            const uint32_t key = cand[c];
            int rank = 0;
            for (int j = 0; j < cnt; j++) rank += (cand[j] < key) ? 1 : 0;
            if (rank < kSlots) {
                const AgentRec o = srec[key & 0xffffu];
                uint8_t *q = row + kWin * kWin + 10 * rank;
                q[0] = (uint8_t)(o.pr - pr + kLimit);
                q[1] = (uint8_t)(o.pc - pc + kLimit);
                q[2] = (uint8_t)(min(max(o.gr - pr, -kLimit), kLimit) + kLimit);
                q[3] = (uint8_t)(min(max(o.gc - pc, -kLimit), kLimit) + kLimit);
                q[4] = o.hist[0]; q[5] = o.hist[1]; q[6] = o.hist[2]; q[7] = o.hist[3]; q[8] = o.hist[4];
                q[9] = o.next;
            }
        }
        // all LDS traffic above is issued by this wave in program order; make it visible to its own reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t packed = row32[lane];
        reinterpret_cast<uint32_t *>(tokens + ((size_t)inst * n_agents + a) * 256)[lane] = packed;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct mgpt_tokenizer {
    int n_inst, n_agents, H, W, n_grids;
    uint8_t *grids = nullptr;
    uint16_t *dist = nullptr;
    AgentRec *recs = nullptr;
    uint8_t *dirty = nullptr;
    bool have_grids = false, have_agents = false;
};

extern "C" int mgpt_tokenizer_create(mgpt_tokenizer **out, const mgpt_input_parameters *cfg, int n_inst,
                                     int n_agents, int H, int W, int n_grids)
{
    MGPT_REQUIRE(out && cfg, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(n_inst > 0 && n_agents > 0 && H > 0 && W > 0 && n_grids > 0 && n_grids <= n_inst, MGPT_ERR_ARG,
                 "bad sizes n_inst=%d n_agents=%d H=%d W=%d n_grids=%d", n_inst, n_agents, H, W, n_grids);
    MGPT_REQUIRE(cfg->cost2go_value_limit == kLimit && cfg->num_agents == kSlots && cfg->num_previous_actions == 5 &&
                     cfg->context_size == MGPT_CONTEXT && cfg->obs_radius == kR && cfg->agents_radius == kR,
                 MGPT_ERR_UNSUPPORTED,
                 "only the reference's InputParameters (20,13,5,256,5,5) are implemented (inference.py:15-29)");
    MGPT_REQUIRE(n_agents <= 2048, MGPT_ERR_UNSUPPORTED, "n_agents=%d > 2048", n_agents);
    // distances are uint16 as in the reference (h:73); shortest paths must stay below 65534
    MGPT_REQUIRE((int64_t)H * W <= (1 << 22), MGPT_ERR_UNSUPPORTED, "H*W=%lld cells is beyond this build's limit (4M)",
                 (long long)H * W);
    mgpt_tokenizer *t = new mgpt_tokenizer();
    t->n_inst = n_inst; t->n_agents = n_agents; t->H = H; t->W = W; t->n_grids = n_grids;
    const size_t cells = (size_t)H * W, total = (size_t)n_inst * n_agents;
    hipError_t e = hipMalloc(&t->grids, (size_t)n_grids * cells);
    if (e == hipSuccess) e = hipMalloc(&t->dist, total * cells * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(&t->recs, total * sizeof(AgentRec));
    if (e == hipSuccess) e = hipMalloc(&t->dirty, total);
    if (e != hipSuccess) {
        set_error("hipMalloc failed in mgpt_tokenizer_create: %s", hipGetErrorString(e));
        mgpt_tokenizer_destroy(t);
        return MGPT_ERR_HIP;
    }
    *out = t;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_destroy(mgpt_tokenizer *t)
{
    if (!t) return MGPT_OK;
    (void)hipFree(t->grids); (void)hipFree(t->dist); (void)hipFree(t->recs); (void)hipFree(t->dirty);
    delete t;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_set_grids(mgpt_tokenizer *t, const uint8_t *d_grids, void *stream)
{
    MGPT_REQUIRE(t && d_grids, MGPT_ERR_ARG, "NULL argument");
    MGPT_HIP(hipMemcpyAsync(t->grids, d_grids, (size_t)t->n_grids * t->H * t->W, hipMemcpyDeviceToDevice,
                            (hipStream_t)stream));
    t->have_grids = true;
    return MGPT_OK;
}

static int launch_bfs(mgpt_tokenizer *t, const uint8_t *dirty, hipStream_t s)
{
    const int total = t->n_inst * t->n_agents;
    const size_t bytes = (size_t)t->H * t->W * sizeof(uint16_t);
    ProfScope ps(P_BFS, s);
    if (bytes <= 64 * 1024) {
        hipLaunchKernelGGL(bfs_kernel<true>, dim3(total), dim3(256), bytes, s, t->grids, t->n_grids, t->n_agents, t->H,
                           t->W, t->recs, dirty, t->dist);
    } else {
        hipLaunchKernelGGL(bfs_kernel<false>, dim3(total), dim3(256), 0, s, t->grids, t->n_grids, t->n_agents, t->H,
                           t->W, t->recs, dirty, t->dist);
    }
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_create_agents(mgpt_tokenizer *t, const int16_t *d_pos, const int16_t *d_goal, void *stream)
{
    MGPT_REQUIRE(t && d_pos && d_goal, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(t->have_grids, MGPT_ERR_STATE, "mgpt_tokenizer_set_grids must precede create_agents");
    hipStream_t s = (hipStream_t)stream;
    const int total = t->n_inst * t->n_agents;
    {
        ProfScope ps(P_TOK_UPDATE, s);
        hipLaunchKernelGGL(tok_create_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, d_pos, d_goal, total);
        MGPT_LAUNCH_CHECK();
    }
    int rc = launch_bfs(t, nullptr, s);
    if (rc != MGPT_OK) return rc;
    {
        ProfScope ps(P_TOK_NEXT, s);
        hipLaunchKernelGGL(tok_next_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, total, t->dist, t->H, t->W);
        MGPT_LAUNCH_CHECK();
    }
    t->have_agents = true;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_update_agents(mgpt_tokenizer *t, const int16_t *d_pos, const int16_t *d_goal,
                                            const int32_t *d_actions, int goals_may_change, void *stream)
{
    MGPT_REQUIRE(t && d_pos && d_actions && (d_goal || !goals_may_change), MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(t->have_agents, MGPT_ERR_STATE, "create_agents must precede update_agents");
    hipStream_t s = (hipStream_t)stream;
    const int total = t->n_inst * t->n_agents;
    {
        ProfScope ps(P_TOK_UPDATE, s);
        hipLaunchKernelGGL(tok_update_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, d_pos, d_goal, d_actions,
                           t->dirty, total, goals_may_change ? 1 : 0, t->dist, t->H, t->W);
        MGPT_LAUNCH_CHECK();
    }
    if (goals_may_change) {
        int rc = launch_bfs(t, t->dirty, s);
        if (rc != MGPT_OK) return rc;
        ProfScope ps(P_TOK_NEXT, s);
        hipLaunchKernelGGL(tok_next_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, total, t->dist, t->H, t->W);
        MGPT_LAUNCH_CHECK();
    }
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_generate_observations(mgpt_tokenizer *t, uint8_t *d_tokens, void *stream)
{
    MGPT_REQUIRE(t && d_tokens, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(t->have_agents, MGPT_ERR_STATE, "create_agents must precede generate_observations");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = cdiv(t->n_agents, kAgentsPerBlock);
    const size_t smem = (size_t)t->n_agents * 16 + 4 * kMaxCand * 4 + 4 * 256;
    ProfScope ps(P_TOKENS, s);
    hipLaunchKernelGGL(tokens_kernel, dim3(t->n_inst * chunks), dim3(256), smem, s, t->recs, t->dist, t->n_agents, t->H,
                       t->W, chunks, d_tokens);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_state(mgpt_tokenizer *t, const uint16_t **d_dist, const void **d_records)
{
    MGPT_REQUIRE(t, MGPT_ERR_ARG, "NULL argument");
    if (d_dist) *d_dist = t->dist;
    if (d_records) *d_records = t->recs;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_copy_state(mgpt_tokenizer *t, uint16_t *d_dist_out, void *d_records_out, void *stream)
{
    MGPT_REQUIRE(t, MGPT_ERR_ARG, "NULL argument");
    const size_t total = (size_t)t->n_inst * t->n_agents;
    if (d_dist_out)
        MGPT_HIP(hipMemcpyAsync(d_dist_out, t->dist, total * t->H * t->W * sizeof(uint16_t), hipMemcpyDeviceToDevice,
                                (hipStream_t)stream));
    if (d_records_out)
        MGPT_HIP(hipMemcpyAsync(d_records_out, t->recs, total * sizeof(AgentRec), hipMemcpyDeviceToDevice,
                                (hipStream_t)stream));
    return MGPT_OK;
}
