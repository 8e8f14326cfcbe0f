// tokenizer.hip -- MI355X observation tokenizer (replaces mapf_gpt/observation_generator.{h,cpp}).
//
// State is a structure-of-arrays over (instance, agent) resident in HBM:
//   grids   uint8  [n_grids, H, W]            obstacle maps (instance i uses map i % n_grids)
//   dist    uint16 [n_inst, n_agents, H, W]   4-connected BFS distance-to-goal fields (65535 = wall/unreached)
//   dist8   uint8  [n_inst, n_agents, H, W]   the same fields in one byte (255 = wall/unreached), valid while every
//                                             finite distance is <= 253 (device flag `u8_ok`); halves the window traffic
//   recs    16 B   [n_inst, n_agents]         {pos, goal, 5 history tokens, greedy-bits token}
// Kernels (all integer, bit-exact against the reference):
//   bfs_kernel        one 256-thread workgroup per agent, monotone relaxation in LDS    (cpp:200-286)
//   create/update     one thread per agent                                              (cpp:391-410, 432-485)
//   tokens_kernel     one wavefront per agent row, 4 rows interleaved per wavefront, the instance's agent
//                     records staged in LDS, window gathered straight from the agent's own distance field,
//                     neighbours ranked through a distance-bucket table in LDS, row assembled in LDS
//                                                                                        (cpp:288-311, 487-528, 352-389)
//   ds_* kernels      dataset-side bulk tokenizer (dataset/tokenizer/*): all-pairs BFS table per map, one row per
//                     (agent, timestep) of a logged episode
#include "common.h"
#include <stdlib.h>
#include <type_traits>

using namespace mgpt;

namespace {

struct __attribute__((aligned(16))) AgentRec {
    int16_t pr, pc, gr, gc;
    uint8_t hist[5];
    uint8_t next;
    uint8_t org[2];     // origin (row / 64, col / 64) of the reference's cached partial window for this agent (cpp:204-207)
};
static_assert(sizeof(AgentRec) == 16, "AgentRec must be 16 bytes");

constexpr int kUnreach = 65535;
constexpr int kFreeUnset = 65534;   // transient marker inside bfs_kernel only
constexpr int kMaxU8Dist = 253;     // longest finite distance the one-byte field represents exactly
constexpr int kR = 5;               // obs_radius == agents_radius == 5 (inference.py:18-19)
constexpr int kWin = 2 * kR + 1;    // 11
constexpr int kLimit = 20;          // cost2go_value_limit (inference.py:17)
constexpr int kSlots = 13;          // num_agents (inference.py:15)
constexpr int kDefaultStep = 64;    // grid_step the reference passes (inference.py:28); run-time value: mgpt_tokenizer::step
constexpr int TOK_UNREACH = 41, TOK_NEG = 42, TOK_POS = 43, TOK_N = 44, TOK_BITS0 = 50, TOK_PAD = 66;   // the default vocabulary (dataset-side kernels)

// struct InputParameters (observation_generator.h:22-40) as the inference-side kernels see it; validated by mgpt_tokenizer_create.
// Vocabulary (Encoder::Encoder, cpp:321-350): -L..L -> 0..2L, -4L ("unreachable") -> 2L+1, -2L -> 2L+2, +2L -> 2L+3,
// n w u d l r -> 2L+4.., "0000".."1111" -> 2L+10.., "!" -> 2L+26.
struct TokCfg {
    int L;      // cost2go_value_limit
    int S;      // num_agents: neighbour records per row (<= 16)
    int Hn;     // num_previous_actions (<= 5: AgentRec::hist keeps the last five)
    int R;      // obs_radius (<= 5: two window cells per lane)
    int A;      // agents_radius (<= 5: twelve distance buckets)
    int rwin;   // 65536 / (2R + 1) + 1: cell index / window side = (index * rwin) >> 16, exact below 128 (checked at create)
    __host__ __device__ int tok_n() const { return 2 * L + 4; }
    __host__ __device__ int tok_bits0() const { return 2 * L + 10; }
};

// ---------------------------------------------------------------------------------------------
// Distance field.  The reference's tiled / border-table / priority-queue construction (cpp:43-286)
// equals the plain 4-connected BFS distance from the goal (SURVEY.md finding 4, re-verified by
// tests/test_oracle_vs_reference.py) -- except for ONE cell per cached partial window: get_cells_on_border (cpp:178-198)
// seeds every border cell of the agent's 129 x 129 window but its (right, bottom) corner, which the flood fill then reaches
// from its two in-window neighbours only (value min + 1, e.g. 2 too large when the goal lies beyond the corner).  The cell is
// in view only from (left + 123, top + 123), reached without a recompute (cpp:469-477).  The window origin is therefore kept
// per agent (AgentRec::org, maintained by create / update exactly as the reference maintains its partials) and
// tokens_kernel patches that one window cell; the distance fields themselves stay plain BFS.  tests/golden/tok_corner_*.npz.
// Here: chaotic min-plus relaxation, d[i] <- min(d[i], min_nb+1),
// until a full sweep changes nothing.  Every value ever stored is the length of a real path, the
// update is monotone, so the fixpoint is the exact shortest distance regardless of thread order.
// `d` lives in LDS when the map fits (<= 32768 cells), else directly in the output buffer.
// ---------------------------------------------------------------------------------------------
template <bool kLds>
__global__ __launch_bounds__(256) void bfs_kernel(const uint8_t *__restrict__ grids, int n_grids, int n_agents,
                                                  int H, int W, const AgentRec *__restrict__ recs,
                                                  const uint8_t *__restrict__ dirty, uint16_t *__restrict__ dist_all,
                                                  uint8_t *__restrict__ dist8_all, int *__restrict__ u8_ok)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ag = blockIdx.x;
    if (dirty != nullptr && dirty[ag] == 0) return;
    const int inst = ag / n_agents;
    const int cells = H * W;
    const uint8_t *grid = grids + (size_t)(inst % n_grids) * cells;
    uint16_t *out = dist_all + (size_t)ag * cells;
    uint8_t *out8 = dist8_all != nullptr ? dist8_all + (size_t)ag * cells : nullptr;
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
    int too_long = 0;
    for (int i = tid; i < cells; i += 256) {
        int v = d[i];
        if (v == kFreeUnset) v = kUnreach;
        out[i] = (uint16_t)v;
        too_long |= (v != kUnreach && v > kMaxU8Dist);
        if (dist8_all != nullptr) out8[i] = (uint8_t)(v == kUnreach ? 255 : min(v, 254));
    }
    if (too_long && u8_ok != nullptr) *u8_ok = 0;     // benign race: every writer stores 0
}

// greedy-direction bits, cpp:412-430: order u(-1,0) d(+1,0) l(0,-1) r(0,+1); bit = neighbour strictly closer
__device__ __forceinline__ int next_action_token(const uint16_t *__restrict__ d, int H, int W, int pr, int pc, int bits0 = TOK_BITS0)
{
    if (pr < 0 || pr >= H || pc < 0 || pc >= W) return bits0;
    const int cur = d[pr * W + pc];
    const int u = (pr > 0) ? (int)d[(pr - 1) * W + pc] : kUnreach;
    const int dn = (pr < H - 1) ? (int)d[(pr + 1) * W + pc] : kUnreach;
    const int l = (pc > 0) ? (int)d[pr * W + pc - 1] : kUnreach;
    const int rt = (pc < W - 1) ? (int)d[pr * W + pc + 1] : kUnreach;
    return bits0 + 8 * (u < cur) + 4 * (dn < cur) + 2 * (l < cur) + (rt < cur);
}

// origin of the partial window the reference computes for an agent standing at (pr, pc), cpp:204-207 (H, W <= 16384: a byte each)
__device__ __forceinline__ void window_origin(AgentRec &r, int gstep, int R)
{
    r.org[0] = (uint8_t)(max(r.pr - R, 0) / gstep);
    r.org[1] = (uint8_t)(max(r.pc - R, 0) / gstep);
}

// What tokens_kernel needs of a row before it can gather: {packed position, window origin | flags}.  Bits 0..28 of y: offset of the window's
// first cell in the agent's field (H * W <= 2^22); bit 30: window not wholly inside the frame (cells read one by one, out-of-frame = wall);
// bit 29: the window's last cell is the unseeded corner of the agent's cached partial window (cpp:178-198).  Kept per agent by create / update
// (round 6: tokens_kernel formed it per wave -- ~25 vector instructions shared by only 4 or 8 rows).
__device__ __forceinline__ uint2 row_header(const AgentRec &r, int H, int W, int gstep, int R)
{
    const int pr = r.pr, pc = r.pc;
    const bool inside = pr >= R && pr + R < H && pc >= R && pc + R < W;
    const int cr = gstep * ((int)r.org[0] + 2), cc = gstep * ((int)r.org[1] + 2);
    const bool corner = cr <= H - 1 && cc <= W - 1 && pr + R == cr && pc + R == cc;
    const uint32_t my0 = (uint32_t)(uint16_t)r.pr | ((uint32_t)(uint16_t)r.pc << 16);
    return make_uint2(my0, inside ? ((uint32_t)((pr - R) * W + (pc - R)) | (corner ? 0x20000000u : 0u)) : 0x40000000u);
}

// create_agents, cpp:391-410 (history <- "n" x 5)
__global__ __launch_bounds__(256) void tok_create_kernel(AgentRec *__restrict__ recs, const int16_t *__restrict__ pos,
                                                         const int16_t *__restrict__ goal, int total,
                                                         int *__restrict__ u8_ok, int gstep, const TokCfg cfg, uint2 *__restrict__ hdrs, int H, int W)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *u8_ok = 1;          // every field is rebuilt next; bfs_kernel clears it if one does not fit a byte
    if (i >= total) return;
    AgentRec r;
    r.pr = pos[2 * i]; r.pc = pos[2 * i + 1];
    r.gr = goal[2 * i]; r.gc = goal[2 * i + 1];
#pragma unroll
    for (int k = 0; k < 5; k++) r.hist[k] = (uint8_t)cfg.tok_n();
    r.next = (uint8_t)cfg.tok_bits0();
    window_origin(r, gstep, cfg.R);                                                                 // cpp:408
    recs[i] = r;
    hdrs[i] = row_header(r, H, W, gstep, cfg.R);
}

// update_agents, cpp:432-485: position, action history (intended action of the previous step),
// goal change detection; greedy bits here when goals are static, else in tok_next_kernel after the BFS.
__global__ __launch_bounds__(256) void tok_update_kernel(AgentRec *__restrict__ recs, const int16_t *__restrict__ pos,
                                                         const int16_t *__restrict__ goal,
                                                         const int32_t *__restrict__ actions, uint8_t *__restrict__ dirty,
                                                         int total, int check_goals, const uint16_t *__restrict__ dist,
                                                         int H, int W, int gstep, const uint8_t *__restrict__ active, int n_agents,
                                                         const TokCfg cfg, uint2 *__restrict__ hdrs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (active != nullptr && active[i / n_agents] == 0) {   // instance not presented in this call: its state does not advance
        if (check_goals) dirty[i] = 0;
        return;
    }
    AgentRec r = recs[i];
    r.pr = pos[2 * i]; r.pc = pos[2 * i + 1];
    const int act = actions[i];
    r.hist[0] = r.hist[1]; r.hist[1] = r.hist[2]; r.hist[2] = r.hist[3]; r.hist[3] = r.hist[4];   // cpp:463
    r.hist[4] = (uint8_t)((act >= 0 && act <= 4) ? cfg.tok_n() + 1 + act : cfg.tok_n());           // cpp:442-462
    bool moved_goal = false;
    if (check_goals) {
        const int16_t gr = goal[2 * i], gc = goal[2 * i + 1];
        moved_goal = gr != r.gr || gc != r.gc;
        dirty[i] = moved_goal ? 1 : 0;                                                              // cpp:464-468
        r.gr = gr; r.gc = gc;
    }
    {   // cpp:464-477: the partial window is recomputed around the new position on a goal change or when the observation
        // window leaves it; only its origin matters here (it decides which cell is the unseeded corner)
        const int left = gstep * r.org[0], top = gstep * r.org[1];
        const int right = min(left + 2 * gstep, H - 1), bottom = min(top + 2 * gstep, W - 1);
        const int R = cfg.R;
        if (moved_goal || r.pr - R < left || r.pr + R > right || r.pc - R < top || r.pc + R > bottom) window_origin(r, gstep, R);
    }
    if (!check_goals) {
        r.next = (uint8_t)next_action_token(dist + (size_t)i * H * W, H, W, r.pr, r.pc, cfg.tok_bits0());   // cpp:483-484
    }
    recs[i] = r;
    hdrs[i] = row_header(r, H, W, gstep, cfg.R);
}

__global__ __launch_bounds__(256) void tok_next_kernel(AgentRec *__restrict__ recs, int total,
                                                       const uint16_t *__restrict__ dist, int H, int W, int bits0)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pr = recs[i].pr, pc = recs[i].pc;
    recs[i].next = (uint8_t)next_action_token(dist + (size_t)i * H * W, H, W, pr, pc, bits0);
}

// ---------------------------------------------------------------------------------------------
// generate_observations, cpp:516-528.  Grid (instance, chunk); workgroup = 4 wavefronts = one chunk of 4*RPW agents of ONE
// instance; a wavefront owns RPW consecutive agents (one 256-token row each), U = 4 of them interleaved phase by phase
// (a row's chain is ~10 dependent LDS round trips).
//   LDS: the instance's agent records (16 B each) + their biased packed positions, per wave U row images (token t at byte
//        t+1 so that 10-token neighbour records are 2-byte aligned), a distance-bucket table, a (rank -> agent) list per
//        row and (KP > 1) the compacted neighbour ids.
//   HBM reads per row: its 8-byte header (scalar load), the window of the agent's own distance field (one byte per cell from
//        `dist8` when every field fits a byte, else two from `dist`); writes: one coalesced 256-B store.
// Rounds 1-5 found the kernel instruction-issue bound (111 VALU + 65 SALU per row at 192 agents), so round 6 rewrote the row body
// around the DYNAMIC instruction count (profiles/r06_pmc_insts_tokenizer.txt: 54.6 + 55.8):
//   * the two window cells of a lane are ONE packed 16-bit pair: saturating v_pk_add/sub_u16, v_pk_min_u16 -- the whole
//     clamp / sentinel / vocabulary chain is branch-free and runs once per pair; the high half is stored with
//     ds_write_b8_d16_hi (round 5's per-cell ternaries compiled to four exec-mask branches per cell);
//   * window addresses are (buffer resource of the chunk's fields) + (wave-uniform scalar offset incl. the window origin) +
//     (per-lane constant): no vector address arithmetic; the row headers (position, window origin, flags) are kept per agent
//     by create / update and reach a wave as ONE scalar load;
//   * neighbour test on packed 16-bit positions (v_pk_sub_u16 / v_pk_max_u16), the ids of the neighbours of the KP passes
//     compacted in id order by ballot + mbcnt -- the compaction area holds every candidate of the KP <= 4 instances, so no
//     per-lane overflow test; the Manhattan distance is formed once per row after the compaction (v_sad_u16);
//   * order = (Manhattan distance, agent id) ascending, first S (cpp:496-506): rank = #candidates in lower distance buckets
//     + #lower ids in the own bucket: every candidate ORs its lane bit into bucket[d] in LDS (ds_or_b64), 16 lanes prefix-sum the
//     bucket populations (DPP row scan), each candidate reads its bucket once (rank = v_mbcnt of the mask + prefix);
//   * a ranked candidate only leaves its id in list[row][rank]; the records of FOUR rows are then emitted at once by lanes
//     16 u + rank (round 5 emitted per row with <= 13 of 64 lanes active);
//   * more than 64 neighbours in one window (> 64 agents in an open room): exact slow path that walks the distance
//     buckets over all passes and fills the same list.
// What bounds it after that is no single resource (profiles/r06_tokenizer_ablation.txt): vector issue ~55 %, the CU's scalar unit
// ~50 %, LDS pipe ~70 % busy, and removing any one LDS phase, the window traffic or raising the occupancy moves the time by <= 10 %.
// KP = ceil(n_agents / 64) candidate passes per row.  All of InputParameters (limit, slots, history, radii) are kernel
// arguments: they sit in scalar registers and cost the default configuration nothing.
// ---------------------------------------------------------------------------------------------
constexpr int kRowsInterleaved = 4; // rows a wave works on at once (U)
constexpr int kRowBytes = 304;      // one row image: token t at byte t+1 (1 .. 256, read back as dwords up to byte 259); bytes 264 .. 295 are dump space.
                                    // (512 until round 6: with 304 the 192-agent instance fits EIGHT workgroups on a CU instead of six)
constexpr int kRowImage = kRowsInterleaved * kRowBytes;   // per wave
constexpr int kDumpTok = 264;       // where lanes without a window cell put their byte (inside the image, never read)
constexpr int kBktBytes = 1024;     // per wave: 64 x 16 B {mask lo, mask hi, prefix, -}; entry 16 u + d = distance bucket d of row u
constexpr int kListEntries = 16;    // per row: rank -> key (uint16; kNoKey = empty); num_agents <= 16
constexpr unsigned kNoKey = 0xffffu;
constexpr int kIdBits = 11;         // key = distance << 11 | agent id (n_agents <= 2048)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short ss2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ ss2 as_ss2(uint32_t x) { return __builtin_bit_cast(ss2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u32(ss2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t rep16(int v) { return (uint32_t)(v & 0xffff) * 0x10001u; }

template <int CTRL>
__device__ __forceinline__ int dpp_shr_add(int x)      // x + (x of the lane CTRL-0x110 to the left in its row of 16, 0 if none)
{
    return x + __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}

// one neighbour record of the DEFAULT layout = 10 tokens at row bytes 122 + 10*rank (cpp:352-373): rel pos (not clamped, within +-5),
// rel goal clamped to +-20, five history tokens oldest first, greedy-direction bits.  (dataset-side kernel; tokens_kernel has its own emission)
__device__ __forceinline__ void emit_record(uint8_t *rw, int rank, uint4 o, uint32_t my0)
{
    const ss2 mys = __builtin_bit_cast(ss2, my0);
    const ss2 rel = __builtin_bit_cast(ss2, o.x) - mys + (ss2){kLimit, kLimit};
    ss2 rg = __builtin_bit_cast(ss2, o.y) - mys;
    rg = __builtin_elementwise_min(__builtin_elementwise_max(rg, (ss2){-kLimit, -kLimit}), (ss2){kLimit, kLimit}) +
         (ss2){kLimit, kLimit};
    const uint32_t qa = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, rel),
                                              0x06040200u);                                   // {rel.r, rel.c, goal.r, goal.c}
    uint16_t *dst = reinterpret_cast<uint16_t *>(rw + 1 + kWin * kWin + 10 * rank);           // 2-byte aligned
    dst[0] = (uint16_t)qa; dst[1] = (uint16_t)(qa >> 16);
    dst[2] = (uint16_t)o.z; dst[3] = (uint16_t)(o.z >> 16);
    dst[4] = (uint16_t)o.w;
}

__device__ __forceinline__ void wave_lds_sync()        // this wave's LDS traffic above is visible to its own reads below
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KP> struct CandWidth { static constexpr int value = KP == 1 ? 0 : (KP <= 4 ? 64 * KP : 64); };

// raw buffer resource over [p, p + 2 GiB): address = base + scalar offset + per-lane offset -- a wave-uniform row / window origin
// and a per-lane constant meet in the instruction (buffer_load_ubyte v, v_off, s[rsrc], s_off offen), no vector address arithmetic
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000);
}
// one cell of a distance field: the one-byte field (255 = wall / unreached) while every finite distance fits it, else the 16-bit one
// (65535); everything after the load is the same packed 16-bit arithmetic with another sentinel
template <bool U8>
__device__ __forceinline__ uint32_t field_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    if constexpr (U8) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
    else return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, 2 * voff, 2 * soff, 0);
}

template <int KP, int RPW>
__global__ __launch_bounds__(256) void tokens_kernel(const AgentRec *__restrict__ recs, const uint16_t *__restrict__ dist,
                                                     const uint8_t *__restrict__ dist8, const int *__restrict__ u8_ok,
                                                     int n_agents, int H, int W, int chunks_per_inst,
                                                     uint8_t *__restrict__ tokens, int gstep, const TokCfg p,
                                                     const uint2 *__restrict__ hdrs)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (read here, consumed at the window gathers: the flag's round trip overlaps the record loads below.  Rounds 1-5 branched on it
    //  into two copies of the body -- one dependent memory round trip in front of everything a workgroup does)
    const bool u8 = *u8_ok != 0;
    const unsigned UNR = u8 ? 255u : (unsigned)kUnreach;
    constexpr int APB = 4 * RPW;
    constexpr int U = kRowsInterleaved;
    constexpr int CW = CandWidth<KP>::value;
    static_assert(RPW % U == 0, "RPW must be a multiple of U");
    uint32_t *hdr = reinterpret_cast<uint32_t *>(smem);                                    // [4][RPW] packed position of a wave's rows
    uint4 *srec = reinterpret_cast<uint4 *>(smem + APB * 8);                               // [n_agents]
    uint32_t *spos = reinterpret_cast<uint32_t *>(smem + APB * 8 + (size_t)n_agents * 16); // [KP*64] biased (r,c) + agents radius
    uint8_t *srow = reinterpret_cast<uint8_t *>(spos + KP * 64);                           // [4][kRowImage]
    uint8_t *sbkt = srow + 4 * kRowImage;                                                  // [4][kBktBytes]
    uint16_t *slist = reinterpret_cast<uint16_t *>(sbkt + 4 * kBktBytes);                  // [4][U][kListEntries]
    uint16_t *scand = slist + 4 * U * kListEntries;                                        // [4][U][CW] compacted neighbour ids (KP > 1)

    const int R = p.R, A = p.A, L = p.L;
    const int win = 2 * R + 1, ncell = win * win, rec_len = 5 + p.Hn;
    const int top = 2 * L + 2;                          // window arithmetic: x = clamp(v - mid + L + 1, 0, 2L + 2)
    const uint32_t top2 = rep16(top), topm2 = rep16(top - 1), one2 = 0x00010001u, lp1_2 = rep16(L + 1);
    const uint32_t unrm2 = rep16((int)UNR - 1), unrtok2 = rep16(2 * L + 1);
    const uint32_t pad4 = (uint32_t)(2 * L + 26) * 0x01010101u;          // "!" (cpp:375-376,386-387)
    const uint32_t rad2 = rep16(A), diam2 = rep16(2 * A);

    const int inst = blockIdx.x, chunk = blockIdx.y;    // (a 2-D grid: no division)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t row0 = (size_t)inst * n_agents;
    const uint4 *grec = reinterpret_cast<const uint4 *>(recs) + row0;
    const int a_begin = chunk * APB;

    // This wave's rows first: agents a_w0 .. a_w0 + RPW - 1, whose headers (kept by create / update: row_header) are RPW x 8 contiguous bytes at a
    // wave-uniform address -- SCALAR loads: the values arrive in scalar registers (no per-row v_readlane) after one short round trip, and the
    // window gathers below are in flight before the instance's records are staged for the neighbour search.  (hdrs is padded by 64 entries.)
    const int a_w0 = a_begin + wave * RPW;
    uint32_t my0s[RPW], infos[RPW];
    {
        const uint2 *hw = hdrs + (row0 + a_w0);          // uniform
#pragma unroll
        for (int q = 0; q < RPW; q++) {
            const uint2 h = hw[q];
            my0s[q] = h.x;
            infos[q] = a_w0 + q < n_agents ? h.y : 0x80000000u;      // bit 31: no such agent
        }
    }
    uint4 stage[(KP * 64 + 255) / 256];                  // the instance's records, on their way to LDS
#pragma unroll
    for (int j = 0; j < (KP * 64 + 255) / 256; j++) {
        const int i = tid + 256 * j;
        stage[j] = make_uint4(0x7fff7fffu, 0u, 0u, 0u);  // sentinel position: never inside a window (H, W <= 16384)
        if (i < n_agents) stage[j] = grec[i];            // {pr|pc<<16, gr|gc<<16, hist0..3, hist4|next<<8|org<<16}
    }
    if (lane < RPW) hdr[wave * RPW + lane] = hdrs[row0 + min(a_w0 + lane, n_agents - 1)].x;     // (read back by this wave's emission lanes only)

    uint8_t *row = srow + wave * kRowImage;
    uint4 *bkt = reinterpret_cast<uint4 *>(sbkt + wave * kBktBytes);
    uint16_t *list = slist + wave * (U * kListEntries);
    uint16_t *cand = scand + wave * (U * CW);
    const int cells = H * W;
    // window cells of this lane: index lane and lane + 64 (lanes without one repeat cell 0 and store into the dump space).
    // (Computed here with 24-bit multiplies and the host's reciprocal of the window side.  A 1-KiB table of these per-lane constants read by
    //  every wave was built and measured: 15 vector instructions fewer per wave and 104 -> 122 us per 524 160 rows -- 65 520 waves fetching
    //  the same eight cache lines serialise on their L2 channels.)
    const bool has0 = lane < ncell, has1 = lane + 64 < ncell;
    const int c0 = has0 ? lane : 0, c1 = has1 ? lane + 64 : 0;
    const int i0 = __mul24(c0, p.rwin) >> 16, j0 = c0 - __mul24(i0, win), i1 = __mul24(c1, p.rwin) >> 16, j1 = c1 - __mul24(i1, win);
    const uint32_t off0 = (uint32_t)(__mul24(i0, W) + j0), off1 = (uint32_t)(__mul24(i1, W) + j1);
    uint8_t *tok0_at = row + (has0 ? 1 + lane : kDumpTok + (lane & 15));
    uint8_t *tok1_at = row + (has1 ? 65 + lane : kDumpTok + 16 + (lane & 15));
    const int centre_lane = R * win + R;                                 // < 64 for every R <= 5
    // emission: lane 16 u + s writes record s of row u
    const int eu = lane >> 4, es = lane & 15;
    uint8_t *rec_at = row + eu * kRowBytes + 1 + ncell + rec_len * es;
    // the chunk's distance fields and token rows as buffer resources: row q of this wave sits (wave + 4 q) fields / rows in
    const __amdgpu_buffer_rsrc_t rd8 = make_rsrc(dist8 + (row0 + a_begin) * cells);       // 64 rows x H*W <= 2^22 cells ...
    const __amdgpu_buffer_rsrc_t rd16 = make_rsrc(dist + (row0 + a_begin) * cells);       // ... x 2 B < 2 GiB
    const __amdgpu_buffer_rsrc_t rtok = make_rsrc(tokens + (row0 + a_begin) * 256);
    const uint32_t lane4 = (uint32_t)lane * 4u;

    // constants of the row loop, pinned in registers (left alone, hipcc re-materialises each of them with v_mov per row)
    uint32_t padv = pad4, zerov = 0u, nokeyv = kNoKey;
    asm volatile("" : "+v"(padv), "+v"(zerov), "+v"(nokeyv));

    // Window gathers: the first two groups of U rows before anything else, then one group ahead of the group being processed.
    // (Measured, profiles/r06_tokenizer_rpw.txt: all 16 rows of a wave up front -- rounds 1-5 -- makes a launch whose workgroups are
    //  all resident at once spend memory + arithmetic instead of their maximum, 46 against 35 us per 131 072 rows of 128 agents; one
    //  group ahead only costs the large launches 3 %; 8 rows per wave with both groups up front is the fastest large-launch form.)
    uint32_t w0[RPW], w1[RPW];
    auto gather = [&](auto u8_c, const __amdgpu_buffer_rsrc_t rd, auto g_c) {
        constexpr bool U8 = decltype(u8_c)::value;
        constexpr unsigned UNRC = U8 ? 255u : (unsigned)kUnreach;
        constexpr int QB = decltype(g_c)::value;
#pragma unroll
        for (int q = QB; q < QB + U; q++) {
            const uint32_t my0 = my0s[q], info = infos[q];
            w0[q] = UNRC; w1[q] = UNRC;
            const uint32_t fld = (uint32_t)(wave * RPW + q) * (uint32_t)cells; // this row's field inside the chunk (uniform)
            if ((info >> 30) == 0) {                                        // wave-uniform; always taken for env states
                const uint32_t org = fld + (info & 0x1fffffffu);            // the window's first cell (uniform)
                w0[q] = field_load<U8>(rd, off0, org);
                w1[q] = field_load<U8>(rd, off1, org);
                if ((info & 0x20000000u) != 0 && lane == ((ncell - 1) & 63)) {
                    // rare (agent at offset (123, 123) of its cached window): the reference reaches the window's last cell only from its
                    // two in-window neighbours, whose values are exact border seeds (cpp:252-268)
                    const uint32_t oc = (uint32_t)((win - 1) * W + (win - 1));
                    const uint32_t n1 = field_load<U8>(rd, oc - W, org), n2 = field_load<U8>(rd, oc - 1, org), m = min(n1, n2);
                    uint32_t v = ncell - 1 < 64 ? w0[q] : w1[q];
                    // (one-byte field: UNR = 255, so m + 1 is kept below the sentinel; any value > centre + limit encodes the same token)
                    if (v != UNRC && v != 0) v = (m == UNRC) ? UNRC : min(m + 1, UNRC - 1);
                    if (ncell - 1 < 64) w0[q] = v; else w1[q] = v;
                }
            } else if ((info >> 31) == 0) {                                 // out-of-frame cells read as walls
                const int pr = (int16_t)(my0 & 0xffffu), pc = (int16_t)(my0 >> 16);
                const int rr0 = pr - R + i0, cc0 = pc - R + j0, rr1 = pr - R + i1, cc1 = pc - R + j1;
                if (rr0 >= 0 && rr0 < H && cc0 >= 0 && cc0 < W) w0[q] = field_load<U8>(rd, (uint32_t)(rr0 * W + cc0), fld);
                if (rr1 >= 0 && rr1 < H && cc1 >= 0 && cc1 < W) w1[q] = field_load<U8>(rd, (uint32_t)(rr1 * W + cc1), fld);
            }
        }
    };
    auto gather_group = [&](auto g_c) {
        if (u8) gather(std::true_type{}, rd8, g_c);      // uniform
        else gather(std::false_type{}, rd16, g_c);
    };
    gather_group(std::integral_constant<int, 0>{});
    if constexpr (RPW >= 8) {
        if ((infos[U] >> 31) == 0) gather_group(std::integral_constant<int, U>{});
    }
    // the instance's records and biased positions for the neighbour search (every wave's gathers are in flight by now)
#pragma unroll
    for (int j = 0; j < (KP * 64 + 255) / 256; j++) {
        const int i = tid + 256 * j;
        if (i < n_agents) srec[i] = stage[j];
        if (i < KP * 64)                                 // + (A, A): the test below is then (candidate - me) <= 2A per half
            spos[i] = as_u32(as_us2(stage[j].x ^ 0x80008000u) + as_us2(rad2));     // int16 -> order-preserving uint16
    }
    __syncthreads();
    // the candidates' biased packed positions (lane + 64 k: the same agents for every row of the instance) and their ids:
    // read once per wave, not per row and pass
    uint32_t cposr[KP], idk[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) { cposr[k] = spos[lane + 64 * k]; idk[k] = (uint32_t)(lane + 64 * k); }
    auto group = [&](auto q0_c) -> bool {
        constexpr int q0 = decltype(q0_c)::value;
        if ((infos[q0] >> 31) != 0) return false;                       // wave-uniform: past the last agent
        if constexpr (q0 + 2 * U < RPW) gather_group(std::integral_constant<int, q0 + 2 * U>{});
        reinterpret_cast<uint2 *>(bkt)[2 * lane] = make_uint2(zerov, zerov);   // entries 16u + d: lane masks of distance bucket d of row u
        list[lane] = (uint16_t)nokeyv;
        uint32_t cid[U];       // candidate of this lane for row u: agent id ...
        uint32_t cmd[U];       // ... its Manhattan distance ...
        bool have[U];          // ... if any
        int ncand[U];          // KP > 1: number of neighbours found (wave-uniform)
        uint4 *mine[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint8_t *rw = row + u * kRowBytes;
            const uint32_t my0 = my0s[q0 + u];
            // "!" (cpp:375-376, 386-387) over bytes 4 .. 259: tokens 3 .. 258; tokens 0 .. 2 are window cells, written below
#if !defined(MGPT_ABL_TOK) || (MGPT_ABL_TOK != 4)
            *reinterpret_cast<uint32_t __attribute__((may_alias)) *>(rw + 4 + 4 * lane) = padv;
#endif

            // --- window tokens (cpp:288-311 + vocabulary cpp:321-357), both cells of the lane as one packed pair ---
            const uint32_t P = (w1[q0 + u] << 16) | w0[q0 + u];
            uint32_t mid = __builtin_amdgcn_readlane(w0[q0 + u], centre_lane);          // centre cell, cpp:297
            if (mid == UNR) mid = kUnreach;                                              // agent on a wall: the 16-bit field's arithmetic
            // x = (v + L + 1) - mid, saturating both ways (the add overflows only for 16-bit values that clamp to the top anyway)
            const unsigned short mid16 = (unsigned short)mid;
            us2 x = __builtin_elementwise_sub_sat(__builtin_elementwise_add_sat(as_us2(P), as_us2(lp1_2)), (us2){mid16, mid16});
            x = __builtin_elementwise_min(x, as_us2(top2));                              // 0 .. 2L+2
            // token of x: 0 -> "-2L" (2L+2), 1 .. 2L+1 -> x - 1, 2L+2 -> "+2L" (2L+3); unreachable -> "-4L" (2L+1)
            us2 t = __builtin_elementwise_min(x - as_us2(one2), as_us2(top2));           // x = 0 wraps to 0xffff -> 2L+2
            const us2 e = __builtin_elementwise_sub_sat(x, as_us2(topm2));               // 1 iff x = 2L+2
            t = t + e + e;
            const us2 isunr = (us2){0, 0} - __builtin_elementwise_sub_sat(as_us2(P), as_us2(unrm2));   // 0xffff iff v = UNR (cpp:308-309)
            const uint32_t tt = (as_u32(isunr) & unrtok2) | (~as_u32(isunr) & as_u32(t));               // -> one bit-select
#if !defined(MGPT_ABL_TOK) || (MGPT_ABL_TOK != 1)      // LDS ablations (results wrong; tools/tok_lds_ablation.sh): 1 window-token stores, 2 bucket atomics,
            tok0_at[u * kRowBytes] = (uint8_t)tt;           // 3 emission (record gather + stores), 4 pad fill, 5 rank read-back + list
            tok1_at[u * kRowBytes] = (uint8_t)(tt >> 16);
#else
            asm volatile("" :: "v"(tt));
#endif

            // --- neighbours: the (2A+1)^2 scan of cpp:492-495 on the LDS-resident positions ---
            const uint32_t myb = my0 ^ 0x80008000u;
            ncand[u] = 0;
            if (KP == 1) {                                                                // lane == agent id
                const us2 t2 = as_us2(cposr[0]) - as_us2(myb);                            // (dr + A, dc + A) mod 2^16
                have[u] = as_u32(__builtin_elementwise_max(t2, as_us2(diam2))) == diam2;
                cid[u] = (uint32_t)lane;
                cmd[u] = __builtin_amdgcn_sad_u16(cposr[0], myb + rad2, 0u);               // |dr| + |dc|, cpp:498-499
            } else {
                bool ink[KP];
                unsigned long long bm[KP];
                us2 t2[KP];                                                               // (stage by stage: back-to-back dependent packed
#pragma unroll                                                                            //  operations cost a wait state each)
                for (int k = 0; k < KP; k++) t2[k] = as_us2(cposr[k]) - as_us2(myb);
#pragma unroll
                for (int k = 0; k < KP; k++) t2[k] = __builtin_elementwise_max(t2[k], as_us2(diam2));
#pragma unroll
                for (int k = 0; k < KP; k++) {
                    ink[k] = as_u32(t2[k]) == diam2;
                    bm[k] = __builtin_amdgcn_ballot_w64(ink[k]);
                }
#pragma unroll
                for (int k = 0; k < KP; k++) {                                            // compact the neighbours' ids, id order kept
                    const int cnt = __popcll(bm[k]);
                    const bool room = CW > 64 || ncand[u] + cnt <= 64;                    // uniform; KP <= 4: the area holds them all
                    uint16_t *dstc = cand + u * CW + ncand[u];                            // uniform
                    if (ink[k] && room)
                        dstc[__builtin_amdgcn_mbcnt_hi((uint32_t)(bm[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm[k], 0u))] = (uint16_t)idk[k];
                    ncand[u] += cnt;
                }
            }
        }
        if (KP > 1) {
            wave_lds_sync();
#pragma unroll
            for (int u = 0; u < U; u++) {
                cid[u] = cand[u * CW + lane];
                have[u] = lane < ncand[u] && ncand[u] <= 64;      // > 64 neighbours in one window: exact slow path below
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                // (lanes without a candidate read some word of the workgroup's LDS: the mask keeps a stale id inside it)
                const uint32_t bpr = spos[cid[u] & (uint32_t)((KP == 3 ? 4 : KP) * 64 - 1)];
                cmd[u] = __builtin_amdgcn_sad_u16(bpr, (my0s[q0 + u] ^ 0x80008000u) + rad2, 0u);
            }
        }
        // --- rank = position in (Manhattan, id) order (cpp:500-506) = #candidates in lower distance buckets +
        //     #lower ids in the own bucket (candidate lanes are in id order) ---
#pragma unroll
        for (int u = 0; u < U; u++) {
            mine[u] = bkt + u * 16 + cmd[u];
#if defined(MGPT_ABL_TOK) && (MGPT_ABL_TOK == 2)
            asm volatile("" :: "v"(mine[u]));
            if (false)
#else
            if (have[u])                // divergent on purpose: same-address LDS atomics serialise, so only real neighbours issue one
#endif
                __hip_atomic_fetch_or(reinterpret_cast<unsigned long long *>(mine[u]), 1ull << lane,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);        // ds_or_b64, order-independent
        }
        wave_lds_sync();
        {   // lanes 16u .. 16u+11 prefix-sum the bucket populations of row u (DPP rows are 16 lanes wide)
            const uint2 pop = *reinterpret_cast<const uint2 *>(bkt + lane);
            const int c = __popc(pop.x) + __popc(pop.y);
            int incl = dpp_shr_add<0x111>(c);
            incl = dpp_shr_add<0x112>(incl);
            incl = dpp_shr_add<0x114>(incl);
            incl = dpp_shr_add<0x118>(incl);
            reinterpret_cast<uint32_t *>(bkt + lane)[2] = (uint32_t)(incl - c);   // #candidates in lower buckets
        }
        wave_lds_sync();
        // --- the first S leave their id at list[row][rank] (cpp:506) ---
#pragma unroll
        for (int u = 0; u < U; u++) {
#if defined(MGPT_ABL_TOK) && (MGPT_ABL_TOK == 5)
            if (false) {
#else
            if (have[u]) {
#endif
                const uint4 e = *mine[u];
                // lower buckets + the bucket's candidates in lanes below this one (v_mbcnt: the lane mask is implicit)
                const uint32_t r = __builtin_amdgcn_mbcnt_hi(e.y, __builtin_amdgcn_mbcnt_lo(e.x, e.z));
                if (r < (uint32_t)p.S) list[u * kListEntries + r] = (uint16_t)cid[u];
            }
        }
        if (KP > 1) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (ncand[u] <= 64) continue;                                   // wave-uniform
                // Exact slow path (> 64 agents inside one window): walk the distance buckets over all passes,
                // recomputing the tests; ranks are known as they are met.
                const uint32_t myb = my0s[q0 + u] ^ 0x80008000u;
                int placed = 0;
#pragma unroll 1
                for (int m = 0; m <= 2 * A && placed < p.S; m++) {
#pragma unroll 1
                    for (int k = 0; k < KP && placed < p.S; k++) {
                        const uint32_t bpr = spos[lane + 64 * k];
                        const us2 t2 = as_us2(bpr) - as_us2(myb);
                        const bool hit = as_u32(__builtin_elementwise_max(t2, as_us2(diam2))) == diam2 &&
                                         (int)__builtin_amdgcn_sad_u16(bpr, myb + rad2, 0u) == m;
                        const unsigned long long bmh = __builtin_amdgcn_ballot_w64(hit);
                        const int rank = placed + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bmh >> 32),
                                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)bmh, 0u));
                        if (hit && rank < p.S) list[u * kListEntries + rank] = (uint16_t)(lane + 64 * k);
                        placed += __popcll(bmh);
                    }
                }
            }
        }
        wave_lds_sync();
        // --- records of the U rows at once: lane 16 u + s = record s of row u (cpp:352-373, 506-512): rel pos (not clamped, within
        //     +-A), rel goal clamped to +-L, the last Hn history tokens oldest first, greedy-direction bits ---
        {
            const uint32_t k16 = list[lane];
#if defined(MGPT_ABL_TOK) && (MGPT_ABL_TOK == 3)
            if (false) {
#else
            if (k16 != kNoKey) {
#endif
                const uint4 o = srec[k16];
                const uint32_t my0 = hdr[wave * RPW + q0 + eu];
                const ss2 base = as_ss2(my0) - as_ss2(rep16(L));                               // pos - L
                const ss2 rel = as_ss2(o.x) - base;                                            // (dr + L, dc + L)
                ss2 rg = as_ss2(o.y) - base;
                rg = __builtin_elementwise_min(__builtin_elementwise_max(rg, (ss2){0, 0}), as_ss2(rep16(2 * L)));
                const uint32_t qa = __builtin_amdgcn_perm(as_u32(rg), as_u32(rel), 0x06040200u);   // {rel.r, rel.c, goal.r, goal.c}
                if (p.Hn == 5) {                                                               // uniform; 2-byte aligned: 1 + ncell is even
                    uint16_t *d16 = reinterpret_cast<uint16_t *>(rec_at);
                    d16[0] = (uint16_t)qa; d16[1] = (uint16_t)(qa >> 16);
                    d16[2] = (uint16_t)o.z; d16[3] = (uint16_t)(o.z >> 16);
                    d16[4] = (uint16_t)o.w;
                } else {
                    // {hist0..4, next} as one 48-bit value, the oldest 5 - Hn tokens dropped
                    const unsigned long long hn = (((unsigned long long)(o.w & 0xffffu) << 32) | o.z) >> (8 * (5 - p.Hn));
                    rec_at[0] = (uint8_t)qa; rec_at[1] = (uint8_t)(qa >> 8); rec_at[2] = (uint8_t)(qa >> 16); rec_at[3] = (uint8_t)(qa >> 24);
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        if (j <= p.Hn) rec_at[4 + j] = (uint8_t)(hn >> (8 * j));
                }
            }
        }
        wave_lds_sync();
#pragma unroll
        for (int u = 0; u < U; u++) {
            if ((infos[q0 + u] >> 31) != 0) break;
            const uint32_t *row32 = reinterpret_cast<const uint32_t *>(row + u * kRowBytes);
            const uint32_t packed = __builtin_amdgcn_alignbyte(row32[lane + 1], row32[lane], 1);   // tokens 4*lane .. 4*lane+3
            __builtin_amdgcn_raw_buffer_store_b32(packed, rtok, lane4, (uint32_t)(wave * RPW + q0 + u) * 256u, 0);
        }
        __builtin_amdgcn_wave_barrier();
        return true;
    };
    static_assert(RPW == 4 || RPW == 8 || RPW == 16, "rows per wavefront");
    if (!group(std::integral_constant<int, 0>{})) return;
    if constexpr (RPW >= 8) { if (!group(std::integral_constant<int, 4>{})) return; }
    if constexpr (RPW >= 16) {
        if (!group(std::integral_constant<int, 8>{})) return;
        (void)group(std::integral_constant<int, 12>{});
    }
}

// ---------------------------------------------------------------------------------------------
// Dataset-side bulk tokenizer (dataset/tokenizer/generate_observations.py + cost2go.cpp + encoder.cpp): every
// (agent, timestep) of a logged episode becomes a row.  Against the inference path above:
//   * neighbours are ordered by the BFS distance from the OBSERVER'S cell to theirs, then id, unreachable ones dropped
//     (generate_observations.py:121-141) -> an all-pairs table D[cell][cell] per map (cost2go.cpp:33-42 keeps the same
//     thing as a map of matrices), built once by bfs_kernel with every cell as a source; an agent's goal field is the
//     row D[goal cell], so no per-agent BFS is needed;
//   * history = executed moves read off the path, 'n' before the episode starts, one 'w' at its last step (:206-228);
//   * goal = the path's last cell (:199).
// Rows are (timestep, agent): one "instance" per timestep, so the records of a timestep are contiguous.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ds_sources_kernel(AgentRec *__restrict__ recs, int cells, int W)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cells) return;
    AgentRec r = {};
    r.gr = (int16_t)(c / W); r.gc = (int16_t)(c - (c / W) * W);
    recs[c] = r;
}

// paths int16 [n_agents][n_steps][2] -> recs [n_steps][n_agents]
// goals (lifelong logs, generate_observations.py:55-60,143-153): the goal each agent pursued at every timestep, same shape
// as paths; NULL = the path's last cell.
__global__ __launch_bounds__(256) void ds_records_kernel(const int16_t *__restrict__ paths, const int16_t *__restrict__ goals,
                                                         int n_agents, int n_steps, int H, int W,
                                                         const uint16_t *__restrict__ allpairs, AgentRec *__restrict__ recs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_agents * n_steps) return;
    const int t = i / n_agents, a = i - t * n_agents;
    const int16_t *p = paths + (size_t)a * n_steps * 2;
    AgentRec r;
    r.pr = p[2 * t]; r.pc = p[2 * t + 1];
    if (goals != nullptr) {                                                         // generate_observations.py:196-197
        r.gr = goals[((size_t)a * n_steps + t) * 2]; r.gc = goals[((size_t)a * n_steps + t) * 2 + 1];
    } else {
        r.gr = p[2 * (n_steps - 1)]; r.gc = p[2 * (n_steps - 1) + 1];              // generate_observations.py:199
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {                                                   // slot s <-> move index i = t - 4 + s
        const int m = t - 4 + s;
        int tok = TOK_N;                                                            // before the episode: 'n' (:207)
        if (m >= 1) {
            if (m <= min(t, n_steps - 2)) {
                const int dr = p[2 * m] - p[2 * (m - 1)], dc = p[2 * m + 1] - p[2 * (m - 1) + 1];
                tok = dr < 0 ? TOK_N + 2 : dr > 0 ? TOK_N + 3 : dc < 0 ? TOK_N + 4 : dc > 0 ? TOK_N + 5 : TOK_N + 1;   // u d l r w (:11-17)
            } else {
                tok = TOK_N + 1;                                                    // the 'w' pad of the last step (:225-228)
            }
        }
        r.hist[s] = (uint8_t)tok;
    }
    const int cells = H * W;
    const bool gok = r.gr >= 0 && r.gr < H && r.gc >= 0 && r.gc < W;
    r.next = (uint8_t)(gok ? next_action_token(allpairs + (size_t)(r.gr * W + r.gc) * cells, H, W, r.pr, r.pc) : TOK_BITS0);   // :230-243
    r.org[0] = r.org[1] = 0;
    recs[i] = r;
}

constexpr int kDsAgentsPerBlock = 16;
constexpr int kDsMaxCand = 128;

// window value -> token (cost2go.cpp:72-85 + encoder.cpp:52-73): unreachable -> -80, else clamp(v - mid) to +-20 / +-40
__device__ __forceinline__ int window_token(int v, int mid)
{
    if (v == kUnreach) return TOK_UNREACH;
    const int w = v - mid;
    return w > kLimit ? TOK_POS : (w < -kLimit ? TOK_NEG : w + kLimit);
}

// The window is cut from the field of the PATH'S LAST CELL (generate_observations.py:75-78) -- in lifelong logs that is not
// the goal the record carries; only_obstacles = the mask_cost2go ablation (cost2go.cpp:52-62: cell -> 0 / 1 = blocked).
__global__ __launch_bounds__(256) void ds_tokens_kernel(const AgentRec *__restrict__ recs, const uint16_t *__restrict__ allpairs,
                                                        const int16_t *__restrict__ paths, int only_obstacles,
                                                        int n_agents, int n_steps, int H, int W, int chunks_per_step,
                                                        uint8_t *__restrict__ tokens)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *srec = reinterpret_cast<uint4 *>(smem);                                        // [n_agents]
    uint32_t *scand = reinterpret_cast<uint32_t *>(smem + (size_t)n_agents * 16);         // [4][kDsMaxCand]
    uint8_t *srow = reinterpret_cast<uint8_t *>(scand + 4 * kDsMaxCand);                  // [4][272], token t at byte t + 1
    const int t = blockIdx.x / chunks_per_step, chunk = blockIdx.x - t * chunks_per_step;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint4 *grec = reinterpret_cast<const uint4 *>(recs) + (size_t)t * n_agents;
    for (int i = tid; i < n_agents; i += 256) srec[i] = grec[i];
    __syncthreads();
    uint32_t *cand = scand + wave * kDsMaxCand;
    uint8_t *row = srow + wave * 272;
    const int cells = H * W;
    for (int q = 0; q < kDsAgentsPerBlock / 4; q++) {
        const int a = chunk * kDsAgentsPerBlock + wave + 4 * q;
        if (a >= n_agents) break;                                                         // wave-uniform
        const uint4 me = srec[a];
        const int pr = (int16_t)(me.x & 0xffffu), pc = (int16_t)(me.x >> 16);
        const int16_t *pend = paths + ((size_t)a * n_steps + (n_steps - 1)) * 2;
        const int gr = pend[0], gc = pend[1];
        const bool gok = gr >= 0 && gr < H && gc >= 0 && gc < W;
        const uint16_t *dg = allpairs + (size_t)(gok ? gr * W + gc : 0) * cells;          // distance-to-goal field = a table row
        const uint16_t *dm = allpairs + (size_t)(pr * W + pc) * cells;                    // distances from my own cell
        if (lane < 34) reinterpret_cast<uint2 *>(row)[lane] = make_uint2(0x42424242u, 0x42424242u);
        // --- window (cost2go.cpp:44-88) ---
        int v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int cell = lane + 64 * u, i = cell / kWin, j = cell - i * kWin;
            const int rr = pr - kR + i, cc = pc - kR + j;
            v[u] = (gok && cell < kWin * kWin && rr >= 0 && rr < H && cc >= 0 && cc < W) ? (int)dg[rr * W + cc] : kUnreach;
        }
        const int mid = __shfl(v[0], kR * kWin + kR);
        // --- neighbours in the window, reachable from my cell, keyed (BFS distance, id) (:121-141) ---
        int cnt = 0;
        for (int b0 = 0; b0 < n_agents; b0 += 64) {
            const int b = b0 + lane;
            bool in = false;
            uint32_t key = 0;
            if (b < n_agents) {
                const uint32_t bp = srec[b].x;
                const int br = (int16_t)(bp & 0xffffu), bc = (int16_t)(bp >> 16);
                if (abs(br - pr) <= kR && abs(bc - pc) <= kR && br >= 0 && br < H && bc >= 0 && bc < W) {
                    const int d = dm[br * W + bc];
                    in = d != kUnreach;
                    key = ((uint32_t)d << 11) | (uint32_t)b;
                }
            }
            const unsigned long long bm = __ballot(in);
            if (in) {
                const int at = cnt + __popcll(bm & ((1ull << lane) - 1ull));
                if (at < kDsMaxCand) cand[at] = key;
            }
            cnt += __popcll(bm);
        }
        cnt = min(cnt, kDsMaxCand);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (only_obstacles) {                                                             // tokens of the integers 1 / 0
            row[1 + lane] = (uint8_t)(kLimit + (v[0] == kUnreach ? 1 : 0));
            if (lane + 64 < kWin * kWin) row[65 + lane] = (uint8_t)(kLimit + (v[1] == kUnreach ? 1 : 0));
        } else {
            row[1 + lane] = (uint8_t)window_token(v[0], mid);
            if (lane + 64 < kWin * kWin) row[65 + lane] = (uint8_t)window_token(v[1], mid);
        }
        for (int c = lane; c < cnt; c += 64) {
            const uint32_t key = cand[c];
            int rank = 0;
            for (int j = 0; j < cnt; j++) rank += (cand[j] < key) ? 1 : 0;
            if (rank < kSlots) emit_record(row, rank, srec[key & 0x7ffu], me.x);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t *row32 = reinterpret_cast<const uint32_t *>(row);
        const uint32_t packed = __builtin_amdgcn_alignbyte(row32[lane + 1], row32[lane], 1);
        reinterpret_cast<uint32_t *>(tokens + ((size_t)a * n_steps + t) * 256)[lane] = packed;      // [agent][timestep][256]
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
struct mgpt_tokenizer {
    int n_inst, n_agents, H, W, n_grids;
    int step = kDefaultStep;            // grid_step: side of the reference's cost-to-go tiles (decides the unseeded window corner)
    TokCfg cfg = {kLimit, kSlots, 5, kR, kR, 65536 / kWin + 1};
    uint8_t *grids = nullptr;
    uint16_t *dist = nullptr;
    uint8_t *dist8 = nullptr;
    int *u8_ok = nullptr;
    AgentRec *recs = nullptr;
    uint8_t *dirty = nullptr;
    uint2 *hdrs = nullptr;              // per agent: row_header (position, window origin, flags), kept by create / update for tokens_kernel
    bool have_grids = false, have_agents = false;
};

extern "C" int mgpt_tokenizer_create(mgpt_tokenizer **out, const mgpt_input_parameters *cfg, int n_inst,
                                     int n_agents, int H, int W, int n_grids)
{
    MGPT_REQUIRE(out && cfg, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(n_inst > 0 && n_agents > 0 && H > 0 && W > 0 && n_grids > 0 && n_grids <= n_inst, MGPT_ERR_ARG,
                 "bad sizes n_inst=%d n_agents=%d H=%d W=%d n_grids=%d", n_inst, n_agents, H, W, n_grids);
    // InputParameters (h:22-40).  What the kernels' layouts bound: ids below 256 (2L + 26 <= 255), AgentRec::hist (five tokens),
    // two window cells per lane ((2R+1)^2 <= 128), twelve distance buckets (2A <= 10), sixteen emission lanes per row, and a
    // row of exactly 256 tokens (cpp:386-387 pads to 256 whatever context_size says; a longer row has no place in this ABI).
    // agents_radius > cost2go_value_limit makes the reference itself throw (int_vocab.at of a relative position, cpp:358-359).
    MGPT_REQUIRE(cfg->context_size == MGPT_CONTEXT, MGPT_ERR_UNSUPPORTED, "context_size=%d: rows are %d tokens (cpp:386)",
                 cfg->context_size, MGPT_CONTEXT);
    MGPT_REQUIRE(cfg->cost2go_value_limit >= 1 && cfg->cost2go_value_limit <= 100 && cfg->num_agents >= 1 && cfg->num_agents <= 16 &&
                     cfg->num_previous_actions >= 0 && cfg->num_previous_actions <= 5 && cfg->obs_radius >= 1 && cfg->obs_radius <= 5 &&
                     cfg->agents_radius >= 0 && cfg->agents_radius <= 5 && cfg->agents_radius <= cfg->cost2go_value_limit,
                 MGPT_ERR_UNSUPPORTED,
                 "InputParameters (limit %d, agents %d, previous actions %d, obs radius %d, agents radius %d) outside the implemented ranges "
                 "(1..100, 1..16, 0..5, 1..5, 0..min(5, limit))",
                 cfg->cost2go_value_limit, cfg->num_agents, cfg->num_previous_actions, cfg->obs_radius, cfg->agents_radius);
    MGPT_REQUIRE((2 * cfg->obs_radius + 1) * (2 * cfg->obs_radius + 1) + cfg->num_agents * (5 + cfg->num_previous_actions) <= MGPT_CONTEXT,
                 MGPT_ERR_UNSUPPORTED, "window %d + %d records of %d tokens do not fit a %d-token row",
                 (2 * cfg->obs_radius + 1) * (2 * cfg->obs_radius + 1), cfg->num_agents, 5 + cfg->num_previous_actions, MGPT_CONTEXT);
    MGPT_REQUIRE(cfg->grid_step > 0, MGPT_ERR_ARG, "grid_step=%d", cfg->grid_step);
    // the per-agent window origin is kept as (row / grid_step, col / grid_step) in one byte each
    MGPT_REQUIRE((int64_t)256 * cfg->grid_step >= H && (int64_t)256 * cfg->grid_step >= W, MGPT_ERR_UNSUPPORTED,
                 "grid_step=%d is too small for a %d x %d map (more than 256 tiles per side)", cfg->grid_step, H, W);
    MGPT_REQUIRE(n_agents <= 2048, MGPT_ERR_UNSUPPORTED, "n_agents=%d > 2048", n_agents);
    MGPT_REQUIRE(H <= 16384 && W <= 16384, MGPT_ERR_UNSUPPORTED, "H=%d W=%d beyond 16384", H, W);
    // distances are uint16 as in the reference (h:73); shortest paths must stay below 65534
    MGPT_REQUIRE((int64_t)H * W <= (1 << 22), MGPT_ERR_UNSUPPORTED, "H*W=%lld cells is beyond this build's limit (4M)",
                 (long long)H * W);
    mgpt_tokenizer *t = new mgpt_tokenizer();
    t->n_inst = n_inst; t->n_agents = n_agents; t->H = H; t->W = W; t->n_grids = n_grids;
    t->step = cfg->grid_step;
    t->cfg = TokCfg{cfg->cost2go_value_limit, cfg->num_agents, cfg->num_previous_actions, cfg->obs_radius, cfg->agents_radius,
                    65536 / (2 * cfg->obs_radius + 1) + 1};
    for (int c = 0; c < 128; c++)                       // the kernel's division of a cell index by the window side
        if (((c * t->cfg.rwin) >> 16) != c / (2 * cfg->obs_radius + 1)) {
            set_error("reciprocal of the window side inexact at %d", c);
            delete t;
            return MGPT_ERR_UNSUPPORTED;
        }
    const size_t cells = (size_t)H * W, total = (size_t)n_inst * n_agents;
    hipError_t e = hipMalloc(&t->grids, (size_t)n_grids * cells);
    if (e == hipSuccess) e = hipMalloc(&t->dist, total * cells * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(&t->dist8, total * cells);
    if (e == hipSuccess) e = hipMalloc(&t->u8_ok, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&t->recs, total * sizeof(AgentRec));
    if (e == hipSuccess) e = hipMalloc(&t->dirty, total);
    if (e == hipSuccess) e = hipMalloc(&t->hdrs, (total + 64) * sizeof(uint2));     // (+ 64: a wave's scalar loads of its rows' headers may run past the last agent)
    if (e == hipSuccess) e = hipMemset(t->hdrs, 0, (total + 64) * sizeof(uint2));
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
    (void)hipFree(t->grids); (void)hipFree(t->dist); (void)hipFree(t->dist8); (void)hipFree(t->u8_ok);
    (void)hipFree(t->recs); (void)hipFree(t->dirty); (void)hipFree(t->hdrs);
    delete t;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_vocab_size(const mgpt_tokenizer *t, int *out)
{
    MGPT_REQUIRE(t && out, MGPT_ERR_ARG, "NULL argument");
    *out = 2 * t->cfg.L + 27;                            // cpp:321-350: 2 L + 1 integers, -80 / -40 / +40, 6 actions, 16 direction strings, "!"
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
                           t->W, t->recs, dirty, t->dist, t->dist8, t->u8_ok);
    } else {
        hipLaunchKernelGGL(bfs_kernel<false>, dim3(total), dim3(256), 0, s, t->grids, t->n_grids, t->n_agents, t->H,
                           t->W, t->recs, dirty, t->dist, t->dist8, t->u8_ok);
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
        hipLaunchKernelGGL(tok_create_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, d_pos, d_goal, total,
                           t->u8_ok, t->step, t->cfg, t->hdrs, t->H, t->W);
        MGPT_LAUNCH_CHECK();
    }
    int rc = launch_bfs(t, nullptr, s);
    if (rc != MGPT_OK) return rc;
    {
        ProfScope ps(P_TOK_NEXT, s);
        hipLaunchKernelGGL(tok_next_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, total, t->dist, t->H, t->W, t->cfg.tok_bits0());
        MGPT_LAUNCH_CHECK();
    }
    t->have_agents = true;
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_update_agents(mgpt_tokenizer *t, const int16_t *d_pos, const int16_t *d_goal,
                                            const int32_t *d_actions, int goals_may_change, void *stream)
{
    return mgpt_tokenizer_update_agents_masked(t, d_pos, d_goal, d_actions, nullptr, goals_may_change, stream);
}

extern "C" int mgpt_tokenizer_update_agents_masked(mgpt_tokenizer *t, const int16_t *d_pos, const int16_t *d_goal,
                                                   const int32_t *d_actions, const uint8_t *d_active, int goals_may_change,
                                                   void *stream)
{
    MGPT_REQUIRE(t && d_pos && d_actions && (d_goal || !goals_may_change), MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(t->have_agents, MGPT_ERR_STATE, "create_agents must precede update_agents");
    hipStream_t s = (hipStream_t)stream;
    const int total = t->n_inst * t->n_agents;
    {
        ProfScope ps(P_TOK_UPDATE, s);
        hipLaunchKernelGGL(tok_update_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, d_pos, d_goal, d_actions,
                           t->dirty, total, goals_may_change ? 1 : 0, t->dist, t->H, t->W, t->step, d_active, t->n_agents, t->cfg, t->hdrs);
        MGPT_LAUNCH_CHECK();
    }
    if (goals_may_change) {
        int rc = launch_bfs(t, t->dirty, s);
        if (rc != MGPT_OK) return rc;
        ProfScope ps(P_TOK_NEXT, s);
        hipLaunchKernelGGL(tok_next_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, t->recs, total, t->dist, t->H, t->W, t->cfg.tok_bits0());
        MGPT_LAUNCH_CHECK();
    }
    return MGPT_OK;
}

extern "C" int mgpt_tokenizer_generate_observations(mgpt_tokenizer *t, uint8_t *d_tokens, void *stream)
{
    MGPT_REQUIRE(t && d_tokens, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(t->have_agents, MGPT_ERR_STATE, "create_agents must precede generate_observations");
    hipStream_t s = (hipStream_t)stream;
    const int kp = cdiv(t->n_agents, 64);
    const int kpp = kp <= 4 ? kp : kp <= 8 ? 8 : kp <= 16 ? 16 : 32;
    // rows per wavefront: 8 (more bytes in flight, records staged once per 32 rows) when the launch still fills the GPU
    const int64_t wg64 = (int64_t)t->n_inst * cdiv(t->n_agents, 64);               // workgroups at 64 rows each
    int rpw = wg64 >= 4096 ? 8 : 4;
    {
        static const int forced = [] { const char *e = getenv("MGPT_TOK_RPW"); return e ? atoi(e) : 0; }();   // experiments only
        if (forced == 4 || forced == 8 || forced == 16) rpw = forced;
    }
    const int apb = 4 * rpw;
    const int chunks = cdiv(t->n_agents, apb);
    const int cw = kpp == 1 ? 0 : (kpp <= 4 ? 64 * kpp : 64);                      // CandWidth<KP>
    const size_t smem = (size_t)apb * 8 + (size_t)t->n_agents * 16 + (size_t)kpp * 64 * 4 + 4 * kRowImage + 4 * kBktBytes +
                        4 * kRowsInterleaved * kListEntries * 2 + (size_t)4 * kRowsInterleaved * cw * 2;
    ProfScope ps(P_TOKENS, s);
#define MGPT_TOKENS_R(KP_, RPW_)                                                                                       \
    hipLaunchKernelGGL((tokens_kernel<KP_, RPW_>), dim3(t->n_inst, chunks), dim3(256), smem, s, t->recs, t->dist,      \
                       t->dist8, t->u8_ok, t->n_agents, t->H, t->W, chunks, d_tokens, t->step, t->cfg, t->hdrs)
#define MGPT_TOKENS(KP_)                                                                                              \
    do {                                                                                                              \
        if (rpw == 16) MGPT_TOKENS_R(KP_, 16);                                                                        \
        else if (rpw == 8) MGPT_TOKENS_R(KP_, 8);                                                                     \
        else MGPT_TOKENS_R(KP_, 4);                                                                                   \
    } while (0)
    switch (kpp) {
    case 1: MGPT_TOKENS(1); break;
    case 2: MGPT_TOKENS(2); break;
    case 3: MGPT_TOKENS(3); break;
    case 4: MGPT_TOKENS(4); break;
    case 8: MGPT_TOKENS(8); break;
    case 16: MGPT_TOKENS(16); break;
    default: MGPT_TOKENS(32); break;
    }
#undef MGPT_TOKENS_R
#undef MGPT_TOKENS
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

// ---------------------------------------------------------------------------------------------
// Dataset-side bulk tokenizer: C ABI
// ---------------------------------------------------------------------------------------------
struct mgpt_dataset {
    int H, W;
    uint16_t *allpairs = nullptr;       // [H*W][H*W]
    AgentRec *recs = nullptr;           // scratch, grows
    size_t recs_cap = 0;
};

extern "C" int mgpt_dataset_create(mgpt_dataset **out, const uint8_t *d_grid, int H, int W, void *stream)
{
    MGPT_REQUIRE(out && d_grid, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(H > 0 && W > 0 && (int64_t)H * W <= 8192, MGPT_ERR_UNSUPPORTED,
                 "H*W=%lld: the all-pairs table is built for maps up to 8192 padded cells", (long long)H * W);
    hipStream_t s = (hipStream_t)stream;
    mgpt_dataset *d = new mgpt_dataset();
    d->H = H; d->W = W;
    const int cells = H * W;
    AgentRec *src = nullptr;
    hipError_t e = hipMalloc(&d->allpairs, (size_t)cells * cells * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMalloc(&src, (size_t)cells * sizeof(AgentRec));
    if (e != hipSuccess) {
        set_error("hipMalloc failed in mgpt_dataset_create: %s", hipGetErrorString(e));
        (void)hipFree(src); (void)hipFree(d->allpairs); delete d;
        return MGPT_ERR_HIP;
    }
    hipLaunchKernelGGL(ds_sources_kernel, dim3(cdiv(cells, 256)), dim3(256), 0, s, src, cells, W);
    {   // one BFS field per source cell (cost2go.cpp:33-42); blocked sources are never looked up
        ProfScope ps(P_BFS, s);
        const size_t bytes = (size_t)cells * sizeof(uint16_t);
        hipLaunchKernelGGL(bfs_kernel<true>, dim3(cells), dim3(256), bytes, s, d_grid, 1, cells, H, W, src, (const uint8_t *)nullptr,
                           d->allpairs, (uint8_t *)nullptr, (int *)nullptr);
    }
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(s);              // `src` is freed below
    (void)hipFree(src);
    if (le != hipSuccess || se != hipSuccess) {
        set_error("mgpt_dataset_create: %s", hipGetErrorString(le != hipSuccess ? le : se));
        (void)hipFree(d->allpairs); delete d;
        return MGPT_ERR_HIP;
    }
    *out = d;
    return MGPT_OK;
}

extern "C" int mgpt_dataset_destroy(mgpt_dataset *d)
{
    if (!d) return MGPT_OK;
    (void)hipFree(d->allpairs); (void)hipFree(d->recs);
    delete d;
    return MGPT_OK;
}

extern "C" int mgpt_dataset_tokenize(mgpt_dataset *d, int n_agents, int n_steps, const int16_t *d_paths, uint8_t *d_tokens, void *stream)
{
    return mgpt_dataset_tokenize_ex(d, n_agents, n_steps, d_paths, nullptr, 0, d_tokens, stream);
}

extern "C" int mgpt_dataset_tokenize_ex(mgpt_dataset *d, int n_agents, int n_steps, const int16_t *d_paths, const int16_t *d_goals,
                                        int only_obstacles, uint8_t *d_tokens, void *stream)
{
    MGPT_REQUIRE(d && d_paths && d_tokens, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(n_agents > 0 && n_agents <= 2048 && n_steps > 0, MGPT_ERR_ARG, "n_agents=%d n_steps=%d", n_agents, n_steps);
    hipStream_t s = (hipStream_t)stream;
    const size_t total = (size_t)n_agents * n_steps;
    if (total > d->recs_cap) {
        MGPT_HIP(hipStreamSynchronize(s));
        (void)hipFree(d->recs);
        d->recs = nullptr; d->recs_cap = 0;
        MGPT_HIP(hipMalloc(&d->recs, total * sizeof(AgentRec)));
        d->recs_cap = total;
    }
    {
        ProfScope ps(P_TOK_UPDATE, s);
        hipLaunchKernelGGL(ds_records_kernel, dim3(cdiv((int)total, 256)), dim3(256), 0, s, d_paths, d_goals, n_agents, n_steps, d->H, d->W,
                           d->allpairs, d->recs);
        MGPT_LAUNCH_CHECK();
    }
    const int chunks = cdiv(n_agents, kDsAgentsPerBlock);
    const size_t smem = (size_t)n_agents * 16 + 4 * kDsMaxCand * 4 + 4 * 272;
    ProfScope ps(P_TOKENS, s);
    hipLaunchKernelGGL(ds_tokens_kernel, dim3(n_steps * chunks), dim3(256), smem, s, d->recs, d->allpairs, d_paths, only_obstacles ? 1 : 0,
                       n_agents, n_steps, d->H, d->W, chunks, d_tokens);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

