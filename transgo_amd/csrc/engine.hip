// engine.hip -- batched WP_MCTS over G concurrent games: tree kernels + the tg_sp_* entry points.
//
// One 64-lane workgroup per game in every kernel.  A "wave" of the reference (`WP_MCTS.run`, self_play.py:607-654) is
// split at the network call into k_collect (selection, leaf stepping, pseudo-expansion, WU-UCT counters, terminal
// backups, feature encoding straight into the evaluation batch) and k_absorb (counter revert, prior renormalisation,
// expansion, value backup).  Between the two the batch is evaluated by the network (net.hip) or, for parity tests,
// by values injected through tg_sp_set_eval.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ctx.h"
#include "engine.h"
#include "tree_dev.h"

using namespace tg;

namespace {

// In-kernel phase stamps of k_collect (diagnostic builds only: -DTG_TREE_STAMP; scripts/stamp_tree.py prints them).
#ifdef TG_TREE_STAMP
__device__ unsigned long long g_stamp[16], g_stamp2[8];
#define TG_ST_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_t0 = st_t
#define TG_ST(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_acc[i] += n_ - st_t; st_t = n_; } while (0)
#define TG_ST_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_stamp[i_], st_acc[i_]); \
                            atomicAdd(&g_stamp[8], __builtin_amdgcn_s_memtime() - st_t0); atomicAdd(&g_stamp[9], 1ull); \
                            for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_stamp2[i_], bw.st2[i_]); } } while (0)
#else
#define TG_ST_DECL
#define TG_ST(i)
#define TG_ST_FLUSH()
#endif

// ---- device helpers -------------------------------------------------------------------------------------------------------
#ifndef TG_PENDING_ATOMIC
#define TG_PENDING_ATOMIC 0                           // 1: WU-UCT counters as no-return atomics -- measured 1-2 % SLOWER on the tree stage than the
                                                     // read-modify-write (0.207 vs 0.203 ms per wave, profiles/r4_ab_tree_counters.txt): not the default
#endif
#ifndef TG_POOL_PREFETCH
#define TG_POOL_PREFETCH 0                            // 1: k_collect pops the chunk a wave may need at its start -- measured 2 % SLOWER on the tree stage
                                                     // (0.2049 vs 0.2006 ms per wave, profiles/r4_ab_tree_prefetch.txt: a third of the games draw a
                                                     // ticket every wave instead of a tenth): not the default
#endif
constexpr int kMaxPath = 512;                        // longest selection path kept in LDS (>= SearchCfg::maxd, checked at create)
// most chunks one game's tree may own (>= SearchCfg::max_chunks, checked at create).  k_play keeps two ints per chunk of the new tree
// in LDS; at 9x9 all 4096 single-wave workgroups of a launch should be resident at once (16 per CU), so the table stays small there
template <int S> struct ChunkCap { static constexpr int N = S == 9 ? 512 : 2048; };

// The record lane `src` (wave-uniform) holds, for every lane.
__device__ __forceinline__ NodeRec bcast_rec(const NodeRec& r, int src) {
    struct Words { uint32_t w[8]; };
    Words in = __builtin_bit_cast(Words, r), out;
#pragma unroll
    for (int i = 0; i < 8; ++i) out.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)in.w[i], src);
    return __builtin_bit_cast(NodeRec, out);
}

template <int S> __device__ __forceinline__ BlockHdr<S>* hdr_of(NodeRec* arena, int blk) {
    return reinterpret_cast<BlockHdr<S>*>(arena + blk);
}

// ---- chunk pool (tree_dev.h) -------------------------------------------------------------------------------------------
// One chunk id off the free ring, or -1 when no chunk is visible (lane 0 calls it).  Only entries published before this launch
// are taken, so the entry read here was written by an earlier kernel.  ONE fetch-add per pop: a ticket below `visible` is a chunk,
// a ticket at or beyond it is a failure (the pool is dry) and k_pool_publish takes the overshoot back before it publishes more.
// (A compare-and-swap loop here cost 0.7 ms per search wave: ~1300 games ask for a chunk in the same wave, and every failed
// exchange retries against the same address.)
__device__ __forceinline__ int pool_pop(const EngineDev& d, unsigned long long vis) {
    PoolCtl* pc = d.pool;
    const unsigned long long h = atomicAdd(&pc->head, 1ull);
    if (h >= vis) { atomicAdd(&pc->exhausted, 1ull); return -1; }
    return d.ring[h % (unsigned long long)d.sc.pool_chunks];
}
// n (+ n2) chunk ids back onto the ring with ONE draw on `tail` (whole wave; poppable once k_pool_publish has run behind this
// kernel).  Returning atomics of 4096 workgroups on one address cost ~12 ns each: every draw saved is 50 us of a k_play launch.
__device__ __forceinline__ void pool_push(const EngineDev& d, const int32_t* ids, int n, const int32_t* ids2 = nullptr, int n2 = 0) {
    if (n < 0) n = 0;
    if (n2 < 0) n2 = 0;
    if (n + n2 <= 0) return;
    const int lane = lane_id();
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(&d.pool->tail, (unsigned long long)(n + n2));
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
    base = ((unsigned long long)hi << 32) | lo;
    for (int i = lane; i < n; i += 64) d.ring[(base + i) % (unsigned long long)d.sc.pool_chunks] = ids[i];
    for (int i = lane; i < n2; i += 64) d.ring[(base + n + i) % (unsigned long long)d.sc.pool_chunks] = ids2[i];
}
// every chunk free (creation, and a reset of all games); the statistics survive a reset
__global__ void k_pool_init(EngineDev d, int first) {
    const int n = d.sc.pool_chunks;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d.ring[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PoolCtl pc; pc.head = 0; pc.tail = pc.visible = (unsigned long long)n;
        pc.min_free = first ? (unsigned long long)n : d.pool->min_free; pc.exhausted = first ? 0 : d.pool->exhausted;
        *d.pool = pc;
    }
}
// behind every kernel that pushes (k_play, k_reset): what it pushed becomes poppable; the fill just before is the cycle's peak
__global__ void k_pool_publish(EngineDev d) {
    PoolCtl* pc = d.pool;
    if (pc->head > pc->visible) pc->head = pc->visible;             // tickets drawn on a dry pool were failures, not pops
    const unsigned long long fr = pc->visible - pc->head;
    if (fr < pc->min_free) pc->min_free = fr;
    pc->visible = pc->tail;
}

// A tree under construction / growing: the chunk-id list it records its chunks in and the bump pointer inside the newest chunk.
// The room left in that chunk follows from the pointer alone (chunks are a power of two and a fresh chunk is used at once, so a
// pointer ON a chunk boundary always means "full" -- or "no chunk yet").  `ctl` set (k_collect): the chunk count and the game's
// high-water are written through to GameCtl on the rare new-chunk path.
struct TreeAlloc {
    int32_t* ids; int n_chunks, free_slot; GameCtl* ctl;
    unsigned long long visible;      // PoolCtl::visible, read once at the top of the kernel (constant during a launch): off the pop's critical path
    int spare;                       // a chunk popped ahead of need (-1: none); k_collect only (GameCtl::spare between launches)
    const int* rsv; int rsv_n, rsv_i; // chunk ids reserved with ONE ticket draw (k_play: LDS table), handed out before single pops
};
// n contiguous slots (n <= chunk_slots) for the tree, from its current chunk or a fresh one off the pool; -1 = the game's cap
// (max_chunks) or the pool is exhausted.  Called by the whole wave; every lane gets the same answer.
__device__ __forceinline__ int tree_alloc(const EngineDev& d, TreeAlloc& al, int n) {
    const int room = (-al.free_slot) & (d.sc.chunk_slots - 1);
    if (n > room) {
        int id = -1;
        if (lane_id() == 0) {
            const int have = al.n_chunks;
            if (have < d.sc.max_chunks) id = al.spare >= 0 ? al.spare : al.rsv_i < al.rsv_n ? al.rsv[al.rsv_i] : pool_pop(d, al.visible);
            if (id >= 0) {
                al.ids[have] = id;
                if (al.ctl) { al.ctl->n_chunks = have + 1; const int hw = (have + 1) * d.sc.chunk_slots; if (hw > al.ctl->hw_slot) al.ctl->hw_slot = hw; }
            }
        }
        id = __builtin_amdgcn_readfirstlane(id);
        if (id < 0) return -1;
        if (al.spare >= 0) al.spare = -1; else if (al.rsv_i < al.rsv_n) ++al.rsv_i;
        ++al.n_chunks;
        al.free_slot = id * d.sc.chunk_slots;
    }
    const int at = al.free_slot;
    al.free_slot = __builtin_amdgcn_readfirstlane(al.free_slot + n);
    return at;
}
__device__ __forceinline__ int32_t* chunk_list(const EngineDev& d, int g, int which) {
    return d.chunk_ids + ((size_t)g * 2 + which) * (size_t)d.sc.max_chunks;
}

// Allocate and write the block of a node whose position `st` is loaded + analysed in `bw`: header + one fresh child
// record per legal action (Node_V.expand with prior 0.0, self_play.py:70-77, :634-636).  Returns the block slot or -1.
template <int S>
__device__ __forceinline__ int make_block(BoardWave<S>& bw, const BoardState<S>& st, NodeRec* arena, const EngineDev& d, TreeAlloc& al, bool children) {
    using G = Geo<S>;
    constexpr int HS = TreeGeo<S>::HS;
    uint64_t lw[G::NW];
    int npts = 0;
    TG_BW_ST_BEGIN(bw);
    if (children) {
        legal_words(bw, st, lw);
        TG_BW_ST(bw, 5);
#pragma unroll
        for (int k = 0; k < G::NW; ++k) npts += __popcll(lw[k]);
    } else {
#pragma unroll
        for (int k = 0; k < G::NW; ++k) lw[k] = 0;
    }
    const int nchild = children ? (npts > 0 ? npts : 1) : 0;          // environment.py:121-129: pass only if alone
    const int blk = tree_alloc(d, al, HS + nchild);
    if (blk < 0) return -1;
    const int lane = bw.lane;
    if (lane == 0) {
        BlockHdr<S> h;
        h.st = st; h.nchild = nchild; h.pad[0] = h.pad[1] = h.pad[2] = 0;
        *hdr_of<S>(arena, blk) = h;
    }
    NodeRec r;
    r.prior = 0.0; r.w = -0.0f; r.var = 0.f; r.n = 0; r.pending = 0; r.block = -1; r.flags = 0; r.term = 0;
    int base = 0;
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        if ((lw[k] >> lane) & 1) {
            int idx = base + __popcll(lw[k] & ((1ull << lane) - 1ull));
            r.action = (uint16_t)bw.pt[k];
            arena[blk + HS + idx] = r;
        }
        base += __popcll(lw[k]);
    }
    if (children && npts == 0 && lane == 0) { r.action = (uint16_t)G::P; arena[blk + HS] = r; }
    TG_BW_ST(bw, 6);
    return blk;
}

// ---- kernels ---------------------------------------------------------------------------------------------------------------

// Before a masked reset: the restarted games hand their chunks back (visible to k_reset's pops after k_pool_publish -- a pool that
// ran completely dry must still be able to restart the games it parked).
__global__ __launch_bounds__(64) void k_release(EngineDev d, const uint8_t* mask) {
    const int g = blockIdx.x;
    if (mask && !mask[g]) return;
    GameCtl* c = &d.ctl[g];
    const int n = c->n_chunks, sp = c->spare;
    pool_push(d, chunk_list(d, g, c->cur), n, &c->spare, sp >= 0 ? 1 : 0);
    __syncthreads();
    if (lane_id() == 0) { c->n_chunks = 0; c->free_slot = 0; c->spare = -1; }
}

// states == nullptr: empty boards (reset_root, self_play.py:595-598).  Otherwise the root of every masked game is the given
// position (select_action, self_play.py:689-700) and unmasked slots are parked (they take no part in searches).
template <int S>
__global__ __launch_bounds__(64) void k_reset(EngineDev d, const uint8_t* mask, const BoardState<S>* states) {
    using G = Geo<S>;
    __shared__ WaveLds<S> lds;
    const int g = blockIdx.x;
    __shared__ uint32_t bits_s[(13 * G::P + 31) / 32];
    if (lane_id() == 0) d.game_nslot[g] = 0;
    if (mask && !mask[g]) {
        if (states && lane_id() == 0) { d.ctl[g].finished = 1; d.ctl[g].searching = 0; d.ctl[g].need_eval = 0; }
        return;
    }
    BoardWave<S> bw; bw.init(&lds);
    GameCtl c = d.ctl[g];                                             // cumulative statistics survive a reset
    if (c.error && lane_id() == 0) atomicSub(&d.counters[CNT_ERRORS], 1);   // CNT_ERRORS = games parked in error right now
    // (the old tree's chunks went back to the pool in k_release, or the ring was just rebuilt with every chunk in it)
    c.cur = 0; c.free_slot = 0; c.n_chunks = 0; c.root = 0; c.spare = -1;
    c.n_target = 0; c.active = 0; c.n_paths = 0; c.need_eval = 0; c.root_row = 0;
    c.finished = 0; c.error = 0; c.searching = 0; c.moves = 0;
    NodeRec* arena = d.arena;
    TreeAlloc al; al.ids = chunk_list(d, g, 0); al.n_chunks = 0; al.free_slot = 0; al.ctl = nullptr; al.visible = d.pool->visible; al.spare = -1; al.rsv = nullptr; al.rsv_n = al.rsv_i = 0;
    BoardState<S> st;
    if (states) st = states[g]; else state_reset(st);
    const bool over = st.terminated != 0;
    bw.load_colors(st.bb[0], st.bb[1]);
    bw.analyze();
    const int root = tree_alloc(d, al, 1);
    int blk = root < 0 ? -1 : make_block(bw, st, arena, d, al, !over);
    if (!over) encode_bits<S, 0>(bw, st, d.rules, bits_s, d.obs_bits + (size_t)g * d.sc.R * d.obs_words);      // the game's slot 0
    if (bw.lane == 0) {
        NodeRec r;                                                    // Node_V(0), self_play.py:596 / :691
        r.prior = 0.0; r.w = 0.f; r.var = 0.f; r.n = 0; r.pending = 0; r.block = blk; r.action = 0xFFFF; r.flags = 0; r.term = 0;
        if (root >= 0) arena[root] = r;
        c.cur = 0; c.free_slot = al.free_slot; c.n_chunks = al.n_chunks; c.root = root < 0 ? 0 : root;
        if (al.n_chunks * d.sc.chunk_slots > c.hw_slot) c.hw_slot = al.n_chunks * d.sc.chunk_slots;
        c.need_eval = over ? 0 : 1; c.root_row = 0; c.error = blk < 0 ? 1 : 0;
        c.finished = over ? 1 : 0;
        if (c.error) { atomicAdd(&d.counters[CNT_ERRORS], 1); c.need_eval = 0; }
        d.ctl[g] = c;
        if (!over && !c.error) d.game_nslot[g] = 1;
    }
}

// root.expand(action_priors, value) with raw, un-normalised priors (self_play.py:599-605, :864-870)
template <int S>
__global__ __launch_bounds__(64) void k_expand_roots(EngineDev d) {
    constexpr int HS = TreeGeo<S>::HS;
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    if (!c->need_eval) return;
    NodeRec* arena = d.arena;
    const int root = c->root;
    const int blk = arena[root].block, row = d.game_off[g];           // the root was the game's only entry of the batch
    const int nchild = hdr_of<S>(arena, blk)->nchild;
    const float* pol = d.policy + (size_t)row * d.sc.A;
    const float val = d.value[row];
    for (int i = lane; i < nchild; i += 64) {
        NodeRec* r = &arena[blk + HS + i];
        r->prior = (double)pol[r->action];
        r->w = -val;
        r->flags = F_PRIOR32;
    }
    __syncthreads();
    if (lane == 0) { arena[root].flags |= F_OPEN; c->need_eval = 0; }
}

// Dirichlet root noise (self_play.py:90-95): prior <- prior*(1-0.25) + noise*0.25, float32*float -> float32, + float64
template <int S>
__global__ __launch_bounds__(64) void k_noise(EngineDev d, const double* noise) {
    constexpr int HS = TreeGeo<S>::HS;
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    if (c->finished || c->error) return;
    NodeRec* arena = d.arena;
    const int blk = arena[c->root].block;
    const int nchild = hdr_of<S>(arena, blk)->nchild;
    for (int i = lane; i < nchild; i += 64) {
        NodeRec* r = &arena[blk + HS + i];
        double p;
        if (r->flags & F_PRIOR32) p = (double)((float)r->prior * 0.75f); else p = r->prior * 0.75;
        r->prior = p + noise[(size_t)g * d.sc.A + i] * 0.25;
        r->flags &= (uint8_t)~F_PRIOR32;
    }
}

template <int S>
__global__ __launch_bounds__(64) void k_begin(EngineDev d, int sims) {
    const int g = blockIdx.x;
    if (lane_id() != 0) return;
    GameCtl* c = &d.ctl[g];
    c->n_target = d.arena[c->root].n + sims;                          // self_play.py:662-663
    c->searching = (c->finished || c->error) ? 0 : 1;
    c->active = c->searching;
}

// 4096 boards = 16 single-wave workgroups per CU = 4 waves per SIMD: the register budget must let all of them be resident at once
// (at 138 VGPRs only 12 fit and the kernel ran in two rounds).
template <int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(S == 9 ? 4 : 1))) void k_collect(EngineDev d) {
    using G = Geo<S>;
    constexpr int HS = TreeGeo<S>::HS, NPASS = TreeGeo<S>::NPASS;
    __shared__ WaveLds<S> lds;
    __shared__ uint32_t mt_scratch[1248];
    __shared__ int path_s[kMaxPath];                 // the path being walked (also written to HBM for k_absorb)
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    const SearchCfg& sc = d.sc;
    __shared__ uint32_t bits_s[(13 * G::P + 31) / 32];
    if (lane == 0) { c->n_paths = 0; d.game_nslot[g] = 0; d.game_act[g] = 0; }
    if (!c->searching || c->error) return;           // a game in error (arena overflow) is parked, never retried
    NodeRec* arena = d.arena;
    const int root = __builtin_amdgcn_readfirstlane(c->root);
    if (arena[root].n >= c->n_target) { if (lane == 0) c->active = 0; return; }
    if (lane == 0) d.game_act[g] = 1;
    int nslot = 0;                                   // evaluation-batch entries this game has written in this wave
    BoardWave<S> bw; bw.init(&lds);
    WaveRng rng; rng.key = d.rng[g].key; rng.pos = d.rng[g].pos; rng.scratch = mt_scratch; rng.draws = 0;
    // allocator state: wave-uniform, scalar registers; everything a new chunk needs is read HERE, with the kernel's other start-up
    // loads, so that the new-chunk path is one returning atomic + one ring read (the kernel's time is its slowest game's)
    TreeAlloc al; al.ids = chunk_list(d, g, __builtin_amdgcn_readfirstlane(c->cur)); al.n_chunks = __builtin_amdgcn_readfirstlane(c->n_chunks); al.ctl = c;
    al.free_slot = __builtin_amdgcn_readfirstlane(c->free_slot);
    { const unsigned long long v = d.pool->visible;
      al.visible = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v); }
    al.rsv = nullptr; al.rsv_n = al.rsv_i = 0;
    // a wave may need a new chunk when fewer than R largest blocks fit the current one: the pop goes out NOW and its two round trips
    // (ticket, ring entry) fly behind the first selection; a chunk that ends up unused waits in GameCtl::spare for the next wave
    {
        int sp = c->spare;
        const int room0 = (-al.free_slot) & (sc.chunk_slots - 1);
        if (TG_POOL_PREFETCH && sp < 0 && room0 < sc.R * (HS + G::A) && al.n_chunks < sc.max_chunks && lane == 0) sp = pool_pop(d, al.visible);
        al.spare = __builtin_amdgcn_readfirstlane(sp);
    }
    int* paths = d.path_nodes + (size_t)g * sc.R * sc.maxd;
    int npaths = 0, err = 0;
    int leafs[8], rows[8];
    unsigned long long sims = 0, depth_sum = 0, evals = 0, child_sum = 0;
    TG_ST_DECL;

    for (int attempt = 0; attempt < 2 * sc.R && npaths < sc.R && !err; ++attempt) {    // self_play.py:616
        int* path = paths + npaths * sc.maxd;
        int node = root, depth = 0;
        if (lane == 0) { path[0] = root; path_s[0] = root; }
        NodeRec cur = arena[root];
        int pblk = -1;                                                 // block of the leaf's parent (its header holds the parent's position)
        // ---- selection (self_play.py:623-627, :706-725) ----
        // A level is ONE dependent HBM round trip: the node's child count and, speculatively, its first 64 child records are
        // requested together (the count only masks them afterwards), and the chosen child's record is taken from the lane that
        // already holds it.  (It used to be three: count, children, chosen child.)
        while (cur.flags & F_OPEN) {
            const int blk = cur.block;
            NodeRec ch[NPASS];
            const int i0 = blk + HS + lane;
            if (i0 < sc.pool_slots) ch[0] = arena[i0];                // inside the pool whatever the child count is
            const int nchild = hdr_of<S>(arena, blk)->nchild;
            pblk = blk;
#pragma unroll
            for (int j = 1; j < NPASS; ++j) { const int i = j * 64 + lane; if (i < nchild) ch[j] = arena[blk + HS + i]; }
            child_sum += nchild;
            const double sq = sqrt((double)(cur.n + cur.pending));
            double scv[NPASS];
            double best = -INFINITY;
#pragma unroll
            for (int j = 0; j < NPASS; ++j) {
                const int i = j * 64 + lane;
                scv[j] = -INFINITY;
                if (i < nchild) scv[j] = puct_score(ch[j], sq, sc);
                best = scv[j] > best ? scv[j] : best;
            }
            best = wave_max(best);
            uint64_t tm[NPASS];
            int ntie = 0;
#pragma unroll
            for (int j = 0; j < NPASS; ++j) { tm[j] = ballot64(scv[j] == best); ntie += __popcll(tm[j]); }
            int k = (ntie > 1) ? rng.choice_index(ntie) : 0;
            int idx = 0;
#pragma unroll
            for (int j = 0; j < NPASS; ++j) {
                const int pc = __popcll(tm[j]);
                if (k >= 0 && k < pc) { idx = j * 64 + nth_set_bit(tm[j], k); k = -1; }
                else if (k >= 0) k -= pc;
            }
            node = blk + HS + idx;
            ++depth;
            if (depth >= sc.maxd || ntie == 0) { err |= 2; break; }
            if (lane == 0) { path[depth] = node; path_s[depth] = node; }
            const int src = __builtin_amdgcn_readfirstlane(idx & 63), pass = __builtin_amdgcn_readfirstlane(idx >> 6);
#pragma unroll
            for (int j = 0; j < NPASS; ++j)
                if (pass == j) cur = bcast_rec(ch[j], src);
        }
        if (depth == 0) err |= 8;                                      // root must be expanded before searching
        if (err) break;
        depth_sum += depth;
        TG_ST(0);
        // ---- leaf ----
        const int len = depth + 1;
        if (cur.term) {                                                // cached terminal result
            const float v = cur.term == 1 ? 1.f : -1.f;
            __syncthreads();
            for (int dd = lane; dd < len; dd += 64) backup_node(&arena[path_s[dd]], ((len - 1 - dd) & 1) ? -v : v);
            __syncthreads();
            ++sims;
            TG_ST(2);
            continue;
        }
        int row = -1;
        if (cur.flags & F_PSEUDO) {
            // same leaf reached twice inside one wave: the reference re-steps and re-expands to the identical result
            // and evaluates it again (self_play.py:629-646); we reuse the first path's row.
            for (int q = 0; q < npaths; ++q) if (leafs[q] == node) { row = rows[q]; break; }
        } else {
            BoardState<S> st = hdr_of<S>(arena, pblk)->st;            // the parent's position
            bool ok;
            const bool done = state_step(bw, st, cur.action, d.rules, /*check=*/false, &ok);   // self_play.py:629
            TG_ST(1);
            if (done) {                                                // self_play.py:638-642
                const float raw = tromp_taylor(bw, st, nullptr);
                const int winner = (raw - d.rules.komi > 0.f) ? kBlack : kWhite;              // environment.py:118-119
                const float v = (st.next_player == winner) ? 1.f : -1.f;
                if (lane == 0) arena[node].term = (v > 0.f) ? 1 : 2;
                __syncthreads();
                for (int dd = lane; dd < len; dd += 64) backup_node(&arena[path_s[dd]], ((len - 1 - dd) & 1) ? -v : v);
                __syncthreads();
                ++sims;
                TG_ST(2);
                continue;
            }
            bw.load_colors(st.bb[0], st.bb[1]);
            bw.analyze();
            TG_ST(3);
            const int blk = make_block(bw, st, arena, d, al, true);
            if (blk < 0) { err |= 1; break; }                         // the game's cap or the pool: parked, restarted by the host
            TG_ST(4);
            row = nslot++;                                             // the game's own next slot: no allocation, no atomic
            encode_bits<S, 1>(bw, st, d.rules, bits_s, d.obs_bits + ((size_t)g * sc.R + row) * d.obs_words);   // self_play.py:798
            TG_ST(5);
            if (lane == 0) {
                arena[node].block = blk;
                arena[node].flags = (uint8_t)(cur.flags | F_PSEUDO);   // (`cur` is this node's record: a store, not a read-modify-write)
            }
            ++evals;
        }
        leafs[npaths] = node; rows[npaths] = row;
        __syncthreads();
#if TG_PENDING_ATOMIC
        for (int dd = lane; dd < len; dd += 64) atomicAdd(&arena[path_s[dd]].pending, sc.wu);    // self_play.py:767-770 (no-return atomic)
#else
        for (int dd = lane; dd < len; dd += 64) arena[path_s[dd]].pending += sc.wu;
#endif
        if (lane == 0) { d.path_len[(size_t)g * sc.R + npaths] = len; d.path_row[(size_t)g * sc.R + npaths] = row; }
        __syncthreads();
        ++npaths;
        TG_ST(6);
    }
    TG_ST_FLUSH();
    if (lane == 0) {
        d.game_nslot[g] = nslot;
        c->n_paths = npaths; c->free_slot = al.free_slot; c->spare = al.spare; c->error |= err;
        c->sims += sims; c->depth_sum += depth_sum; c->evals += evals; c->tie_draws += rng.draws; c->child_sum += child_sum;
        d.rng[g].pos = rng.pos;
        if (err) { atomicAdd(&d.counters[CNT_ERRORS], 1); c->searching = 0; c->active = 0; }
    }
}

template <int S>
__global__ __launch_bounds__(64) void k_absorb(EngineDev d) {
    constexpr int HS = TreeGeo<S>::HS;
    __shared__ float pol_s[Geo<S>::A + 64];
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    const SearchCfg& sc = d.sc;
    const int npaths = c->n_paths;
    if (npaths == 0) return;
    NodeRec* arena = d.arena;
    const int* paths = d.path_nodes + (size_t)g * sc.R * sc.maxd;
    unsigned long long sims = 0;
    for (int q = 0; q < npaths; ++q) {                                 // self_play.py:651-654
        const int* path = paths + q * sc.maxd;
        const int len = d.path_len[(size_t)g * sc.R + q], row = d.game_off[g] + d.path_row[(size_t)g * sc.R + q];
        // (a no-return atomic: the read-modify-write it replaces was a dependent HBM round trip in front of everything below)
#if TG_PENDING_ATOMIC
        for (int dd = lane; dd < len; dd += 64) atomicSub(&arena[path[dd]].pending, sc.wu);     // self_play.py:772-774
#else
        for (int dd = lane; dd < len; dd += 64) arena[path[dd]].pending -= sc.wu;
#endif
        __syncthreads();
        const int leaf = path[len - 1];
        NodeRec lr = arena[leaf];
        if (lr.flags & F_OPEN) continue;                               // self_play.py:732-734: simulation dropped
        const int blk = lr.block;
        const int nchild = hdr_of<S>(arena, blk)->nchild;
        const float* pol = d.policy + (size_t)row * sc.A;
        const float val = d.value[row];
        for (int i = lane; i < nchild; i += 64) pol_s[i] = pol[arena[blk + HS + i].action];
        __syncthreads();
        float scale = pol_s[0];                                        // sum(policy[legal]): sequential float32
        for (int i = 1; i < nchild; ++i) scale = scale + pol_s[i];     // (0 + p0 == p0 exactly)
        if (scale > 0.f) {                                             // self_play.py:739-752
            for (int i = lane; i < nchild; i += 64) {
                NodeRec* r = &arena[blk + HS + i];
                r->prior = (double)(pol_s[i] / scale);
                r->w = -val;
                r->flags |= F_PRIOR32;
            }
        }
        if (lane == 0) arena[leaf].flags = (uint8_t)((lr.flags & ~F_PSEUDO) | F_OPEN);
        __syncthreads();
        for (int dd = lane; dd < len; dd += 64) backup_node(&arena[path[dd]], ((len - 1 - dd) & 1) ? -val : val);
        __syncthreads();
        ++sims;
    }
    if (lane == 0) { c->n_paths = 0; c->sims += sims; }
}

template <int S>
__global__ __launch_bounds__(64) void k_root_info(EngineDev d, int32_t* visits, int32_t* root_n, int32_t* player,
                                                  int32_t* step, int32_t* nchild_out, float* obs) {
    using G = Geo<S>;
    constexpr int HS = TreeGeo<S>::HS;
    __shared__ WaveLds<S> lds;
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    NodeRec* arena = d.arena;
    const int root = c->root;
    const int blk = arena[root].block;
    BlockHdr<S>* h = hdr_of<S>(arena, blk);
    const int nchild = h->nchild;
    if (visits) {
        for (int a = lane; a < G::A; a += 64) visits[(size_t)g * G::A + a] = 0;
        __syncthreads();
        for (int i = lane; i < nchild; i += 64) { NodeRec r = arena[blk + HS + i]; visits[(size_t)g * G::A + r.action] = r.n; }
    }
    if (lane == 0) {
        if (root_n) root_n[g] = arena[root].n;
        if (player) player[g] = h->st.next_player;
        if (step) step[g] = h->st.step_count;
        if (nchild_out) nchild_out[g] = (c->finished || c->error) ? 0 : nchild;
    }
    if (obs) {                                                         // env.encode(root.state), self_play.py:685
        BoardWave<S> bw; bw.init(&lds);
        BoardState<S> st = h->st;
        bw.load_colors(st.bb[0], st.bb[1]);
        bw.analyze();
        encode_planes(bw, st, d.rules, obs + (size_t)g * d.rules.encode_dim * G::P);
    }
}

// update_with_action (self_play.py:857-872) + tree compaction into the other half arena.
// Before the root moves on, the game's record gets this move's entry -- env.encode(root) bit-packed, the raw visit counts and
// the side to move: what self_play.py:917-926 appends to its three Python lists.
template <int S>
__global__ __launch_bounds__(64) void k_play(EngineDev d, const int32_t* actions, uint8_t* done_out, int32_t* moves_out) {
    using G = Geo<S>;
    constexpr int HS = TreeGeo<S>::HS, NPASS = TreeGeo<S>::NPASS;
    __shared__ WaveLds<S> lds;
    __shared__ uint32_t bits_s[(13 * G::P + 31) / 32];
    __shared__ int fill_s[ChunkCap<S>::N], cstart_s[ChunkCap<S>::N];   // chunk k of the new tree: first slot, and where its copied blocks end
    __shared__ int rsv_s[64];                          // chunk ids reserved for the new tree with one ticket draw
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    if (lane == 0) d.game_nslot[g] = 0;
    if (c->finished || c->error) { if (lane == 0) { done_out[g] = c->error ? 2 : 1; moves_out[g] = c->moves; } return; }
    const SearchCfg& sc = d.sc;
    NodeRec* const pool = d.arena;
    NodeRec* const old = pool; NodeRec* const nw = pool;               // (the old and the new tree live in the same pool)
    const int a = actions[g];
    const int rblk = old[c->root].block;
    const int nchild = hdr_of<S>(old, rblk)->nchild;
    int idx = -1;
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int i = j * 64 + lane;
        bool hit = i < nchild && old[rblk + HS + i].action == a;
        uint64_t m = ballot64(hit);
        if (m && idx < 0) idx = j * 64 + __ffsll((long long)m) - 1;
    }
    if (idx < 0) { if (lane == 0) { c->error |= 4; done_out[g] = 2; moves_out[g] = c->moves; atomicAdd(&d.counters[CNT_ERRORS], 1); } return; }
    BoardWave<S> bw; bw.init(&lds);
    BoardState<S> st = hdr_of<S>(old, rblk)->st;
    const int t = c->moves;
    if (d.hist_obs && t < d.hist_T) {
        const size_t e = (size_t)g * d.hist_T + t;
        int32_t* cnt = d.hist_cnt + e * G::A;
        for (int a2 = lane; a2 < G::A; a2 += 64) cnt[a2] = 0;
        __syncthreads();
        for (int i = lane; i < nchild; i += 64) { NodeRec r = old[rblk + HS + i]; cnt[r.action] = r.n; }
        bw.load_colors(st.bb[0], st.bb[1]);
        bw.analyze();
        encode_bits<S, 2>(bw, st, d.rules, bits_s, d.hist_obs + e * d.obs_words);
        if (lane == 0) d.hist_pl[e] = st.next_player;
    }
    bool ok;
    const bool done = state_step(bw, st, a, d.rules, /*check=*/false, &ok);      // self_play.py:859
    NodeRec child = old[rblk + HS + idx];
    // the new tree grows in fresh chunks, recorded in the game's OTHER chunk-id list
    TreeAlloc na; na.ids = chunk_list(d, g, c->cur ^ 1); na.n_chunks = 0; na.free_slot = 0; na.ctl = nullptr; na.visible = d.pool->visible; na.spare = -1;
    // ONE ticket draw for the chunks the new tree is expected to need (the kept sub-tree has at most one block per visit of the
    // chosen child; ~48 slots each): 4096 games drawing their chunks one by one serialise on the ring's head (the launch took 0.22 ms
    // instead of 0.09 with 64-simulation searches).  What is not used goes back with the old tree's chunks.
    {
        int want = 1;
        if (child.flags & F_OPEN) { want += (int)(((long long)child.n * 48 + sc.chunk_slots - 1) / sc.chunk_slots); }
        want = want > sc.max_chunks ? sc.max_chunks : want;
        want = want > 64 ? 64 : want;
        unsigned long long h0 = 0;
        if (lane == 0) h0 = atomicAdd(&d.pool->head, (unsigned long long)want);
        h0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(h0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)h0);
        const long long avail = (long long)(na.visible - h0);          // tickets at or beyond `visible` are failures (k_pool_publish takes them back)
        const int valid = avail <= 0 ? 0 : avail >= want ? want : (int)avail;
        if (valid < want && lane == 0) atomicAdd(&d.pool->exhausted, 1ull);
        if (lane < valid) rsv_s[lane] = d.ring[(h0 + lane) % (unsigned long long)sc.pool_chunks];
        __syncthreads();
        na.rsv = rsv_s; na.rsv_n = valid; na.rsv_i = 0;
    }
    const int nroot = tree_alloc(d, na, 1);
    if (nroot < 0) {                                                   // not one chunk left in the pool: the game is parked with its tree intact
        pool_push(d, na.rsv + na.rsv_i, na.rsv_n - na.rsv_i);
        if (lane == 0) { c->error |= 1; c->searching = 0; c->active = 0; done_out[g] = 2; moves_out[g] = c->moves; atomicAdd(&d.counters[CNT_ERRORS], 1); }
        return;
    }
    if (lane == 0) cstart_s[0] = nroot;
    int kept = 1;                                                      // slots of the new tree
    if (child.flags & F_OPEN) {
        // keep the subtree: breadth-first copy, block by block, fixing block pointers as we go.  The reference's tree lives
        // in unbounded Python memory; here the copy stops at `sc.keep_slots` (the game's cap minus the room one full search
        // can need) or when the pool has no chunk left, shallow blocks first: a node whose block no longer fits keeps its
        // statistics but becomes an unexpanded leaf again (it is re-evaluated on its next visit).  Never happens unless the
        // kept tree outgrows the cap (very peaked policies, many moves in a row) or the pool; counted in GameCtl::truncs.
        // Copied blocks sit back to back inside each chunk of the new list, chunk after chunk: fill_s[k] = where chunk k's
        // blocks end, which is what the scan below walks.
        int dropped = 0;
        auto copy_block = [&](int src) -> int {
            const int n = HS + hdr_of<S>(old, src)->nchild;
            if (kept + n > sc.keep_slots) { ++dropped; return -1; }
            const int before = na.n_chunks, fill_before = na.free_slot;
            const int at = tree_alloc(d, na, n);
            if (at < 0) { ++dropped; return -1; }
            if (na.n_chunks != before && lane == 0) { fill_s[before - 1] = fill_before; cstart_s[before] = at; }
            const uint4* s4 = reinterpret_cast<const uint4*>(old + src);
            uint4* d4 = reinterpret_cast<uint4*>(nw + at);
            for (int i = lane; i < 2 * n; i += 64) d4[i] = s4[i];
            kept += n;
            return at;
        };
        const int first = copy_block(child.block);
        child.block = first;
        if (first < 0) child.flags &= (uint8_t)~(F_OPEN | F_PSEUDO);
        __syncthreads();
        // scan cursor: chunk sc_k of the new list, slot `scan`; the root record occupies the first slot of chunk 0
        int sc_k = 0, scan = nroot + 1;
        while (first >= 0) {
            // end of the blocks of chunk sc_k: recorded when the allocator left it, else the allocator's bump pointer
            __syncthreads();
            const int fill = sc_k < na.n_chunks - 1 ? fill_s[sc_k] : na.free_slot;
            if (scan >= fill) {
                if (sc_k >= na.n_chunks - 1) break;                    // caught up with the allocator: every copied block was scanned
                ++sc_k;
                scan = cstart_s[sc_k];
                continue;
            }
            const int nc = hdr_of<S>(nw, scan)->nchild;
            for (int j0 = 0; j0 < nc; j0 += 64) {
                const int i = j0 + lane;
                int cb = -1;
                if (i < nc) cb = nw[scan + HS + i].block;
                uint64_t m = ballot64(cb >= 0);
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int src = __shfl(cb, b);
                    const int at = copy_block(src);
                    if (lane == b) {
                        nw[scan + HS + i].block = at;
                        if (at < 0) nw[scan + HS + i].flags &= (uint8_t)~(F_OPEN | F_PSEUDO);
                    }
                }
            }
            scan += HS + nc;
        }
        if (lane == 0) { nw[nroot] = child; c->truncs += dropped; }
    } else {
        // fresh root: it will be evaluated and expanded with raw priors (self_play.py:861-870).  The reference does
        // so even when the game just ended; that evaluation has no observable effect and is skipped here.
        if (!done) { bw.load_colors(st.bb[0], st.bb[1]); bw.analyze(); }
        const int blk = make_block(bw, st, nw, d, na, !done);
        if (blk < 0) {                                                 // pool exhausted: park the game (old tree intact), give the new chunks back
            pool_push(d, na.ids, na.n_chunks, na.rsv + na.rsv_i, na.rsv_n - na.rsv_i);
            if (lane == 0) { c->error |= 1; c->searching = 0; c->active = 0; done_out[g] = 2; moves_out[g] = c->moves; atomicAdd(&d.counters[CNT_ERRORS], 1); }
            return;
        }
        child.block = blk; child.term = 0; child.flags &= (uint8_t)~F_PSEUDO;
        if (lane == 0) nw[nroot] = child;
        if (!done) {
            encode_bits<S, 2>(bw, st, d.rules, bits_s, d.obs_bits + (size_t)g * sc.R * d.obs_words);  // the game's slot 0
            if (lane == 0) { c->need_eval = 1; c->root_row = 0; d.game_nslot[g] = 1; }
        }
    }
    __syncthreads();
    // the old tree's chunks, and what was reserved and not needed, go back to the pool (poppable after k_pool_publish)
    pool_push(d, chunk_list(d, g, c->cur), c->n_chunks, na.rsv + na.rsv_i, na.rsv_n - na.rsv_i);
    if (lane == 0) {
        c->cur ^= 1; c->free_slot = na.free_slot; c->n_chunks = na.n_chunks; c->root = nroot;
        if (na.n_chunks * sc.chunk_slots > c->hw_slot) c->hw_slot = na.n_chunks * sc.chunk_slots;
        c->finished = done ? 1 : 0; c->searching = 0; c->active = 0; c->moves = t + 1;
        done_out[g] = done ? 1 : 0; moves_out[g] = t + 1;
    }
}

// Finished games -> training positions, on the device (self_play.py:929-967 minus the 8-fold augmentation, which the replay
// sampler applies on the fly): one workgroup per finished game scores its final position (getScoreAndTerritory, getWinner)
// and writes, for each of its recorded moves, the bit-packed observation, the raw visit counts, z = +1 if the mover is the
// winner else -1 (:931-934) and the territory seen from the mover (:938-940), at the game's offset of a position-major batch
// -- exactly the arrays tg_replay_append stores.
template <int S>
__global__ __launch_bounds__(64) void k_harvest(EngineDev d, const int32_t* slot, const int32_t* off, uint32_t* o_obs,
                                                int32_t* o_cnt, float* o_z, int8_t* o_own, uint8_t* o_pl, int32_t* o_winner,
                                                float* o_score, int8_t* o_terr) {
    using G = Geo<S>;
    __shared__ WaveLds<S> lds;
    __shared__ int8_t terr_s[G::PPAD];
    const int f = blockIdx.x, lane = lane_id();
    const int g = slot[f];
    const size_t base = (size_t)off[f];
    GameCtl* c = &d.ctl[g];
    const int n = c->moves < d.hist_T ? c->moves : d.hist_T;
    NodeRec* arena = d.arena;
    BoardWave<S> bw; bw.init(&lds);
    BoardState<S> st = hdr_of<S>(arena, arena[c->root].block)->st;
    uint8_t owner[G::NW];
    const float raw = tromp_taylor(bw, st, owner);
    const float sc = raw - d.rules.komi;                               // go_env.cc:129
    const int winner = sc > 0.f ? kBlack : kWhite;                     // environment.py:118-119
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        const int8_t tv = owner[k] == 1 ? 1 : owner[k] == 2 ? -1 : 0;  // go_env.cc:141-146
        terr_s[bw.pt[k]] = tv;
        if (o_terr && bw.pt[k] < G::P) o_terr[(size_t)f * G::P + bw.pt[k]] = tv;
    }
    if (lane == 0) { if (o_winner) o_winner[f] = winner; if (o_score) o_score[f] = sc; }
    __syncthreads();
    const uint32_t* h_obs = d.hist_obs + (size_t)g * d.hist_T * d.obs_words;
    const int32_t* h_cnt = d.hist_cnt + (size_t)g * d.hist_T * G::A;
    const uint8_t* h_pl = d.hist_pl + (size_t)g * d.hist_T;
    for (int i = lane; i < n * d.obs_words; i += 64) o_obs[base * d.obs_words + i] = h_obs[i];
    for (int i = lane; i < n * G::A; i += 64) o_cnt[base * G::A + i] = h_cnt[i];
    for (int i = lane; i < n; i += 64) {
        const int pl = h_pl[i];
        o_z[base + i] = pl == winner ? 1.f : -1.f;
        if (o_pl) o_pl[base + i] = (uint8_t)pl;
    }
    for (int i = lane; i < n * G::P; i += 64) {
        const int tt = i / G::P, p = i - tt * G::P;
        o_own[base * G::P + i] = h_pl[tt] == kBlack ? terr_s[p] : (int8_t)-terr_s[p];
    }
}

template <int S>
__global__ __launch_bounds__(64) void k_root_states(EngineDev d, BoardState<S>* out) {
    const int g = blockIdx.x;
    if (lane_id() != 0) return;
    GameCtl* c = &d.ctl[g];
    NodeRec* arena = d.arena;
    out[g] = hdr_of<S>(arena, arena[c->root].block)->st;
}

template <int S>
__global__ __launch_bounds__(64) void k_final(EngineDev d, float* score, float* terr, int32_t* winner) {
    using G = Geo<S>;
    __shared__ WaveLds<S> lds;
    const int g = blockIdx.x, lane = lane_id();
    GameCtl* c = &d.ctl[g];
    NodeRec* arena = d.arena;
    BoardWave<S> bw; bw.init(&lds);
    BoardState<S> st = hdr_of<S>(arena, arena[c->root].block)->st;
    uint8_t owner[G::NW];
    const float raw = tromp_taylor(bw, st, owner);
    const float sc = raw - d.rules.komi;
    if (lane == 0) { if (score) score[g] = sc; if (winner) winner[g] = sc > 0.f ? kBlack : kWhite; }
    if (terr)
#pragma unroll
        for (int k = 0; k < G::NW; ++k)
            if (bw.pt[k] < G::P) terr[(size_t)g * G::P + bw.pt[k]] = owner[k] == 1 ? 1.f : owner[k] == 2 ? -1.f : 0.f;
}

// Float planes [rows][C][P] of the pending batch (what env.encode returns, self_play.py:798), for hosts that evaluate it themselves.
__global__ __launch_bounds__(256) void k_bits_to_planes(const uint32_t* __restrict__ bits, const int32_t* __restrict__ row_slot,
                                                        float* __restrict__ out, int rows, int CP, int W) {
    const size_t total = (size_t)rows * CP;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / CP), k = (int)(i % CP);
        out[i] = (float)((bits[(size_t)row_slot[r] * W + (k >> 5)] >> (k & 31)) & 1u);
    }
}

// Counts -> dense rows: one workgroup scans game_nslot over the games (exclusive prefix = game_off), lists the slots in row order
// (row_slot) and leaves the totals where the host reads them (CNT_ROWS, CNT_ACTIVE).  4096 games: ~5 us.
__global__ __launch_bounds__(1024) void k_compact(EngineDev d, int R, int count_active) {
    __shared__ int part[1024];
    __shared__ int act_s;
    const int tid = threadIdx.x, G = d.G;
    const int per = (G + 1023) / 1024, lo = tid * per, hi = lo + per < G ? lo + per : G;
    if (tid == 0) act_s = 0;
    int sum = 0, act = 0;
    for (int g = lo; g < hi; ++g) { sum += d.game_nslot[g]; if (count_active) act += d.game_act[g]; }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {                        // inclusive scan over the 1024 partial sums
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (act) atomicAdd(&act_s, act);
    int base = part[tid] - sum;
    for (int g = lo; g < hi; ++g) {
        const int n = d.game_nslot[g];
        d.game_off[g] = base;
        for (int k = 0; k < n; ++k) d.row_slot[base + k] = g * R + k;
        base += n;
    }
    __syncthreads();
    if (tid == 0) { d.counters[CNT_ROWS] = part[1023]; if (count_active) d.counters[CNT_ACTIVE] = act_s; }
}

// TG_TRACE_LAUNCH=1 in the environment: every tree-kernel launch is announced on stderr and waited for, so that a GPU fault is
// reported right behind the name of the kernel that caused it (diagnostic aid; off = no cost beyond one static read)
static const bool g_trace_launch = getenv("TG_TRACE_LAUNCH") && atoi(getenv("TG_TRACE_LAUNCH")) != 0;
static long g_trace_seq = 0;
#define TG_LAUNCH(ctx, kern, grid, ...)                                                                   \
    do {                                                                                                  \
        if (g_trace_launch) { fprintf(stderr, "[tg %ld] %s<%d> grid %d\n", ++g_trace_seq, #kern, (ctx)->S, (int)(grid)); fflush(stderr); } \
        if ((ctx)->S == 9) hipLaunchKernelGGL(kern<9>, dim3(grid), dim3(64), 0, (ctx)->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(kern<19>, dim3(grid), dim3(64), 0, (ctx)->stream, __VA_ARGS__);             \
        TG_HIP(ctx, hipGetLastError());                                                                   \
        if (g_trace_launch) TG_HIP(ctx, hipStreamSynchronize((ctx)->stream));                             \
    } while (0)

int read_counters(tg_ctx* ctx, int32_t* out) {
    Engine* e = ctx->eng;
    TG_HIP(ctx, hipMemcpyAsync(out, e->dev.counters, sizeof(int32_t) * CNT_N, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}
int zero_counter(tg_ctx* ctx, int which) {
    TG_HIP(ctx, hipMemsetAsync(ctx->eng->dev.counters + which, 0, sizeof(int32_t), ctx->stream));
    return TG_OK;
}

// The per-game MT19937 streams live in HBM (tie-break draws happen inside k_collect) and are mirrored to pinned host memory
// for the draws NumPy makes in float64 with libm (Dirichlet noise, the move's random_sample()).  Between the end of a search
// and the next tg_sp_begin_move the host copy is the authoritative one, so a move costs one round trip, not one per draw site.
int rng_to_host(tg_ctx* ctx) {
    Engine* e = ctx->eng;
    if (e->rng_on_host) return TG_OK;
    TG_HIP(ctx, hipMemcpyAsync(e->h_rng, e->dev.rng, sizeof(tg_mt19937) * e->G, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    e->rng_on_host = true;
    return TG_OK;
}
int rng_to_device(tg_ctx* ctx) {
    Engine* e = ctx->eng;
    if (!e->rng_on_host) return TG_OK;
    TG_HIP(ctx, hipMemcpyAsync(e->dev.rng, e->h_rng, sizeof(tg_mt19937) * e->G, hipMemcpyHostToDevice, ctx->stream));
    e->rng_on_host = false;
    return TG_OK;
}

// f(g) for g in [0, n) on a few host threads (independent games: per-game RNG streams, disjoint outputs)
template <class F> void parallel_games(int n, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)(hw ? (hw > 16 ? 16 : hw) : 1);
    if (n < 256 || nt <= 1) { for (int g = 0; g < n; ++g) f(g); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=]() { for (int g = (int)((long long)n * t / nt), e2 = (int)((long long)n * (t + 1) / nt); g < e2; ++g) f(g); });
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

// Default provision of the tree pool, slots per game on average: the per-game cap is (4*sims + 256) largest blocks, a game's tree
// averages well under half of it (DESIGN.md 3 "Pool sizing": measured fill), and the sub-tree k_play keeps comes out of the same
// slack before the old tree is returned.
long long tg_default_pool_slots(int board_size, int num_simulation) {
    const int A = board_size * board_size + 1, HS = board_size == 9 ? 2 : 4;
    return (2LL * num_simulation + 128) * (HS + A);
}

int tg_engine_create(tg_ctx* ctx) {
    const tg_config& cfg = ctx->cfg;
    if (cfg.parallel_readouts < 1 || cfg.parallel_readouts > 8) TG_FAIL(ctx, TG_ERR_ARG, "parallel_readouts must be 1..8");
    if (cfg.num_simulation < 1) TG_FAIL(ctx, TG_ERR_ARG, "num_simulation must be >= 1");
    Engine* e = new Engine();
    ctx->eng = e;
    const int G = cfg.n_games, R = cfg.parallel_readouts, A = ctx->A, P = ctx->P, C = cfg.encode_dim;
    const int HS = ctx->S == 9 ? TreeGeo<9>::HS : TreeGeo<19>::HS;
    e->G = G; e->R = R; e->rows_cap = G * R;
    SearchCfg& sc = e->dev.sc;
    sc.R = R; sc.wu = cfg.wu_loss; sc.c1 = cfg.c_puct1; sc.c2 = cfg.c_puct2;
    sc.c1f = (float)cfg.c_puct1; sc.c2f = (float)cfg.c_puct2; sc.A = A;
    sc.maxd = ((cfg.max_step + 2 + 63) / 64) * 64;
    if (sc.maxd > kMaxPath) TG_FAIL(ctx, TG_ERR_ARG, "max_step too large for the path buffer (at most 510)");
    // per-game cap: most slots one game's tree may hold (3x the simulations truncated kept sub-trees in 0.2 % of the game-moves of
    // full-length games); what bounds the SUM over the games is the pool below
    long long slots = cfg.arena_slots > 0 ? cfg.arena_slots : (4LL * cfg.num_simulation + 256) * (HS + A);
    if (slots < 4LL * (HS + A) || slots > 0x3fffffffLL) TG_FAIL(ctx, TG_ERR_ARG, "arena_slots out of range");
    sc.arena_slots = (int)slots;
    // room a full search can need: one block per evaluated leaf, at most num_simulation + R of them per move
    const long long headroom = ((long long)cfg.num_simulation + R + 1) * (HS + A);
    sc.keep_slots = (int)(slots - headroom > 1 + HS + A ? slots - headroom : 1 + HS + A);
    // chunk pool: chunks of 1024 slots (32 KB) at 9x9, 4096 (128 KB) at 19x19 -- a dozen largest blocks, so the tail a block
    // leaves unused when it does not fit the current chunk is a few per cent; a game owns at least one chunk
    sc.chunk_slots = ctx->S == 9 ? 1024 : 4096;
    sc.max_chunks = (int)((slots + sc.chunk_slots - 1) / sc.chunk_slots) + 1;
    if (sc.max_chunks > (ctx->S == 9 ? ChunkCap<9>::N : ChunkCap<19>::N))
        TG_FAIL(ctx, TG_ERR_ARG, "arena_slots too large for the chunk table (at most 511 chunks of 1024 slots per game at 9x9, 2047 of 4096 at 19x19)");
    // pool size: cfg.pool_slots slots per game ON AVERAGE (0: default_pool_slots below), at least two chunks per game and one
    // largest tree
    long long per_game = cfg.pool_slots > 0 ? cfg.pool_slots : tg_default_pool_slots(ctx->S, cfg.num_simulation);
    long long pool_chunks = ((long long)G * per_game + sc.chunk_slots - 1) / sc.chunk_slots;
    if (pool_chunks < 2LL * G + sc.max_chunks) pool_chunks = 2LL * G + sc.max_chunks;
    if (pool_chunks * sc.chunk_slots > 0x7ff00000LL)               // slot indices are int32 (and selection reads up to 64 slots past a block start)
        TG_FAIL(ctx, TG_ERR_ARG, "tree pool larger than 2^31 slots (64 GiB): lower pool_slots or n_games");
    sc.pool_chunks = (int)pool_chunks; sc.pool_slots = (int)(pool_chunks * sc.chunk_slots);
    e->dev.rules = ctx->rules;
    size_t arena_bytes = (size_t)sc.pool_slots * sizeof(NodeRec);
    TG_HIP(ctx, hipMalloc((void**)&e->dev.arena, arena_bytes));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.ring, sizeof(int32_t) * (size_t)sc.pool_chunks));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.pool, sizeof(PoolCtl)));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.chunk_ids, sizeof(int32_t) * (size_t)G * 2 * sc.max_chunks));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.ctl, sizeof(GameCtl) * G));
    TG_HIP(ctx, hipMemsetAsync(e->dev.ctl, 0, sizeof(GameCtl) * G, ctx->stream));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.rng, sizeof(tg_mt19937) * G));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.path_nodes, sizeof(int32_t) * (size_t)G * R * sc.maxd));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.path_len, sizeof(int32_t) * (size_t)G * R));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.path_row, sizeof(int32_t) * (size_t)G * R));
    e->dev.G = G;
    e->dev.obs_words = (C * P + 31) / 32;
    TG_HIP(ctx, hipMalloc((void**)&e->dev.obs_bits, sizeof(uint32_t) * (size_t)e->rows_cap * e->dev.obs_words));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.row_slot, sizeof(int32_t) * (size_t)e->rows_cap));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.game_nslot, sizeof(int32_t) * (size_t)G));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.game_off, sizeof(int32_t) * (size_t)G));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.game_act, (size_t)G));
    TG_HIP(ctx, hipMemsetAsync(e->dev.game_nslot, 0, sizeof(int32_t) * (size_t)G, ctx->stream));
    TG_HIP(ctx, hipMemsetAsync(e->dev.game_off, 0, sizeof(int32_t) * (size_t)G, ctx->stream));
    TG_HIP(ctx, hipMemsetAsync(e->dev.game_act, 0, (size_t)G, ctx->stream));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.policy, sizeof(float) * (size_t)e->rows_cap * A));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.value, sizeof(float) * (size_t)e->rows_cap));
    TG_HIP(ctx, hipMalloc((void**)&e->dev.counters, sizeof(int32_t) * CNT_N));
    TG_HIP(ctx, hipMemsetAsync(e->dev.counters, 0, sizeof(int32_t) * CNT_N, ctx->stream));
    TG_HIP(ctx, hipMalloc((void**)&e->d_noise, sizeof(double) * (size_t)G * A));
    TG_HIP(ctx, hipMalloc((void**)&e->d_i32, sizeof(int32_t) * (size_t)G * (A + 8)));
    TG_HIP(ctx, hipMalloc((void**)&e->d_f32, sizeof(float) * (size_t)G * (C * P + P + 8)));
    TG_HIP(ctx, hipMalloc((void**)&e->d_u8, (size_t)G * 4));
    TG_HIP(ctx, hipMalloc((void**)&e->d_fin, sizeof(int32_t) * 2 * (size_t)G));
    TG_HIP(ctx, hipHostMalloc((void**)&e->h_rng, sizeof(tg_mt19937) * (size_t)G, hipHostMallocDefault));
    memset(e->h_rng, 0, sizeof(tg_mt19937) * (size_t)G);
    e->h_moves.assign(G, 0);
    if (cfg.record_games) {
        // a game has at most max_step moves (step_count starts at 1 and the game ends when it exceeds max_step, go_env.cc:67)
        e->dev.hist_T = cfg.max_step > 0 ? cfg.max_step : 1;
        const size_t n = (size_t)G * e->dev.hist_T;
        TG_HIP(ctx, hipMalloc((void**)&e->dev.hist_obs, sizeof(uint32_t) * n * e->dev.obs_words));
        TG_HIP(ctx, hipMalloc((void**)&e->dev.hist_cnt, sizeof(int32_t) * n * A));
        TG_HIP(ctx, hipMalloc((void**)&e->dev.hist_pl, n));
    }
    e->h_noise.resize((size_t)G * A);
    e->h_nchild.resize(G);
    e->arena_bytes = arena_bytes;
    hipLaunchKernelGGL(k_pool_init, dim3(64), dim3(256), 0, ctx->stream, e->dev, 1);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (g_trace_launch) {                                             // where every engine buffer lives: a fault address can be placed
        const EngineDev& v = e->dev;
        auto pr = [](const char* n, const void* p, size_t b) { fprintf(stderr, "[tg buf] %-10s %p .. %p (%zu B)\n", n, p, (const char*)p + b, b); };
        pr("arena", v.arena, arena_bytes); pr("ctl", v.ctl, sizeof(GameCtl) * G); pr("rng", v.rng, sizeof(tg_mt19937) * G);
        pr("path_nodes", v.path_nodes, sizeof(int32_t) * (size_t)G * R * sc.maxd); pr("path_len", v.path_len, sizeof(int32_t) * (size_t)G * R);
        pr("path_row", v.path_row, sizeof(int32_t) * (size_t)G * R); pr("obs_bits", v.obs_bits, sizeof(uint32_t) * (size_t)e->rows_cap * v.obs_words);
        pr("row_slot", v.row_slot, sizeof(int32_t) * (size_t)e->rows_cap); pr("game_nslot", v.game_nslot, sizeof(int32_t) * (size_t)G);
        pr("game_off", v.game_off, sizeof(int32_t) * (size_t)G); pr("game_act", v.game_act, (size_t)G);
        pr("policy", v.policy, sizeof(float) * (size_t)e->rows_cap * A); pr("value", v.value, sizeof(float) * (size_t)e->rows_cap);
        pr("counters", v.counters, sizeof(int32_t) * CNT_N);
        if (v.hist_obs) { const size_t n = (size_t)G * v.hist_T; pr("hist_obs", v.hist_obs, sizeof(uint32_t) * n * v.obs_words);
                          pr("hist_cnt", v.hist_cnt, sizeof(int32_t) * n * A); pr("hist_pl", v.hist_pl, n); }
        fflush(stderr);
    }
    return TG_OK;
}

void tg_engine_destroy(tg_ctx* ctx) {
    Engine* e = ctx->eng;
    if (!e) return;
    void* ptrs[] = {e->dev.arena, e->dev.ring, e->dev.pool, e->dev.chunk_ids, e->dev.ctl, e->dev.rng, e->dev.path_nodes, e->dev.path_len, e->dev.path_row, e->dev.row_slot,
                    e->dev.obs_bits, e->dev.game_nslot, e->dev.game_off, e->dev.game_act, e->dev.policy, e->dev.value, e->dev.counters, e->d_noise, e->d_i32, e->d_f32, e->d_u8,
                    e->d_fin, e->dev.hist_obs, e->dev.hist_cnt, e->dev.hist_pl};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (e->h_rng) (void)hipHostFree(e->h_rng);
    e->obs_f32.release(); e->hv_obs.release(); e->hv_cnt.release(); e->hv_z.release(); e->hv_own.release(); e->hv_pl.release(); e->hv_game.release();
    for (hipEvent_t ev : e->tev) (void)hipEventDestroy(ev);
    tg_net_destroy(ctx);
    delete e;
    ctx->eng = nullptr;
}

#define NEED_ENGINE(ctx) do { if (!(ctx) || !(ctx)->eng || (ctx)->eng->G <= 0) return TG_ERR_ARG; TG_HIP(ctx, hipSetDevice((ctx)->cfg.device)); } while (0)

int tg_sp_reset(tg_ctx* ctx, const uint32_t* seeds, const uint8_t* mask) {
    NEED_ENGINE(ctx);
    if (!seeds) return TG_ERR_ARG;
    Engine* e = ctx->eng;
    const int G = e->G;
    if (e->batch_kind != BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_reset: an evaluation batch is pending");
    if (mask && !e->all_reset) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_reset: the first reset must cover every game (mask = NULL)");
    // RNG streams are seeded in the host mirror (np.random.seed(seed) per game); they travel with the next tg_sp_begin_move
    if (mask) { int rc = rng_to_host(ctx); if (rc) return rc; }
    else e->rng_on_host = true;                                       // every stream is overwritten: nothing to fetch
    parallel_games(G, [&](int g) { if (!mask || mask[g]) tg_host_mt_seed(&e->h_rng[g], seeds[g]); });
    for (int g = 0; g < G; ++g) if (!mask || mask[g]) e->h_moves[g] = 0;
    e->fin_slot.clear(); e->fin_off.clear(); e->fin_positions = 0;   // unharvested records of restarted slots are gone
    uint8_t* d_mask = nullptr;
    if (mask) {
        TG_HIP(ctx, hipMemcpyAsync(e->d_u8, mask, G, hipMemcpyHostToDevice, ctx->stream));
        d_mask = e->d_u8;
    }
    if (g_trace_launch) { fprintf(stderr, "[tg %ld] k_reset<%d> grid %d (masked %d)\n", ++g_trace_seq, ctx->S, G, mask ? 1 : 0); fflush(stderr); }
    // every game at once: the ring is rebuilt with all chunks free and no game returns anything; a masked reset returns the
    // restarted games' chunks (poppable after the publish below)
    if (!mask) hipLaunchKernelGGL(k_pool_init, dim3(64), dim3(256), 0, ctx->stream, e->dev, 0);
    else {
        hipLaunchKernelGGL(k_release, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask);
        hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(1), 0, ctx->stream, e->dev);
    }
    TG_HIP(ctx, hipGetLastError());
    if (ctx->S == 9) hipLaunchKernelGGL(k_reset<9>, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask, (const BoardState<9>*)nullptr);
    else hipLaunchKernelGGL(k_reset<19>, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask, (const BoardState<19>*)nullptr);
    TG_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, ctx->stream, e->dev, e->R, 0);
    TG_HIP(ctx, hipGetLastError());
    e->batch_kind = BATCH_ROOTS; e->batch_ready = false;
    if (!mask) e->all_reset = true;
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_reset_from(tg_ctx* ctx, const void* states, const uint8_t* mask) {
    NEED_ENGINE(ctx);
    if (!states) return TG_ERR_ARG;
    Engine* e = ctx->eng;
    const int G = e->G;
    if (e->batch_kind != BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_reset_from: an evaluation batch is pending");
    if (!e->all_reset) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_reset_from: call tg_sp_reset first (it seeds the RNG streams)");
    if (ctx->env_in.reserve((size_t)G * ctx->state_bytes)) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
    TG_HIP(ctx, hipMemcpyAsync(ctx->env_in.p, states, (size_t)G * ctx->state_bytes, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d_mask = nullptr;
    std::vector<uint8_t> all;
    if (mask) { TG_HIP(ctx, hipMemcpyAsync(e->d_u8, mask, G, hipMemcpyHostToDevice, ctx->stream)); d_mask = e->d_u8; }
    for (int g = 0; g < G; ++g) if (!mask || mask[g]) e->h_moves[g] = 0;
    e->fin_slot.clear(); e->fin_off.clear(); e->fin_positions = 0;
    // (unmasked slots are parked by k_reset and keep the one chunk they own until their next reset)
    hipLaunchKernelGGL(k_release, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask);
    hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(1), 0, ctx->stream, e->dev);
    TG_HIP(ctx, hipGetLastError());
    if (ctx->S == 9) hipLaunchKernelGGL(k_reset<9>, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask, (const BoardState<9>*)ctx->env_in.p);
    else hipLaunchKernelGGL(k_reset<19>, dim3(G), dim3(64), 0, ctx->stream, e->dev, (const uint8_t*)d_mask, (const BoardState<19>*)ctx->env_in.p);
    TG_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, ctx->stream, e->dev, e->R, 0);
    TG_HIP(ctx, hipGetLastError());
    int32_t cnt[CNT_N];
    int rc = read_counters(ctx, cnt);
    if (rc) return rc;
    e->batch_kind = BATCH_ROOTS; e->batch_ready = cnt[CNT_ROWS] == 0; e->last_rows = cnt[CNT_ROWS];
    return TG_OK;
}

int tg_sp_root_states(tg_ctx* ctx, void* states) {
    NEED_ENGINE(ctx);
    if (!states) return TG_ERR_ARG;
    Engine* e = ctx->eng;
    const size_t bytes = (size_t)e->G * ctx->state_bytes;
    if (ctx->env_out.reserve(bytes)) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
    if (ctx->S == 9) hipLaunchKernelGGL(k_root_states<9>, dim3(e->G), dim3(64), 0, ctx->stream, e->dev, (BoardState<9>*)ctx->env_out.p);
    else hipLaunchKernelGGL(k_root_states<19>, dim3(e->G), dim3(64), 0, ctx->stream, e->dev, (BoardState<19>*)ctx->env_out.p);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipMemcpyAsync(states, ctx->env_out.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_batch_rows(tg_ctx* ctx, int32_t* n_rows) {
    NEED_ENGINE(ctx);
    int32_t cnt[CNT_N];
    int rc = read_counters(ctx, cnt);
    if (rc) return rc;
    *n_rows = ctx->eng->batch_kind == BATCH_NONE ? 0 : cnt[CNT_ROWS];
    return TG_OK;
}

int tg_sp_batch_obs(tg_ctx* ctx, float* obs, int32_t n_rows) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (n_rows < 0 || n_rows > e->rows_cap) return TG_ERR_ARG;
    if (n_rows == 0) return TG_OK;
    const size_t per = (size_t)ctx->cfg.encode_dim * ctx->P;
    if (e->obs_f32.reserve(sizeof(float) * per * n_rows)) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
    int grid = (int)((per * n_rows + 255) / 256); if (grid > 65535) grid = 65535;
    hipLaunchKernelGGL(k_bits_to_planes, dim3(grid), dim3(256), 0, ctx->stream, (const uint32_t*)e->dev.obs_bits,
                       (const int32_t*)e->dev.row_slot, (float*)e->obs_f32.p, (int)n_rows, (int)per, e->dev.obs_words);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipMemcpyAsync(obs, e->obs_f32.p, sizeof(float) * per * n_rows, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_set_eval(tg_ctx* ctx, const float* policy, const float* value, int32_t n_rows) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (e->batch_kind == BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_set_eval: no batch pending");
    if (n_rows < 0 || n_rows > e->rows_cap) return TG_ERR_ARG;
    TG_HIP(ctx, hipMemcpyAsync(e->dev.policy, policy, sizeof(float) * (size_t)n_rows * ctx->A, hipMemcpyHostToDevice, ctx->stream));
    TG_HIP(ctx, hipMemcpyAsync(e->dev.value, value, sizeof(float) * (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    e->batch_ready = true;
    return TG_OK;
}

int tg_sp_eval(tg_ctx* ctx) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (e->batch_kind == BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_eval: no batch pending");
    int32_t cnt[CNT_N];
    int rc = read_counters(ctx, cnt);
    if (rc) return rc;
    rc = tg_net_forward(ctx, cnt[CNT_ROWS]);
    if (rc) return rc;
    e->batch_ready = true;
    return TG_OK;
}

int tg_sp_expand_roots(tg_ctx* ctx) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (e->batch_kind != BATCH_ROOTS) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_expand_roots: no root batch pending");
    if (!e->batch_ready) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_expand_roots: batch not evaluated");
    TG_LAUNCH(ctx, k_expand_roots, e->G, e->dev);
    e->batch_kind = BATCH_NONE; e->batch_ready = false;
    return TG_OK;
}

int tg_sp_begin_move(tg_ctx* ctx, int selfplay, int num_simulation) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    const int G = e->G, A = ctx->A;
    if (e->batch_kind != BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_begin_move: an evaluation batch is pending");
    if (num_simulation <= 0) num_simulation = ctx->cfg.num_simulation;
    // weights change between moves only (the reference swaps them at game start, self_play.py:913; with G games in lock step the
    // move boundary is the closest point every game shares): one search = one weight set
    { int rc = tg_net_adopt_ready(ctx); if (rc) return rc; }
    if (selfplay) {                                                    // root.dirichlet_prior(), self_play.py:659-660
        int32_t* d_nchild = e->d_i32;
        TG_LAUNCH(ctx, k_root_info, G, e->dev, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                  d_nchild, (float*)nullptr);
        TG_HIP(ctx, hipMemcpyAsync(e->h_nchild.data(), d_nchild, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        { int rc = rng_to_host(ctx); if (rc) return rc; }
        parallel_games(G, [&](int g) {
            if (e->h_nchild[g] > 0) tg_host_mt_dirichlet(&e->h_rng[g], 0.03, e->h_nchild[g], &e->h_noise[(size_t)g * A]);
        });
        TG_HIP(ctx, hipMemcpyAsync(e->d_noise, e->h_noise.data(), sizeof(double) * (size_t)G * A, hipMemcpyHostToDevice, ctx->stream));
        TG_LAUNCH(ctx, k_noise, G, e->dev, (const double*)e->d_noise);
    }
    { int rc = rng_to_device(ctx); if (rc) return rc; }
    TG_LAUNCH(ctx, k_begin, G, e->dev, num_simulation);
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_collect(tg_ctx* ctx, int32_t* n_active, int32_t* n_rows) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (e->batch_kind != BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_collect: an evaluation batch is pending");
    { int rc = rng_to_device(ctx); if (rc) return rc; }
    const bool timed = e->tprof && e->tev_used + 4 <= e->tev.size();
    if (timed) { TG_HIP(ctx, hipEventRecord(e->tev[e->tev_used], ctx->stream)); }
    TG_LAUNCH(ctx, k_collect, e->G, e->dev);
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, ctx->stream, e->dev, e->R, 1);
    TG_HIP(ctx, hipGetLastError());
    if (timed) { TG_HIP(ctx, hipEventRecord(e->tev[e->tev_used + 1], ctx->stream)); e->tev_kind[e->tev_used / 2] = 0; e->tev_used += 2; e->tree_waves++; }
    int32_t cnt[CNT_N];
    int rc = read_counters(ctx, cnt);
    if (rc) return rc;
    if (n_active) *n_active = cnt[CNT_ACTIVE];
    if (n_rows) *n_rows = cnt[CNT_ROWS];
    e->batch_kind = BATCH_LEAVES; e->batch_ready = cnt[CNT_ROWS] == 0;
    e->last_rows = cnt[CNT_ROWS];
    e->n_errors = cnt[CNT_ERRORS];          // games parked in error (tg_sp_game_errors); the others are unaffected
    return TG_OK;
}

int tg_sp_absorb(tg_ctx* ctx) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (e->batch_kind != BATCH_LEAVES) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_absorb: no leaf batch pending");
    if (!e->batch_ready) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_absorb: batch not evaluated");
    const bool timed = e->tprof && e->tev_used + 2 <= e->tev.size();
    if (timed) { TG_HIP(ctx, hipEventRecord(e->tev[e->tev_used], ctx->stream)); }
    TG_LAUNCH(ctx, k_absorb, e->G, e->dev);
    if (timed) { TG_HIP(ctx, hipEventRecord(e->tev[e->tev_used + 1], ctx->stream)); e->tev_kind[e->tev_used / 2] = 1; e->tev_used += 2; }
    e->batch_kind = BATCH_NONE; e->batch_ready = false;
    return TG_OK;
}

int tg_sp_search(tg_ctx* ctx, int32_t* n_waves) {
    NEED_ENGINE(ctx);
    int waves = 0;
    // every wave completes at least one simulation per active game, so num_simulation waves always suffice
    const int max_waves = 2 * ctx->cfg.num_simulation + 1024;
    for (;;) {
        if (waves > max_waves) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_search: no progress (wave limit exceeded)");
        int32_t active = 0, rows = 0;
        int rc = tg_sp_collect(ctx, &active, &rows);
        if (rc) return rc;
        if (rows > 0) {
            if (g_trace_launch) { fprintf(stderr, "[tg %ld] network forward, %d rows\n", ++g_trace_seq, rows); fflush(stderr); }
            rc = tg_net_forward(ctx, rows); if (rc) return rc; ctx->eng->batch_ready = true;
            if (g_trace_launch) TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        rc = tg_sp_absorb(ctx);
        if (rc) return rc;
        if (active == 0) break;
        ++waves;
    }
    if (n_waves) *n_waves = waves;
    return TG_OK;
}

int tg_sp_root_info(tg_ctx* ctx, int32_t* visits, int32_t* root_n, int32_t* player, int32_t* step, float* obs) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    const int G = e->G, A = ctx->A, P = ctx->P, C = ctx->cfg.encode_dim;
    int32_t* d_vis = e->d_i32; int32_t* d_n = d_vis + (size_t)G * A; int32_t* d_pl = d_n + G; int32_t* d_st = d_pl + G;
    float* d_obs = e->d_f32;
    TG_LAUNCH(ctx, k_root_info, G, e->dev, visits ? d_vis : nullptr, root_n ? d_n : nullptr, player ? d_pl : nullptr,
              step ? d_st : nullptr, (int32_t*)nullptr, obs ? d_obs : nullptr);
    if (visits) TG_HIP(ctx, hipMemcpyAsync(visits, d_vis, sizeof(int32_t) * (size_t)G * A, hipMemcpyDeviceToHost, ctx->stream));
    if (root_n) TG_HIP(ctx, hipMemcpyAsync(root_n, d_n, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
    if (player) TG_HIP(ctx, hipMemcpyAsync(player, d_pl, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
    if (step) TG_HIP(ctx, hipMemcpyAsync(step, d_st, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
    if (obs) TG_HIP(ctx, hipMemcpyAsync(obs, d_obs, sizeof(float) * (size_t)G * C * P, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_draw_uniform(tg_ctx* ctx, double* u, const uint8_t* mask) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    const int G = e->G;
    { int rc = rng_to_host(ctx); if (rc) return rc; }
    for (int g = 0; g < G; ++g) u[g] = (!mask || mask[g]) ? tg_host_mt_random_sample(&e->h_rng[g]) : 0.0;
    return TG_OK;
}

int tg_sp_rng_state(tg_ctx* ctx, int game, tg_mt19937* out) {
    NEED_ENGINE(ctx);
    if (game < 0 || game >= ctx->eng->G || !out) return TG_ERR_ARG;
    if (ctx->eng->rng_on_host) { *out = ctx->eng->h_rng[game]; return TG_OK; }
    TG_HIP(ctx, hipMemcpyAsync(out, ctx->eng->dev.rng + game, sizeof(tg_mt19937), hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_rng_get(tg_ctx* ctx, tg_mt19937* out) {
    NEED_ENGINE(ctx);
    if (!out) return TG_ERR_ARG;
    int rc = rng_to_host(ctx);
    if (rc) return rc;
    memcpy(out, ctx->eng->h_rng, sizeof(tg_mt19937) * (size_t)ctx->eng->G);
    return TG_OK;
}

int tg_sp_rng_set(tg_ctx* ctx, const tg_mt19937* in, const uint8_t* mask) {
    NEED_ENGINE(ctx);
    if (!in) return TG_ERR_ARG;
    Engine* e = ctx->eng;
    int rc = rng_to_host(ctx);                                         // the host mirror becomes (stays) the authoritative copy
    if (rc) return rc;
    for (int g = 0; g < e->G; ++g) if (!mask || mask[g]) e->h_rng[g] = in[g];
    return TG_OK;
}

int tg_sp_play(tg_ctx* ctx, const int32_t* actions, uint8_t* done) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    const int G = e->G;
    if (!actions || !done) return TG_ERR_ARG;
    if (e->batch_kind != BATCH_NONE) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_play: an evaluation batch is pending");
    int32_t* d_act = e->d_i32;
    TG_HIP(ctx, hipMemcpyAsync(d_act, actions, sizeof(int32_t) * G, hipMemcpyHostToDevice, ctx->stream));
    int32_t* d_moves = e->d_i32 + G;
    std::vector<int32_t> prev = e->h_moves;
    TG_LAUNCH(ctx, k_play, G, e->dev, (const int32_t*)d_act, e->d_u8, d_moves);
    hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(1), 0, ctx->stream, e->dev);      // the old trees' chunks are poppable from here on
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, ctx->stream, e->dev, e->R, 0);
    TG_HIP(ctx, hipGetLastError());
    TG_HIP(ctx, hipMemcpyAsync(done, e->d_u8, G, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipMemcpyAsync(e->h_moves.data(), d_moves, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
    int32_t cnt[CNT_N];
    int rc = read_counters(ctx, cnt);
    if (rc) return rc;
    e->batch_kind = BATCH_ROOTS; e->batch_ready = cnt[CNT_ROWS] == 0; e->last_rows = cnt[CNT_ROWS];
    e->n_errors = cnt[CNT_ERRORS];
    // games this call finished (a slot that was already over does not move and is not listed again), ascending slot order
    e->fin_slot.clear(); e->fin_off.clear(); e->fin_positions = 0;
    for (int g = 0; g < G; ++g)
        if (done[g] == 1 && e->h_moves[g] == prev[g] + 1) {
            const int n = e->h_moves[g] < e->dev.hist_T ? e->h_moves[g] : e->dev.hist_T;
            e->fin_slot.push_back(g); e->fin_off.push_back(e->fin_positions); e->fin_positions += n;
        }
    return TG_OK;
}

int tg_sp_finished(tg_ctx* ctx, int32_t* n_games, int32_t* n_positions) {
    NEED_ENGINE(ctx);
    if (n_games) *n_games = (int32_t)ctx->eng->fin_slot.size();
    if (n_positions) *n_positions = ctx->eng->fin_positions;
    return TG_OK;
}

int tg_sp_harvest(tg_ctx* ctx, uint32_t* obs_bits, int32_t* counts, float* z, int8_t* own, uint8_t* player, int device_out,
                  int32_t* slot, int32_t* n_moves, int32_t* winner, float* score, int8_t* terr) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (!e->dev.hist_obs) TG_FAIL(ctx, TG_ERR_STATE, "tg_sp_harvest: the context was created with record_games = 0");
    const int nf = (int)e->fin_slot.size(), np = e->fin_positions, P = ctx->P, A = ctx->A, W = e->dev.obs_words;
    if (nf == 0) return TG_OK;
    if (!obs_bits || !counts || !z || !own) return TG_ERR_ARG;
    TG_HIP(ctx, hipMemcpyAsync(e->d_fin, e->fin_slot.data(), sizeof(int32_t) * nf, hipMemcpyHostToDevice, ctx->stream));
    TG_HIP(ctx, hipMemcpyAsync(e->d_fin + e->G, e->fin_off.data(), sizeof(int32_t) * nf, hipMemcpyHostToDevice, ctx->stream));
    uint32_t* d_obs = obs_bits; int32_t* d_cnt = counts; float* d_z = z; int8_t* d_own = own; uint8_t* d_pl = player;
    if (!device_out) {
        if (e->hv_obs.reserve(sizeof(uint32_t) * (size_t)np * W) || e->hv_cnt.reserve(sizeof(int32_t) * (size_t)np * A) ||
            e->hv_z.reserve(sizeof(float) * (size_t)np) || e->hv_own.reserve((size_t)np * P) || e->hv_pl.reserve((size_t)np))
            TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
        d_obs = (uint32_t*)e->hv_obs.p; d_cnt = (int32_t*)e->hv_cnt.p; d_z = (float*)e->hv_z.p; d_own = (int8_t*)e->hv_own.p;
        d_pl = player ? (uint8_t*)e->hv_pl.p : nullptr;
    }
    // per-game results: winner i32[nf] | score f32[nf] | terr i8[nf][P]
    if (e->hv_game.reserve((size_t)nf * (8 + P))) TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed");
    int32_t* d_w = (int32_t*)e->hv_game.p; float* d_s = (float*)(d_w + nf); int8_t* d_t = (int8_t*)(d_s + nf);
    TG_LAUNCH(ctx, k_harvest, nf, e->dev, (const int32_t*)e->d_fin, (const int32_t*)(e->d_fin + e->G), d_obs, d_cnt, d_z, d_own,
              d_pl, d_w, d_s, d_t);
    if (!device_out) {
        TG_HIP(ctx, hipMemcpyAsync(obs_bits, d_obs, sizeof(uint32_t) * (size_t)np * W, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(counts, d_cnt, sizeof(int32_t) * (size_t)np * A, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(z, d_z, sizeof(float) * (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(own, d_own, (size_t)np * P, hipMemcpyDeviceToHost, ctx->stream));
        if (player) TG_HIP(ctx, hipMemcpyAsync(player, d_pl, (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (winner) TG_HIP(ctx, hipMemcpyAsync(winner, d_w, sizeof(int32_t) * nf, hipMemcpyDeviceToHost, ctx->stream));
    if (score) TG_HIP(ctx, hipMemcpyAsync(score, d_s, sizeof(float) * nf, hipMemcpyDeviceToHost, ctx->stream));
    if (terr) TG_HIP(ctx, hipMemcpyAsync(terr, d_t, (size_t)nf * P, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));                    // device outputs are complete on return as well
    for (int i = 0; i < nf; ++i) {
        if (slot) slot[i] = e->fin_slot[i];
        if (n_moves) n_moves[i] = (i + 1 < nf ? e->fin_off[i + 1] : np) - e->fin_off[i];
    }
    return TG_OK;
}

int tg_sp_game_errors(tg_ctx* ctx, int32_t* n_errors, int32_t* err) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    if (n_errors) *n_errors = e->n_errors;
    if (err) {
        std::vector<GameCtl> h(e->G);
        TG_HIP(ctx, hipMemcpyAsync(h.data(), e->dev.ctl, sizeof(GameCtl) * e->G, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        int n = 0;
        for (int g = 0; g < e->G; ++g) { err[g] = h[g].error; n += h[g].error != 0; }
        e->n_errors = n;
        if (n_errors) *n_errors = n;
    }
    return TG_OK;
}

int tg_sp_tree_truncations(tg_ctx* ctx, uint64_t* blocks) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    std::vector<GameCtl> h(e->G);
    TG_HIP(ctx, hipMemcpyAsync(h.data(), e->dev.ctl, sizeof(GameCtl) * e->G, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t n = 0;
    for (const GameCtl& c : h) n += c.truncs;
    if (blocks) *blocks = n;
    return TG_OK;
}

int tg_sp_final(tg_ctx* ctx, float* score, float* terr, int32_t* winner) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    const int G = e->G, P = ctx->P;
    float* d_score = e->d_f32; float* d_terr = d_score + G; int32_t* d_w = e->d_i32;
    TG_LAUNCH(ctx, k_final, G, e->dev, d_score, terr ? d_terr : nullptr, d_w);
    if (score) TG_HIP(ctx, hipMemcpyAsync(score, d_score, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream));
    if (terr) TG_HIP(ctx, hipMemcpyAsync(terr, d_terr, sizeof(float) * (size_t)G * P, hipMemcpyDeviceToHost, ctx->stream));
    if (winner) TG_HIP(ctx, hipMemcpyAsync(winner, d_w, sizeof(int32_t) * G, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return TG_OK;
}

int tg_sp_stats(tg_ctx* ctx, uint64_t* sims, uint64_t* evals, uint64_t* depth_sum, uint64_t* tie_draws, int32_t* errors,
                int32_t* max_slots) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    std::vector<GameCtl> h(e->G);
    TG_HIP(ctx, hipMemcpyAsync(h.data(), e->dev.ctl, sizeof(GameCtl) * e->G, hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t s = 0, ev = 0, ds = 0, td = 0; int32_t er = 0, ms = 0;
    for (const GameCtl& c : h) {
        s += c.sims; ev += c.evals; ds += c.depth_sum; td += c.tie_draws;
        er += c.error ? 1 : 0; ms = c.hw_slot > ms ? c.hw_slot : ms;   // a true high-water (chunks x chunk size), not the current fill
    }
    if (sims) *sims = s; if (evals) *evals = ev; if (depth_sum) *depth_sum = ds; if (tie_draws) *tie_draws = td;
    if (errors) *errors = er; if (max_slots) *max_slots = ms;
    return TG_OK;
}

int tg_sp_pool_stats(tg_ctx* ctx, uint64_t* pool_slots, uint64_t* high_water_slots, uint64_t* in_use_slots, uint64_t* exhausted) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    PoolCtl pc;
    TG_HIP(ctx, hipMemcpyAsync(&pc, e->dev.pool, sizeof(pc), hipMemcpyDeviceToHost, ctx->stream));
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const SearchCfg& sc = e->dev.sc;
    if (pc.head > pc.visible) pc.head = pc.visible;                    // (failed tickets on a dry pool, not yet taken back by a publish)
    const unsigned long long fr = pc.tail - pc.head;                   // free right now (everything pushed so far included)
    unsigned long long mn = pc.min_free; const unsigned long long vis_free = pc.visible - pc.head; if (vis_free < mn) mn = vis_free;
    if (pool_slots) *pool_slots = (uint64_t)sc.pool_slots;
    if (high_water_slots) *high_water_slots = (uint64_t)((unsigned long long)sc.pool_chunks - mn) * sc.chunk_slots;
    if (in_use_slots) *in_use_slots = (uint64_t)((unsigned long long)sc.pool_chunks - fr) * sc.chunk_slots;
    if (exhausted) *exhausted = pc.exhausted;
    return TG_OK;
}

// HIP-event time of the tree stage (k_collect / k_absorb launches since tg_prof_enable(ctx, 1, n)) and the PUCT fan-out counter
int tg_prof_read_tree(tg_ctx* ctx, double* collect_ms, double* absorb_ms, int64_t* waves, uint64_t* children_scored) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 1 < e->tev_used; i += 2) {
        float t = 0; TG_HIP(ctx, hipEventElapsedTime(&t, e->tev[i], e->tev[i + 1]));
        (e->tev_kind[i / 2] ? e->absorb_ms : e->collect_ms) += t;
    }
    e->tev_used = 0;
    if (collect_ms) *collect_ms = e->collect_ms;
    if (absorb_ms) *absorb_ms = e->absorb_ms;
    if (waves) *waves = e->tree_waves;
    if (children_scored) {
        std::vector<GameCtl> h(e->G);
        TG_HIP(ctx, hipMemcpy(h.data(), e->dev.ctl, sizeof(GameCtl) * e->G, hipMemcpyDeviceToHost));
        uint64_t cs = 0; for (const GameCtl& c : h) cs += c.child_sum;
        *children_scored = cs;
    }
    return TG_OK;
}

#ifdef TG_TREE_STAMP
// diagnostic builds: summed s_memtime cycles per k_collect phase [0..7], whole-kernel cycles [8], game-waves [9]; reset = 1 clears
int tg_debug_tree_stamps(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 16) != hipSuccess) return TG_ERR_HIP;
    if (out && hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_stamp2), sizeof(unsigned long long) * 8) != hipSuccess) return TG_ERR_HIP;   // out: 24 words
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return TG_ERR_HIP;
                 if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp2), z, sizeof(unsigned long long) * 8) != hipSuccess) return TG_ERR_HIP; }
    return TG_OK;
}
#endif

int tg_prof_enable_tree(tg_ctx* ctx, int on, int max_waves) {
    NEED_ENGINE(ctx);
    Engine* e = ctx->eng;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (on) {
        const size_t want = 4 * (size_t)(max_waves > 0 ? max_waves : 4096);
        while (e->tev.size() < want) { hipEvent_t ev; TG_HIP(ctx, hipEventCreate(&ev)); e->tev.push_back(ev); }
        e->tev_kind.assign(e->tev.size() / 2 + 1, 0);
        e->tev_used = 0; e->collect_ms = e->absorb_ms = 0; e->tree_waves = 0;
    }
    e->tprof = on != 0;
    return TG_OK;
}

}  // extern "C"
