// tree_dev.h -- device data structures + wave primitives of the batched WP_MCTS (self_play.py:51-95, :575-875).
//
// Layout in HBM (round 4).  ONE pool of 32-byte slots serves every game of the context, cut into chunks of SearchCfg::chunk_slots
// slots; a game's tree lives in the chunks listed in its chunk-id list and grows by taking a chunk off the pool's free ring when
// its current one is full (tree_alloc, engine.hip).  The pool is provisioned for the POPULATION of games, not for the worst game
// times the number of games (rounds 1-3: two fixed half-arenas per game, 41 GB at BASELINE configs[1] for trees that fill a third
// of it on average).  Node and block references are GLOBAL slot indices into the pool, so the hot kernels never translate
// anything; GameCtl::root is the root record.  When the root advances (k_play) the kept sub-tree is copied breadth-first into
// fresh chunks and the old tree's chunks go back to the ring, so nothing is ever freed piecemeal.  An
// expanded (or pseudo-expanded) node owns a *block*: HS header slots holding its board state, followed by one
// NodeRec per legal action in ascending action order (pass last) -- the iteration order of the reference's
// `children` dict (self_play.py:603, :635) -- contiguous inside one chunk.  Selection therefore reads one contiguous, fully
// coalesced run of nchild*32 B per tree level; a wave's 64 lanes each take one child.
#pragma once
#include "board_dev.h"
#include "../../include/transgo_hip.h"

namespace tg {

constexpr uint8_t F_OPEN = 1;      // real_expanded (self_play.py:62)
constexpr uint8_t F_PRIOR32 = 2;   // prior holds a float32 value (NumPy scalar-kind ladder, SURVEY.md Note N)
constexpr uint8_t F_PSEUDO = 4;    // pseudo-expanded in the current wave, evaluation pending (self_play.py:634-636)

struct alignas(16) NodeRec {       // Node_V, self_play.py:51-63
    double prior;                  // prior
    float w;                       // value_sum (float32, accumulated leaf->root)
    float var;                     // value_var (value_mean is recomputable: w/(n+1) of the previous backup)
    int32_t n;                     // total_visit_count
    int32_t pending;               // ons
    int32_t block;                 // slot of this node's block, -1 if none
    uint16_t action;               // action leading here
    uint8_t flags;
    uint8_t term;                  // cached terminal result: 0 none, 1 => +1, 2 => -1 (self_play.py:638-642)
};
static_assert(sizeof(NodeRec) == 32, "tree slot is 32 B");

template <int S> struct alignas(32) BlockHdr {
    BoardState<S> st;
    int32_t nchild;
    int32_t pad[3];
};
template <int S> struct TreeGeo {
    static constexpr int HS = (int)(sizeof(BlockHdr<S>) / 32);
    static constexpr int NPASS = (Geo<S>::A + 63) / 64;
};
static_assert(sizeof(BlockHdr<9>) == 64 && sizeof(BlockHdr<19>) == 128, "header slots");

// The free ring of the chunk pool.  head / tail count pops / pushes since creation (64-bit: never wrap); entry k of the ring is
// ring[k % pool_chunks].  Chunks pushed by a kernel become poppable when k_pool_publish has run behind it (`visible`): a pop never
// reads a ring entry that another workgroup of the same launch may still be writing.
struct PoolCtl {
    unsigned long long head, tail, visible;
    unsigned long long min_free;     // fewest free chunks seen at a publish point (pool high-water = pool_chunks - min_free)
    unsigned long long exhausted;    // pops that found the ring empty (the game is then parked in error 1 / its kept sub-tree truncated)
};

struct GameCtl {
    int32_t cur;          // which of the game's two chunk-id lists holds the live tree (0/1; the other one is filled by k_play)
    int32_t free_slot;    // next unused slot (global index) in the chunk the live tree is growing in (on a chunk boundary = no room)
    int32_t n_chunks;     // chunks the live tree owns
    int32_t root;         // global slot of the root record
    int32_t spare;        // a chunk taken off the pool ahead of need by k_collect and not used yet (-1: none); it stays with the game
    int32_t n_target;     // root visit target of the current move (self_play.py:662-663)
    int32_t active;       // still below target in this move
    int32_t n_paths;      // paths collected by the wave in flight
    int32_t need_eval;    // root awaits evaluation + expansion (self_play.py:599-605, :861-870)
    int32_t root_row;     // its row in the evaluation batch
    int32_t finished;     // root state is terminal
    int32_t error;        // sticky: 1 arena overflow, 2 depth overflow, 4 bad action
    int32_t searching;    // begin_move issued
    int32_t moves;        // moves played in this game = entries of its device-side record (self_play.py:917-926)
    int32_t hw_slot;      // most slots (chunks x chunk_slots) a tree of this game has held (cumulative like the counters below)
    unsigned long long sims;        // completed backups (terminal ones included)
    unsigned long long evals;       // leaves sent to the evaluator
    unsigned long long depth_sum;   // sum of selection depths
    unsigned long long tie_draws;   // RNG words consumed by tie breaks
    unsigned long long child_sum;   // children scored by PUCT, summed over selection levels (mean fan-out = child_sum / depth_sum)
    unsigned long long truncs;      // sub-tree blocks dropped at re-rooting because the kept tree outgrew the arena (k_play)
};

struct SearchCfg {
    int R;              // parallel_readouts
    int wu;             // wu_loss
    double c1, c2;      // c_puct1, c_puct2
    float c1f, c2f;     // their float32 roundings (weak Python scalars next to float32 operands)
    int arena_slots;    // most slots ONE game's tree may hold (per-game cap; the pool is what bounds the sum)
    int keep_slots;     // most slots a re-rooted tree may keep (arena_slots minus the room of one full search)
    int chunk_slots;    // slots per chunk (>= the largest block)
    int max_chunks;     // entries of a game's chunk-id list = chunks its tree may own
    int pool_chunks;    // chunks in the pool
    int pool_slots;     // pool_chunks * chunk_slots (< 2^31: slot indices are int32)
    int maxd;           // path capacity
    int A;
};

// ---- wave helpers -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) { double o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ int nth_set_bit(uint64_t m, int k) {      // k-th (0-based) set bit of m
    for (int i = 0; i < k; ++i) m &= m - 1;
    return __ffsll((long long)m) - 1;
}

// MT19937 on the device, NumPy legacy semantics (see rng_host.cpp).  `key` lives in HBM, pos in a register.
struct WaveRng {
    uint32_t* key;
    int pos;
    uint32_t* scratch;      // 1248 words of LDS
    unsigned long long draws;
    __device__ __forceinline__ void twist() {
        const int lane = lane_id();
        uint32_t* o = scratch; uint32_t* nw = scratch + 624;
        __syncthreads();
        for (int i = lane; i < 624; i += 64) o[i] = key[i];
        __syncthreads();
        auto tw = [](uint32_t a, uint32_t b) { uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
                                               return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); };
        for (int i = lane; i < 227; i += 64) nw[i] = o[i + 397] ^ tw(o[i], o[i + 1]);
        __syncthreads();
        for (int i = 227 + lane; i < 454; i += 64) nw[i] = nw[i - 227] ^ tw(o[i], o[i + 1]);
        __syncthreads();
        for (int i = 454 + lane; i < 623; i += 64) nw[i] = nw[i - 227] ^ tw(o[i], o[i + 1]);
        __syncthreads();
        if (lane == 0) nw[623] = nw[396] ^ tw(o[623], nw[0]);
        __syncthreads();
        for (int i = lane; i < 624; i += 64) key[i] = nw[i];
        __syncthreads();
        pos = 0;
    }
    __device__ __forceinline__ uint32_t next32() {
        if (pos >= 624) twist();
        uint32_t y = key[pos++];
        ++draws;
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    }
    // RandomState.choice(list of k): masked rejection on 32-bit words, no draw when k == 1 (self_play.py:709-713)
    __device__ __forceinline__ int choice_index(int k) {
        if (k <= 1) return 0;
        uint32_t rng = (uint32_t)(k - 1), mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        do { v = next32() & mask; } while (v > rng);
        return (int)v;
    }
};

// PUCT + variance bonus of one child, self_play.py:716-725, in the exact scalar kinds of Note N.
__device__ __forceinline__ double puct_score(const NodeRec& c, double sqrt_parent, const SearchCfg& sc) {
    const double denom = (double)(c.n + c.pending + 1);
    double u;
    if (c.flags & F_PRIOR32) { float t = sc.c1f * (float)c.prior; u = (double)t * sqrt_parent / denom; }
    else u = sc.c1 * c.prior * sqrt_parent / denom;
    double s;
    if (c.n == 0) s = sc.c2;                                   // var is still the Python float 0.: all float64
    else {
        float v = c.var; v = v < 0.f ? 0.f : (v > 3.f ? 3.f : v);
        s = (double)(sc.c2f * sqrtf(1.0f + v));
    }
    const float q = -(c.w / (float)(c.n + 1));
    return u + s + (double)q;
}

// backpropagate + value_mean_var for the node at path position d of `len` (self_play.py:758-764, :84-88): nodes of a
// path are distinct, so the leaf->root loop is data-parallel once the alternating sign is known.
__device__ __forceinline__ void backup_node(NodeRec* rec, float v) {
    const float w0 = rec->w; const int n0 = rec->n;
    const float t = (n0 == 0) ? 0.f : w0 / (float)(n0 + 1);
    const float w1 = w0 + v;
    const float mean = w1 / (float)(n0 + 2);
    rec->var = rec->var + (v - t) * (v - mean);
    rec->w = w1;
    rec->n = n0 + 1;
}

}  // namespace tg
