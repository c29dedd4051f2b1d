// board_dev.h -- wave-cooperative Go board engine for gfx950 (one 64-lane wavefront = one board).
//
// Replaces, behaviour-for-behaviour, the reference rules engine /root/reference/GoEnv/cpp_src (board.cc, board_feature.cc,
// go_env.cc; cited per function).  The mechanism is different by design: the reference keeps incremental liberty
// counts on linked lists of stones (board.cc:217-428, AoS, 1188 B per state); here a state is two bitboards plus 16 B
// of scalars (48 B at 9x9, 112 B at 19x19), lanes own points (point p = slot*64 + lane), groups are found by min-label
// propagation with pointer jumping through LDS and liberties are counted with LDS atomics (at 9x9: by growing each point's
// group as an 81-bit board in registers, label_groups()), so every per-point answer (legal, suicide, liberty class, eye,
// alive) is a lane-local read of LDS tables.  New bitboards come straight out of 64-bit ballots.
//
// A workgroup is exactly one wavefront (64 threads): __syncthreads() is then a single-wave s_barrier that orders the
// LDS traffic between lanes at negligible cost, and games never wait for each other's flood-fill trip counts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

constexpr int kEmpty = 0, kBlack = 1, kWhite = 2, kWall = 3;
constexpr int kPass = -1, kResign = -2, kInvalid = -3;     // go_comm.h:24-26
constexpr uint32_t kNone = 0xFFFFu;

struct RulesCfg {
    int max_step;     // go_env.cc:11
    float komi;       // go_env.cc:12
    int encode_dim;   // 9 / 10 / 13 (go_env.cc:96-115)
};

template <int S> struct Geo {
    static constexpr int P = S * S;
    static constexpr int A = P + 1;
    static constexpr int NW = (P + 63) / 64;     // bitboard words == points per lane
    static constexpr int PPAD = NW * 64;
};

// Game state: value type, caller-visible as an opaque blob through the C ABI (replaces GoState, go_env.h:15-18).
template <int S> struct alignas(16) BoardState {
    uint64_t bb[2][Geo<S>::NW];   // [0] black, [1] white; bit p of the row-major point index
    int16_t last_move1;           // board.h:47
    int16_t last_move2;           // board.h:48
    int16_t ko_location;          // board.h:51
    int16_t ko_age;               // board.h:53
    uint16_t step_count;          // board.h:46
    uint8_t ko_color;             // board.h:52
    uint8_t next_player;          // board.h:45
    uint8_t terminated;           // go_env.h:17
    uint8_t pad[3];
};
static_assert(sizeof(BoardState<9>) == 48, "9x9 state is 48 B");
static_assert(sizeof(BoardState<19>) == 112, "19x19 state is 112 B");

// LDS scratch of one wave.
template <int S> struct WaveLds {
    uint16_t col[Geo<S>::PPAD + 64];   // colour per point
    uint16_t lab[Geo<S>::PPAD + 64];   // group label per point (kNone for unlabelled)
    uint32_t cnt[Geo<S>::PPAD + 64];   // per label: liberties (or touch mask when scoring)
    uint32_t aux[Geo<S>::PPAD + 64];   // per label: good-eye count; per point: eye flag in bit 16
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

// 9x9 only: an 81-bit board as two words, and its 4-neighbourhood dilation (bit p = point p, row-major).  A shift by one moves
// stones across row ends, which the column masks take out again.
struct B81 { uint64_t lo, hi; };
__device__ __forceinline__ B81 b81_dilate(B81 g) {
    constexpr uint64_t C0_LO = 0x8040201008040201ull, C0_HI = 0x100ull;             // x == 0: bits 0, 9, ..., 63 | 72
    constexpr uint64_t C8_LO = 0x4020100804020100ull, C8_HI = 0x10080ull;           // x == 8: bits 8, 17, ..., 62 | 71, 80
    B81 r;
    r.lo = g.lo | (((g.lo >> 1) | (g.hi << 63)) & ~C8_LO) | ((g.lo << 1) & ~C0_LO) | ((g.lo >> 9) | (g.hi << 55)) | (g.lo << 9);
    r.hi = g.hi | ((g.hi >> 1) & ~C8_HI) | (((g.hi << 1) | (g.lo >> 63)) & ~C0_HI) | (g.hi >> 9) | ((g.hi << 9) | (g.lo >> 55));
    return r;
}
template <int S> struct BoardRegs {};
template <> struct BoardRegs<9> {
    uint64_t bq[2][2];      // the bitboards load_colors() was given (wave-uniform)
    uint16_t lab_r[2];      // label_groups(): label of the own points' groups (kNone if unlabelled)
    uint16_t lib_r[2];      //                 and their liberty counts
};

// Diagnostic builds (-DTG_TREE_STAMP): s_memtime sub-phase stamps inside the board code, accumulated in the BoardWave and flushed
// by k_collect together with its own phases (scripts/stamp_tree.py)
#ifdef TG_TREE_STAMP
#define TG_BW_ST_BEGIN(bw) (bw).st2_t = __builtin_amdgcn_s_memtime()
#define TG_BW_ST(bw, i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (bw).st2[i] += n_ - (bw).st2_t; (bw).st2_t = n_; } while (0)
#else
#define TG_BW_ST_BEGIN(bw)
#define TG_BW_ST(bw, i)
#endif

// Everything a lane knows about its NW points.
template <int S> struct BoardWave : BoardRegs<S> {
    using G = Geo<S>;
    static constexpr int NW = G::NW;
    WaveLds<S>* L;
    int lane;
    int pt[NW];         // point index (may be >= P: invalid)
    uint8_t nbv[NW];    // bit d set: neighbour d on board (d: 0 L, 1 U, 2 R, 3 D = go_comm.h:44-45); bits 4-7: diagonals
    uint8_t col[NW];    // colour of own point (kWall if invalid)
#ifdef TG_TREE_STAMP
    unsigned long long st2[8], st2_t;
#endif
    uint8_t su_valid;   // 0, or 1 + the player su_bits was computed for (reset by analyze())
    uint8_t su_bits;    // bit k: isSuicideMove(player) at own slot k (only meaningful for empty points)

    __device__ __forceinline__ void init(WaveLds<S>* lds) {
        L = lds;
        lane = lane_id();
        su_valid = 0; su_bits = 0;
#ifdef TG_TREE_STAMP
        for (int i = 0; i < 8; ++i) st2[i] = 0;
        st2_t = 0;
#endif
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int p = k * 64 + lane;
            pt[k] = p;
            int x = p % S, y = p / S;
            uint8_t v = 0;
            if (p < G::P) {
                if (x > 0) v |= 1;
                if (y > 0) v |= 2;
                if (x < S - 1) v |= 4;
                if (y < S - 1) v |= 8;
                // diagonals in the order of go_comm.h:49-50: (-1,-1) (-1,+1) (+1,+1) (+1,-1)
                if (x > 0 && y > 0) v |= 16;
                if (x > 0 && y < S - 1) v |= 32;
                if (x < S - 1 && y < S - 1) v |= 64;
                if (x < S - 1 && y > 0) v |= 128;
            }
            nbv[k] = v;
        }
    }
    __device__ __forceinline__ static int nb_off(int d) { return d == 0 ? -1 : d == 1 ? -S : d == 2 ? 1 : S; }
    __device__ __forceinline__ static int dg_off(int d) { return d == 0 ? -S - 1 : d == 1 ? S - 1 : d == 2 ? S + 1 : -S + 1; }

    // colours from bitboards -> registers + LDS
    __device__ __forceinline__ void load_colors(const uint64_t* bbB, const uint64_t* bbW) {
        if constexpr (S == 9) {
#pragma unroll
            for (int k = 0; k < NW; ++k) { this->bq[0][k] = bbB[k]; this->bq[1][k] = bbW[k]; }
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            int c = kWall;
            if (pt[k] < G::P) c = ((bbB[k] >> lane) & 1) ? kBlack : ((bbW[k] >> lane) & 1) ? kWhite : kEmpty;
            col[k] = (uint8_t)c;
            L->col[pt[k]] = (uint16_t)c;
        }
        __syncthreads();
    }
    // after editing col[] in registers
    __device__ __forceinline__ void publish_colors() {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NW; ++k) L->col[pt[k]] = col[k];
        __syncthreads();
    }
    __device__ __forceinline__ void to_bitboards(uint64_t* bbB, uint64_t* bbW) const {
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            bbB[k] = ballot64(col[k] == kBlack);
            bbW[k] = ballot64(col[k] == kWhite);
        }
    }

    // Connected components by min-label propagation + pointer jumping.  with_empty: empty regions are labelled too
    // (needed for Tromp-Taylor scoring only).  Result in L->lab.
    __device__ __forceinline__ void label_groups(bool with_empty) {
        if constexpr (S == 9) {
            // Barrier-free: every lane grows the group of each of its own points as an 81-bit board in registers (dilate, mask with
            // the stones of that colour -- or the empty points -- until no lane's board changes), then label = lowest point of the
            // group and liberties = empty points its dilation touches.  Same labels and counts as the propagation below.
            constexpr uint64_t ALL_HI = 0x1FFFFull;
            const B81 blk{this->bq[0][0], this->bq[0][1]}, wht{this->bq[1][0], this->bq[1][1]};
            const B81 emp{~(blk.lo | wht.lo), ~(blk.hi | wht.hi) & ALL_HI};
            B81 g[NW], own[NW];
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const bool on = (col[k] == kBlack || col[k] == kWhite || (with_empty && col[k] == kEmpty));
                own[k] = col[k] == kBlack ? blk : col[k] == kWhite ? wht : emp;
                g[k].lo = (on && k == 0) ? 1ull << lane : 0ull;
                g[k].hi = (on && k == 1) ? 1ull << lane : 0ull;
            }
            for (;;) {
                bool ch = false;
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    B81 n = b81_dilate(g[k]);
                    n.lo &= own[k].lo; n.hi &= own[k].hi;
                    ch |= (n.lo != g[k].lo) | (n.hi != g[k].hi);
                    g[k] = n;
                }
                if (!__any(ch)) break;                                   // a board only grows and has 81 bits: at most 81 trips
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const bool on = (g[k].lo | g[k].hi) != 0;
                const int lb = g[k].lo ? __ffsll((long long)g[k].lo) - 1 : 63 + __ffsll((long long)g[k].hi);
                const B81 d = b81_dilate(g[k]);
                this->lab_r[k] = on ? (uint16_t)lb : (uint16_t)kNone;
                this->lib_r[k] = (uint16_t)(__popcll(d.lo & emp.lo) + __popcll(d.hi & emp.hi));
                L->lab[pt[k]] = this->lab_r[k];
            }
            __syncthreads();
            return;
        }
        uint32_t lab[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            bool on = (col[k] == kBlack || col[k] == kWhite || (with_empty && col[k] == kEmpty));
            lab[k] = on ? (uint32_t)pt[k] : kNone;
            L->lab[pt[k]] = (uint16_t)lab[k];
        }
        __syncthreads();
        for (int it = 0; it < G::P + 2; ++it) {       // bounded: converges in far fewer rounds
            bool ch = false;
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                if (lab[k] == kNone) continue;
                uint32_t m = lab[k];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (!(nbv[k] >> d & 1)) continue;
                    int q = pt[k] + nb_off(d);
                    if (L->col[q] == col[k]) { uint32_t t = L->lab[q]; m = t < m ? t : m; }
                }
                ch |= (m != lab[k]);
                lab[k] = m;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NW; ++k) L->lab[pt[k]] = (uint16_t)lab[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NW; ++k)
                if (lab[k] != kNone) { uint32_t t = L->lab[lab[k]]; lab[k] = t < lab[k] ? t : lab[k]; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NW; ++k) L->lab[pt[k]] = (uint16_t)lab[k];
            __syncthreads();
            if (!__any(ch)) break;
        }
    }

    // Liberties per group label into L->cnt (equivalent observable of Block::liberties, board.h:23).
    __device__ __forceinline__ void count_liberties() {
        su_valid = 0;
        if constexpr (S == 9) {
            // label_groups(false) left every group's count with ALL of its stones' lanes, so at 9x9 the table is indexed by POINT:
            // cnt[p] = liberties of the group at p -- one LDS read per lookup (lib_at) instead of label then count
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                L->cnt[pt[k]] = (uint32_t)this->lib_r[k];
                L->aux[pt[k]] = 0;
            }
            __syncthreads();
            return;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) { L->cnt[pt[k]] = 0; L->aux[pt[k]] = 0; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if (col[k] != kEmpty) continue;
            uint32_t seen[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                seen[d] = kNone;
                if (!(nbv[k] >> d & 1)) continue;
                int q = pt[k] + nb_off(d);
                uint32_t l = L->lab[q];
                if (L->col[q] == kEmpty) l = kNone;
                bool dup = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < d && seen[e] == l) dup = true;
                seen[d] = l;
                if (l != kNone && !dup) atomicAdd(&L->cnt[l], 1u);
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ void analyze() { label_groups(false); count_liberties(); }

    __device__ __forceinline__ int lib_at(int q) const { return S == 9 ? (int)L->cnt[q] : (int)L->cnt[L->lab[q]]; }

    // board.cc:130-158 isSuicideMove for `player` at own slot k (point must be empty).
    __device__ __forceinline__ bool suicide(int k, int player) const {
        bool s = true;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (!(nbv[k] >> d & 1)) continue;
            int q = pt[k] + nb_off(d);
            int c = L->col[q];
            if (c == kEmpty) { s = false; continue; }
            int l = lib_at(q);
            if (c == player) { if (l > 1) s = false; }
            else if (l == 1) s = false;
        }
        return s;
    }
    // isSuicideMove for `player` at every own empty point, computed once per analysed position: legality (make_block) and the
    // ko/suicide feature plane (encode_mask) ask for the same player's answers (board.cc:467-489, board_feature.cc:69-89)
    __device__ __forceinline__ void ensure_suicide(int player) {
        if (su_valid == 1 + player) return;                              // wave-uniform
        uint8_t b = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (pt[k] < G::P && col[k] == kEmpty && suicide(k, player)) b |= (uint8_t)(1u << k);
        su_bits = b; su_valid = (uint8_t)(1 + player);
    }
    __device__ __forceinline__ bool suicide_cached(int k) const { return (su_bits >> k) & 1; }
    // board.cc:432-464 TryPlay for board points, on an analysed position.
    template <class St> __device__ __forceinline__ bool legal(int k, const St& st, int player) const {
        if (col[k] != kEmpty) return false;
        if (st.ko_location == pt[k] && st.ko_age == 0 && st.ko_color == player) return false;   // board.cc:198-200
        return !suicide(k, player);
    }
    template <class St> __device__ __forceinline__ bool legal_cached(int k, const St& st, int player) const {   // after ensure_suicide(player)
        if (col[k] != kEmpty) return false;
        if (st.ko_location == pt[k] && st.ko_age == 0 && st.ko_color == player) return false;
        return !suicide_cached(k);
    }
    // board.cc:665-714 isTrueEye
    // The own point as a value the optimiser cannot see through: without it hipcc hoists one LDS address per (slot, neighbour offset,
    // table) out of the read-out loop of k_collect -- dozens of registers live across the whole kernel at a 128-register budget,
    // spilled to scratch and reloaded ~65 times per read-out, each reload waiting behind the block's HBM stores (one in-order counter)
    __device__ __forceinline__ int own_point(int k) const { int p = pt[k]; asm volatile("" : "+v"(p)); return p; }
    __device__ __forceinline__ bool true_eye(int k, int player) const {
        if (col[k] != kEmpty) return false;
        bool eye = true;
        int nwall = 0, nopp = 0;
        const int pk = own_point(k);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (nbv[k] >> d & 1) { if (L->col[pk + nb_off(d)] != player) eye = false; }
            if (nbv[k] >> (4 + d) & 1) { if (L->col[pk + dg_off(d)] == (3 - player)) ++nopp; }
            else ++nwall;
        }
        bool fake = (nwall > 0 && nopp >= 1) || (nwall == 0 && nopp >= 2);
        return eye && !fake;
    }

    // board.cc:731-817 GivenBlockLives for every group of colour `c` at once.  Afterwards bit 0 of aux[label] >> 17
    // ... see alive_at().  Requires analyze().  Uses aux: per point bit 16 = true eye of colour c; low 16 bits per
    // label = number of qualifying eyes.
    __device__ __forceinline__ void mark_alive(int c) {
#pragma unroll
        for (int k = 0; k < NW; ++k) L->aux[pt[k]] = 0;
        __syncthreads();
        bool eye[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            eye[k] = (pt[k] < G::P) && true_eye(k, c);
            if (eye[k]) atomicOr(&L->aux[pt[k]], 1u << 16);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            if (!eye[k]) continue;
            uint32_t gl[4];
            const int pk = own_point(k);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                gl[d] = kNone;
                if (!(nbv[k] >> d & 1)) continue;
                uint32_t g = L->lab[pk + nb_off(d)];
                bool dup = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < d && gl[e] == g) dup = true;
                gl[d] = g;
                if (dup) continue;
                // candidate eye `pt[k]` judged for group g (board.cc:775-811)
                int nb = 0, nt = 0;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    if (!(nbv[k] >> (4 + dd) & 1)) { ++nb; continue; }
                    int q = pk + dg_off(dd);
                    int cq = L->col[q];
                    if (cq == c) { ++nt; continue; }
                    if (cq != kEmpty || !(L->aux[q] >> 16 & 1)) continue;
                    // q is a true eye of colour c: is it a candidate of group g (adjacent to g)?
                    int qx = q % S, qy = q / S;
                    bool adj = false;
                    if (qx > 0 && L->lab[q - 1] == g) adj = true;
                    if (qy > 0 && L->lab[q - S] == g) adj = true;
                    if (qx < S - 1 && L->lab[q + 1] == g) adj = true;
                    if (qy < S - 1 && L->lab[q + S] == g) adj = true;
                    if (adj) ++nt;
                }
                if ((nb >= 1 && nb + nt == 4) || (nb == 0 && nt >= 3)) atomicAdd(&L->aux[g], 1u);
            }
        }
        __syncthreads();
    }
    __device__ __forceinline__ bool alive_at(int k) const {
        uint32_t g = L->lab[pt[k]];
        return lib_at(pt[k]) >= 2 && (L->aux[g] & 0xFFFFu) >= 2;
    }
};

// ---- state-level operations (all lanes of the wave call these together; scalars are wave-uniform) ----------------------

template <int S> __device__ __forceinline__ void state_reset(BoardState<S>& st) {     // board.cc:13-26
#pragma unroll
    for (int k = 0; k < Geo<S>::NW; ++k) { st.bb[0][k] = 0; st.bb[1][k] = 0; }
    st.last_move1 = kInvalid; st.last_move2 = kInvalid; st.ko_location = kInvalid; st.ko_age = 0;
    st.step_count = 1; st.ko_color = 0; st.next_player = kBlack; st.terminated = 0;
    st.pad[0] = st.pad[1] = st.pad[2] = 0;
}

template <int S> __device__ __forceinline__ bool point_in(const uint64_t* bb, int p) { return (bb[p >> 6] >> (p & 63)) & 1; }

// go_env.cc:44-80 Step_ + board.cc:546-653 Play.  `st` is updated in place (wave-uniform copy held by every lane).
// action: [0,P) point, P or -1 pass, -2 resign.  Returns done; *ok = move accepted.  If `check` is false the caller
// guarantees legality (tree search only ever plays moves from a legal list) and the test is skipped.
template <int S>
__device__ __forceinline__ bool state_step(BoardWave<S>& bw, BoardState<S>& st, int action, const RulesCfg& cfg, bool check, bool* ok) {
    using G = Geo<S>;
    *ok = true;
    if (st.terminated) return true;                                   // go_env.cc:52-55
    if (action == G::P) action = kPass;                               // go_env.cc:56-57
    const int player = st.next_player, other = 3 - player;
    if (action == kPass || action == kResign) {                       // board.cc:554-558 (ko untouched)
        st.next_player = (uint8_t)other;
        st.last_move2 = st.last_move1; st.last_move1 = (int16_t)action; st.step_count++;
        bool done = st.step_count > 1 && ((st.last_move1 == kPass && st.last_move2 == kPass) || st.last_move1 == kResign);
        if (done || st.step_count > cfg.max_step) { st.terminated = 1; return true; }
        return false;
    }
    if (action < 0 || action > G::P) { *ok = false; return false; }
    bw.load_colors(st.bb[0], st.bb[1]);
    bw.analyze();
    const int ak = action >> 6, al = action & 63;
    if (check) {
        bool mine = false;
#pragma unroll
        for (int k = 0; k < G::NW; ++k) if (k == ak && bw.lane == al) mine = bw.legal(k, st, player);
        uint64_t b = ballot64(mine);
        if (b == 0) { *ok = false; return false; }                    // go_env.cc:75-79: state unchanged
    }
    // pre-move neighbourhood of the played point (board.cc:90-127), evaluated by every lane redundantly (uniform)
    const int ax = action % S, ay = action / S;
    int self_lib = 0, n11 = 0, ko_at = kInvalid;
    bool own_nb = false;
    uint32_t cap[4] = {kNone, kNone, kNone, kNone};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        bool v = d == 0 ? ax > 0 : d == 1 ? ay > 0 : d == 2 ? ax < S - 1 : ay < S - 1;
        if (!v) continue;
        int q = action + BoardWave<S>::nb_off(d);
        int c = bw.L->col[q];
        if (c == kEmpty) { ++self_lib; continue; }
        if (c == player) { own_nb = true; continue; }
        uint32_t g = bw.L->lab[q];
        if (bw.lib_at(q) == 1) {
            cap[d] = g;                                               // captured: its only liberty is `action`
            // single-stone group <=> no same-colour neighbour (board.cc:181-187)
            int qx = q % S, qy = q / S;
            bool single = !((qx > 0 && bw.L->col[q - 1] == other) || (qy > 0 && bw.L->col[q - S] == other) ||
                            (qx < S - 1 && bw.L->col[q + 1] == other) || (qy < S - 1 && bw.L->col[q + S] == other));
            if (single) { ++n11; ko_at = q; }
        }
    }
    if (self_lib == 0 && !own_nb && n11 == 1) {                       // board.cc:561-570
        st.ko_location = (int16_t)ko_at; st.ko_color = (uint8_t)other; st.ko_age = 0;
    } else {
        st.ko_age++;
    }
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        if (bw.col[k] == other) {
            uint32_t g = bw.L->lab[bw.pt[k]];
            if (g == cap[0] || g == cap[1] || g == cap[2] || g == cap[3]) bw.col[k] = kEmpty;   // board.cc:606-630
        }
        if (k == ak && bw.lane == al) bw.col[k] = (uint8_t)player;
    }
    bw.to_bitboards(st.bb[0], st.bb[1]);
    st.next_player = (uint8_t)other;                                  // board.cc:536-542
    st.last_move2 = st.last_move1; st.last_move1 = (int16_t)action; st.step_count++;
    if (st.step_count > cfg.max_step) { st.terminated = 1; return true; }   // go_env.cc:67
    return false;
}

// Legal points of the side to move as bitboard words (board.cc:467-489).  Position must be loaded + analysed.
template <int S> __device__ __forceinline__ void legal_words(BoardWave<S>& bw, const BoardState<S>& st, uint64_t* out) {
    bw.ensure_suicide(st.next_player);
#pragma unroll
    for (int k = 0; k < Geo<S>::NW; ++k) out[k] = ballot64(bw.pt[k] < Geo<S>::P && bw.legal_cached(k, st, st.next_player));
}

// board_feature.cc:213-253 encode9/10/13 as one bit mask per owned point: bit c of m[k] = plane c at point pt[k].
// Position must be loaded + analysed.  Clobbers L->aux.
template <int S> __device__ __forceinline__ void encode_mask(BoardWave<S>& bw, const BoardState<S>& st, const RulesCfg& cfg, uint32_t* m) {
    using G = Geo<S>;
    const int C = cfg.encode_dim, me = st.next_player, op = 3 - me;
    int pl_h2 = -1, pl_ko = 7, pl_eye = 8, pl_oeye = -1, pl_live = (C == 9) ? -1 : 9, pl_olive = -1;
    if (C == 13) { pl_h2 = 7; pl_ko = 8; pl_eye = 9; pl_oeye = 10; pl_live = 11; pl_olive = 12; }
    bool live_me[G::NW], live_op[G::NW];
#pragma unroll
    for (int k = 0; k < G::NW; ++k) { live_me[k] = false; live_op[k] = false; }
    // liberty classes, ko/suicide and eyes first: mark_alive() clobbers nothing they need, but keep the order explicit
    int cls[G::NW];          // 0/1/2 liberty class of a stone, -1 otherwise
    bool kosu[G::NW], eye_me[G::NW], eye_op[G::NW];
    TG_BW_ST_BEGIN(bw);
    bw.ensure_suicide(me);
    TG_BW_ST(bw, 0);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        cls[k] = -1; kosu[k] = false; eye_me[k] = false; eye_op[k] = false;
        if (bw.pt[k] >= G::P) continue;
        const int c = bw.col[k];
        if (c == kBlack || c == kWhite) {                             // board_feature.cc:17-41
            int l = bw.lib_at(bw.pt[k]);
            cls[k] = l == 1 ? 0 : l == 2 ? 1 : l >= 3 ? 2 : -1;
        } else {
            bool ko = (st.ko_age == 0 && st.ko_location == bw.pt[k]);   // board.cc:205-213 (ko_color ignored)
            kosu[k] = ko || bw.suicide_cached(k);                       // board.cc:520-533, board_feature.cc:69-89
            eye_me[k] = bw.true_eye(k, me);                             // board_feature.cc:142-161
            if (pl_oeye >= 0) eye_op[k] = bw.true_eye(k, op);
        }
    }
    TG_BW_ST(bw, 1);
    if (pl_live >= 0) {                                                 // board_feature.cc:164-182
        bw.mark_alive(me);
#pragma unroll
        for (int k = 0; k < G::NW; ++k) live_me[k] = (bw.col[k] == me) && bw.alive_at(k);
    }
    if (pl_olive >= 0) {
        bw.mark_alive(op);
#pragma unroll
        for (int k = 0; k < G::NW; ++k) live_op[k] = (bw.col[k] == op) && bw.alive_at(k);
    }
    TG_BW_ST(bw, 2);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        const int p = bw.pt[k];
        uint32_t v = 0;
        if (p < G::P) {
            const bool mine = bw.col[k] == me, theirs = bw.col[k] == op;
            if (mine && cls[k] >= 0) v |= 1u << cls[k];
            if (theirs && cls[k] >= 0) v |= 1u << (3 + cls[k]);
            if (st.last_move1 == p) v |= 1u << 6;                       // board_feature.cc:92-100
            if (pl_h2 >= 0 && st.last_move2 == p) v |= 1u << pl_h2;
            if (kosu[k]) v |= 1u << pl_ko;
            if (eye_me[k]) v |= 1u << pl_eye;
            if (pl_oeye >= 0 && eye_op[k]) v |= 1u << pl_oeye;
            if (pl_live >= 0 && live_me[k]) v |= 1u << pl_live;
            if (pl_olive >= 0 && live_op[k]) v |= 1u << pl_olive;
        }
        m[k] = v;
    }
}

// -> f32 planes [C][P] at `out` (plane-major, like the reference's Encode, go_env.cc:96-115).
template <int S> __device__ __forceinline__ void encode_planes(BoardWave<S>& bw, const BoardState<S>& st, const RulesCfg& cfg, float* out) {
    using G = Geo<S>;
    uint32_t m[G::NW];
    encode_mask(bw, st, cfg, m);
    const int C = cfg.encode_dim;
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        const int p = bw.pt[k];
        if (p >= G::P) continue;
#pragma unroll
        for (int c = 0; c < 13; ++c)
            if (c < C) out[c * G::P + p] = (m[k] >> c & 1u) ? 1.f : 0.f;
    }
}

// -> the same planes bit-packed: bit i of the plane-major flat index i = c*P + p, little-endian in u32 words
// (ceil(C*P/32) words; the layout tg_replay_append stores).  `lds` = that many words of LDS scratch (19x19 form only).
// 9x9 (the size every headline configuration runs at): no LDS and no atomics -- plane c's bits of the points k*64..k*64+63 are
// one 64-bit ballot (wave-uniform), and every lane assembles the output words it owns (w = lane, lane + 64, ...) from the ballots
// that overlap them by shifts.  Other sizes OR the bits into an LDS image with atomics (up to 32 lanes per word: ~4x slower per
// position, but 19x19 steps are dominated by the tower anyway).
#ifndef TG_ENCODE_BALLOT19
#define TG_ENCODE_BALLOT19 1
#endif
// INL: inline the packing into the calling kernel (what happens by itself at 9x9).  At 19x19 hipcc keeps the function out of
// line unless told otherwise; see DESIGN.md (round 3, item 6) for why the 19x19 ballot form is only ever used inlined.
template <int S> __device__ __forceinline__ void encode_bits_body(BoardWave<S>& bw, const BoardState<S>& st, const RulesCfg& cfg, uint32_t* lds,
                                             uint32_t* out) {
    using G = Geo<S>;
    uint32_t m[G::NW];
    encode_mask(bw, st, cfg, m);
    const int C = cfg.encode_dim, W = (C * G::P + 31) / 32;
    TG_BW_ST(bw, 3);
    if constexpr (S == 9 || TG_ENCODE_BALLOT19) {
        constexpr int WPL = ((13 * G::P + 31) / 32 + 63) / 64;      // output words per lane (1 at 9x9, 3 at 19x19)
        uint32_t word[WPL];
#pragma unroll
        for (int q = 0; q < WPL; ++q) word[q] = 0;
#pragma unroll
        for (int c = 0; c < 13; ++c) {
            if (c < C) {                                            // wave-uniform
#pragma unroll
                for (int k = 0; k < G::NW; ++k) {
                    const uint64_t B = ballot64((m[k] >> c) & 1u);  // bits of plane c for points k*64 .. (invalid points are 0)
                    const int base = c * G::P + k * 64;             // stream position of B's bit 0
#pragma unroll
                    for (int q = 0; q < WPL; ++q) {
                        const int off = base - 32 * (q * 64 + bw.lane);     // B's bit 0 relative to this lane's word
                        if (off > -64 && off < 32) word[q] |= off >= 0 ? (uint32_t)(B << off) : (uint32_t)(B >> (-off));
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < WPL; ++q) { const int w = q * 64 + bw.lane; if (w < W) out[w] = word[q]; }
        TG_BW_ST(bw, 4);
    } else {
        for (int i = bw.lane; i < W; i += 64) lds[i] = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < G::NW; ++k) {
            const int p = bw.pt[k];
#pragma unroll
            for (int c = 0; c < 13; ++c)
                if (c < C && (m[k] >> c & 1u)) { const int i = c * G::P + p; atomicOr(&lds[i >> 5], 1u << (i & 31)); }
        }
        __syncthreads();
        for (int i = bw.lane; i < W; i += 64) out[i] = lds[i];
        __syncthreads();
    }
}

template <int S> __device__ __noinline__ void encode_bits_call(BoardWave<S>& bw, const BoardState<S>& st, const RulesCfg& cfg, uint32_t* lds,
                                                              uint32_t* out) { encode_bits_body(bw, st, cfg, lds, out); }
// TG_ENCODE_INLINE_MASK (diagnostic builds): bit 0 k_reset, bit 1 k_collect, bit 2 k_play take the inlined form; default: the
// compiler decides for the LDS form, and the ballot form is always inlined
#ifndef TG_ENCODE_INLINE_MASK
#define TG_ENCODE_INLINE_MASK -1
#endif
template <int S, int SITE> __device__ __forceinline__ void encode_bits(BoardWave<S>& bw, const BoardState<S>& st, const RulesCfg& cfg,
                                                                       uint32_t* lds, uint32_t* out) {
    if constexpr (S == 9 || TG_ENCODE_INLINE_MASK == -1 || ((TG_ENCODE_INLINE_MASK >> SITE) & 1)) encode_bits_body(bw, st, cfg, lds, out);
    else encode_bits_call(bw, st, cfg, lds, out);
}

// board.cc:822-958 getTTScore.  Loads colours itself.  Returns raw Tromp-Taylor area difference (0 on an empty board,
// board.cc:932-935); owner[k] (optional, per lane slot): 1 black, 2 white, 3 dame.
template <int S> __device__ __forceinline__ float tromp_taylor(BoardWave<S>& bw, const BoardState<S>& st, uint8_t* owner) {
    using G = Geo<S>;
    bw.load_colors(st.bb[0], st.bb[1]);
    bw.label_groups(true);
#pragma unroll
    for (int k = 0; k < G::NW; ++k) bw.L->cnt[bw.pt[k]] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        if (bw.col[k] != kEmpty) continue;
        uint32_t touch = 0;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (!(bw.nbv[k] >> d & 1)) continue;
            int c = bw.L->col[bw.pt[k] + BoardWave<S>::nb_off(d)];
            if (c == kBlack || c == kWhite) touch |= (uint32_t)c;
        }
        if (touch) atomicOr(&bw.L->cnt[bw.L->lab[bw.pt[k]]], touch);
    }
    __syncthreads();
    int nb = 0, nw = 0, ns = 0;
#pragma unroll
    for (int k = 0; k < G::NW; ++k) {
        int o = 0;
        if (bw.col[k] == kBlack || bw.col[k] == kWhite) o = bw.col[k];
        else if (bw.col[k] == kEmpty) { uint32_t t = bw.L->cnt[bw.L->lab[bw.pt[k]]]; o = (t == 1) ? 1 : (t == 2) ? 2 : 3; }
        if (owner) owner[k] = (uint8_t)o;
        nb += __popcll(ballot64(o == 1 && bw.pt[k] < G::P));
        nw += __popcll(ballot64(o == 2 && bw.pt[k] < G::P));
        ns += __popcll(ballot64(bw.col[k] == kBlack || bw.col[k] == kWhite));
    }
    __syncthreads();
    if (ns == 0) return 0.f;
    return (float)(nb - nw);
}

}  // namespace tg
