// net.hip -- policy/value/ownership network forward for gfx950 (replaces TransGoNetwork.main_prediction,
// model.py:17-20 -> MainNetwork.forward, model.py:79-114, built from CNNBlock model.py:317-324 and the pre-activation
// ResidualBlock model.py:238-248; inference only, BatchNorm in eval mode).
//
// Tower: stem conv3x3(C->F)+BN+ReLU, N x [BN,ReLU,conv3x3,BN,ReLU,conv3x3,+x], BN+ReLU, then the two heads of
// model.py:65-76 / :97-111 (value/ownership: conv3x3(F->2)+BN+ReLU, FC(2P->64)+ReLU, FC(64->1) tanh, FC(64->P) tanh;
// policy: conv3x3(F->4)+BN+ReLU, FC(4P->P+1), softmax).
//
// The 3x3 convolutions are the one dense contraction of the whole engine and run on the matrix cores as an implicit
// GEMM in exact fp32 (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, so results stay within 1e-3 of torch fp32):
//   D[cout][position] += W[tap][cout][cin] * X[position + tap][cin]
// Activations are position-major / channel-minor (NHWC flattened over the batch: row m = leaf*P + point), so a tile of
// 128 consecutive rows plus a halo of S+1 rows on each side contains every 3x3 neighbour of its rows regardless of
// where leaf boundaries fall; taps that leave the board are zeroed in registers from a 9-bit per-row mask instead
// of materialising padded boards.  BatchNorm is folded on the host (transgo_amd/model.py): BN that follows a conv goes
// into its weights/bias, BN that precedes one (pre-activation) is applied with ReLU while the tile is staged into LDS.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

using namespace tg;

namespace tg {

struct ConvW { const float* w; const float* b; };
struct BlockW { const float* s1; const float* t1; ConvW c1; ConvW c2; const float* g1; const float* g2;
                const _Float16* h1; const _Float16* h2; };   // g1/g2: stage-ordered f32, h1/h2: stage-ordered fp16 copies of c1.w / c2.w
struct AttW { ConvW qkv; const float* gamma; const float* s; const float* t;          // Self_Attention, model.py:288-315
              const _Float16* x3w = nullptr; const float* x3sc = nullptr; };          // split q|k|v weights (LDS image) + 2^-s of k_attention_x3
struct Layer { int kind; int ridx; AttW a; };                                         // kind 0: residual block blocks[ridx]; 1: attention

struct Net {
    int F = 0, NB = 0, C = 0, S = 0, P = 0, A = 0;
    float* blob = nullptr; size_t blob_floats = 0;
    float* wstage = nullptr; // [2*NB][F/16*9][F][16] stage-ordered F->F conv weights (k_conv3x3_sg)
    ConvW stem; std::vector<BlockW> blocks; std::vector<Layer> layers; const float* s_end = nullptr; const float* t_end = nullptr;
    bool pol_att = false; AttW patt; ConvW head_a; std::string arch;
    float* bufQ = nullptr; float* hca = nullptr;   // q|k|v projections [rows][P][1.5F]; policy head conv output
    float* bufAct = nullptr;                       // pre-activated input of the next conv (DMA path)
    int* tile_ctr = nullptr;                       // [2*NB] tile counters of the persistent conv launches (zeroed per forward)
    int prec = 0;                                  // cfg.net_precision: 0 = f32, 1 = fp16 storage + f32 accumulate (k_conv3x3_h2), 2 = 1 + fp16 residual stream,
                                                   // 3 = split precision ("f32x3"): every operand as fp16 hi + lo, three of the four products on the fp16 MFMA, f32 accumulate
    const float* head_g = nullptr; const float* head_ag = nullptr;   // [64][F] head conv weights for k_head_gemm (tap*6 + cout rows)
    const _Float16* head_x2 = nullptr; const float* head_x2sc = nullptr;   // prec 3, F = 128: the value/policy head conv's split weights (k_head_gemm_x2) + 2^-s
    float* wsc = nullptr;                          // prec 3: [2*NB + 1] power-of-two factor 2^-s each conv's weights were scaled by before splitting (stem last)
    _Float16* stem_h = nullptr; _Float16* head_h = nullptr; _Float16* x0h = nullptr;   // fp16 path: stem [2*9][F][32] (16 planes padded to 64), head [F/32*9][16][32], input [rows][P][64]
    _Float16* wh = nullptr; _Float16* act16 = nullptr; _Float16* h16 = nullptr;   // fp16 path: weights [2*NB][F/32*9][F][32], activations [rows][P][F]
    int dma = 0;                                   // attention-free F=128/256 f32 tower: 1 = k_conv3x3_sg chain (default), 0 = k_conv3x3 (TG_DMA_CONV=0)
    ConvW head; const float* w_vo = nullptr; const float* b_vo = nullptr; const float* w_v = nullptr; const float* b_v = nullptr;
    const float* w_o = nullptr; const float* b_o = nullptr; const float* w_a = nullptr; const float* b_a = nullptr;
    // engine batches arrive bit-packed and sparse (engine.h: obs_bits through row_slot); tg_net_predict hands float planes
    const uint32_t* in_bits = nullptr; const int32_t* in_slot = nullptr; int in_words = 0;
    float* x0 = nullptr;   // [rows][P][16] input planes, channel-minor
    float* bufA = nullptr; float* bufB = nullptr; float* bufH = nullptr;   // [rows][P][F]
    float* hc = nullptr;   // [rows][P][16] head conv output
    float* own = nullptr;  // [rows][P]
    int rows_cap = 0;
    // range guard: [0] sticky count of epilogue tiles (conv) / boards (attention) that rounded a value beyond +-65504 to fp16 since
    // the network was created (fp16-carrying precisions only); [1 + k] bit pattern of max |w| over weight set k's BN-folded blob
    unsigned* range = nullptr;
    // Two complete weight sets (f2, "double-buffered so search never stalls"): the forward pass reads set `active` (the pointer
    // fields above are bound to it); a refresh fills the other one -- tg_net_load_async on a side stream while searches keep
    // running -- and the next forward that finds it complete rebinds.  `swapped` orders a later refill of the retired set behind
    // every kernel that may still read it.
    struct WeightSet { float* blob = nullptr; float* wstage = nullptr; _Float16* wh = nullptr; _Float16* stem_h = nullptr; _Float16* head_h = nullptr;
                       float* wsc = nullptr; float* head_g = nullptr; float* head_ag = nullptr;
                       _Float16* att_h = nullptr; float* att_sc = nullptr;
                       _Float16* head_x2 = nullptr; float* head_x2sc = nullptr; };                          // k_attention_x3: per attention layer, split weight image + scale   // head_g / head_ag: k_head_gemm copies of head / head_a
    WeightSet sets[2]; int active = 0; bool pending = false;
    hipStream_t side = nullptr; hipEvent_t loaded = nullptr, swapped = nullptr; float* pinned = nullptr;
    // profiling of the dominant kernel (3x3 conv F->F) with HIP events on the launch stream
    bool prof = false; std::vector<hipEvent_t> ev; size_t ev_used = 0; double conv_ms = 0; long conv_launches = 0; double conv_flops = 0;
    long conv_skipped = 0;                         // launches that found the event pool full (drain it with tg_prof_read between steps)
};

}  // namespace tg

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using h2 = __attribute__((ext_vector_type(2))) _Float16;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h4 = __attribute__((ext_vector_type(4))) _Float16;
__device__ __forceinline__ int tid0() { return (int)threadIdx.x; }



template <int S>
__global__ __launch_bounds__(256) void k_obs_to_rows(const float* __restrict__ obs, float* __restrict__ x0, int rows, int C) {
    constexpr int P = S * S;
    const size_t total = (size_t)rows * P * 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 15);
        const size_t m = i >> 4;
        const int p = (int)(m % P);
        const size_t r = m / P;
        x0[i] = c < C ? obs[(r * C + c) * P + p] : 0.f;
    }
}


// The same from the engine's evaluation batch: row r = entry row_slot[r] of the bit-packed batch (bit c*P + p of its W words).
template <int S>
__global__ __launch_bounds__(256) void k_bits_to_rows(const uint32_t* __restrict__ bits, const int32_t* __restrict__ row_slot,
                                                      float* __restrict__ x0, int rows, int C, int W) {
    constexpr int P = S * S;
    const size_t total = (size_t)rows * P * 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 15);
        const size_t m = i >> 4;
        const int p = (int)(m % P), k = c * P + p;
        const size_t r = m / P;
        x0[i] = c < C ? (float)((bits[(size_t)row_slot[r] * W + (k >> 5)] >> (k & 31)) & 1u) : 0.f;
    }
}

// Conv epilogue shared by the conv kernels.  acc[ct][t] holds couts ct*16 + kq*4 .. +3 of row mrow[t].
// Loads and stores retire through ONE in-order counter (vmcnt), so a load issued after stores waits for all of them: per-cout
// parameters therefore come from LDS (par = bias | s2 | t2, COUT floats each), and the residual is fetched one 4-tile chunk
// ahead of the stores of the previous chunk, which turns ~CT*NPT serialized memory round trips into counted, overlapped ones.
// EPI 0: relu(acc + bias)   EPI 1: acc + bias + res   EPI 2: acc + bias;   out2 (optional) = relu(v * s2 + t2).
// SM / SM2: `out` / `out2` is a SLICE-MAJOR f32 tensor [COUT/16][M][16] (f32_sm_index) -- the layout the DMA-fed kernel reads its
// inputs in: a slab piece (16 rows x 64 B) is then one contiguous KB, and so is what one store instruction here writes.
__device__ __forceinline__ size_t f32_sm_index(int m, int c, int M) { return ((size_t)(c >> 4) * M + m) * 16 + (c & 15); }

template <int COUT, int CT, int NPT, int EPI, bool SM = false, bool SM2 = false>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[CT][NPT], const int (&mrow)[NPT], int M, int co_base, int kq,
                                              float* __restrict__ out, const float* __restrict__ res, float* __restrict__ out2,
                                              const float* par) {
    constexpr int CH = CT % 4 == 0 ? 4 : CT % 3 == 0 ? 3 : CT % 2 == 0 ? 2 : 1;   // cout tiles per chunk
    constexpr int NCH = (CT / CH) * NPT;
    static_assert(CT % CH == 0, "chunking");
    f32x4 r[2][CH];
    auto load_res = [&](int c, f32x4* dst) {
        const int t = c / (CT / CH), h = c % (CT / CH);
        if (mrow[t] < M) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
                dst[i] = *reinterpret_cast<const f32x4*>(res + (size_t)mrow[t] * COUT + co_base + (h * CH + i) * 16 + kq * 4);
        }
    };
    if (EPI == 1) load_res(0, r[0]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int t = c / (CT / CH), h = c % (CT / CH);
        if (EPI == 1 && c + 1 < NCH) load_res(c + 1, r[(c + 1) & 1]);
        if (mrow[t] < M) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int ct = h * CH + i;
                const int col = ct * 16 + kq * 4;      // column within this kernel's cout range
                f32x4 v = acc[ct][t] + *reinterpret_cast<const f32x4*>(par + col);
                if (EPI == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                } else if (EPI == 1) {
                    v = v + r[c & 1][i];
                }
                *reinterpret_cast<f32x4*>(out + (SM ? f32_sm_index(mrow[t], co_base + col, M) : (size_t)mrow[t] * COUT + co_base + col)) = v;
                if (out2) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(par + COUT + col);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(par + 2 * COUT + col);
                    f32x4 u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { float w = v[e] * sc[e] + sh[e]; u[e] = w > 0.f ? w : 0.f; }
                    *reinterpret_cast<f32x4*>(out2 + (SM2 ? f32_sm_index(mrow[t], co_base + col, M) : (size_t)mrow[t] * COUT + co_base + col)) = u;
                }
            }
        }
    }
}

// EPI 0: out = relu(acc + bias)      EPI 1: out = acc + bias + res      EPI 2: out = acc + bias
// NTAP 9: 3x3 convolution with zero padding; NTAP 1: 1x1 convolution (the q/k/v projections of Self_Attention, model.py:294-296)
template <int S, int CIN, int COUT, bool PRO, int EPI, int NTAP = 9, int NPT = 2, bool SM2 = false>
// NPT = 3 only pays with two waves per SIMD (<= 256 registers, a handful of spills): measured 132 vs 118 TFLOP/s at one
__global__ __launch_bounds__(256, (NPT > 2 ? 2 : 1)) void k_conv3x3(const float* __restrict__ in, float* __restrict__ out,
                                                 const float* __restrict__ res, const float* __restrict__ Wt,
                                                 const float* __restrict__ bias, const float* __restrict__ ps,
                                                 const float* __restrict__ pt, int M, float* __restrict__ out2 = nullptr,
                                                 const float* __restrict__ s2 = nullptr, const float* __restrict__ t2 = nullptr) {
    constexpr int P = S * S, HALO = NTAP == 9 ? S + 1 : 0;
    constexpr int TM = 64 * NPT;                    // output rows per workgroup: 4 waves x NPT position tiles x 16
    constexpr int CC = CIN < 32 ? CIN : 32;         // input channels staged per pass
    constexpr int RS = CC + 4;                      // LDS row stride (floats)
    constexpr int NROW = TM + 2 * HALO;
    constexpr int CT = COUT / 16;                   // cout tiles, all held by every wave
    constexpr int C4 = CC / 4;
    __shared__ float xs[NROW * RS];
    __shared__ __attribute__((aligned(16))) float par[3 * COUT];            // bias | s2 | t2 for the epilogue
    for (int i = tid0(); i < COUT; i += 256) { par[i] = bias[i]; par[COUT + i] = out2 ? s2[i] : 0.f; par[2 * COUT + i] = out2 ? t2[i] : 0.f; }
    // WALL: a narrow output (the 16-wide head conv) is latency-bound, not MFMA-bound -- stage all NTAP weight tiles of a channel
    // slice at once (2 barriers per slice instead of 2 per tap; 480 -> ~ us per forward at 16384 leaves)
    constexpr bool WALL = COUT <= 16 && NTAP == 9;
    constexpr int WT = WALL ? NTAP : 1;             // weight tiles resident in LDS
    __shared__ float ws[WT * COUT * RS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int j = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * TM;

    unsigned vmask[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int m = m0 + (wave * NPT + t) * 16 + j;
        unsigned mk = 0;
        if (m < M) {
            const int p = m % P, x = p % S, y = p / S;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) mk |= 1u << tap;
            }
            if (NTAP == 1) mk = 1u;
        }
        vmask[t] = mk;
    }
    f32x4 acc[CT][NPT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < NPT; ++t) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the global loads of stage s+1 (one tap's weights; every 9th stage also the next 32-channel
    // slice of the activation tile) are issued before the MFMA block of stage s and land in registers while it runs;
    // they are written to LDS after the barrier that ends it.
    constexpr int WL = (WT * COUT * C4 + 255) / 256;
    constexpr int XL = (NROW * C4 + 255) / 256;
    f32x4 wreg[WL], xreg[XL];
    auto load_w = [&](int cc, int tap) {
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int idx = tid + i * 256;
            if (idx < WT * COUT * C4) {                                   // WALL: idx also runs over the taps (tap argument = 0)
                const int co = idx / C4, c4 = idx % C4;
                wreg[i] = *reinterpret_cast<const f32x4*>(Wt + ((size_t)tap * COUT + co) * CIN + cc + c4 * 4);
            }
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            const int idx = tid + i * 256;
            if (idx < WT * COUT * C4) { const int co = idx / C4, c4 = idx % C4; *reinterpret_cast<f32x4*>(&ws[co * RS + c4 * 4]) = wreg[i]; }
        }
    };
    auto load_x = [&](int cc) {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int idx = tid + i * 256;
            xreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < NROW * C4) {
                const int r = idx / C4, c4 = idx % C4;
                const int m = m0 - HALO + r;
                if (m >= 0 && m < M) xreg[i] = *reinterpret_cast<const f32x4*>(in + (size_t)m * CIN + cc + c4 * 4);
            }
        }
    };
    auto store_x = [&](int cc) {
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int idx = tid + i * 256;
            if (idx < NROW * C4) {
                const int r = idx / C4, c4 = idx % C4;
                f32x4 v = xreg[i];
                if (PRO) {
                    const int m = m0 - HALO + r;
                    if (m >= 0 && m < M) {             // the zero rows outside the batch stay zero (padding is post-activation)
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(ps + cc + c4 * 4);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(pt + cc + c4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float u = v[e] * sc[e] + sh[e]; v[e] = u > 0.f ? u : 0.f; }
                    }
                }
                *reinterpret_cast<f32x4*>(&xs[r * RS + c4 * 4]) = v;
            }
        }
    };
    load_x(0);
    load_w(0, 0);
    for (int cc = 0; cc < CIN; cc += CC) {
        for (int tap = 0; tap < NTAP; ++tap) {
            if (!WALL || tap == 0) {
                __syncthreads();
                if (tap == 0) store_x(cc);
                store_w();
                __syncthreads();
                if (!WALL && tap < NTAP - 1) load_w(cc, tap + 1);
                else if (cc + CC < CIN) { load_w(cc + CC, 0); load_x(cc + CC); }
            }
            const int toff = NTAP == 9 ? (tap / 3 - 1) * S + (tap % 3 - 1) : 0;
#pragma unroll
            for (int sub = 0; sub < CC / 16; ++sub) {
                f32x4 b[NPT];
#pragma unroll
                for (int t = 0; t < NPT; ++t) {
                    const int r = (wave * NPT + t) * 16 + j + HALO + toff;
                    b[t] = *reinterpret_cast<const f32x4*>(&xs[r * RS + sub * 16 + kq * 4]);
                    if (!((vmask[t] >> tap) & 1)) b[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(&ws[((WALL ? tap * COUT : 0) + ct * 16 + j) * RS + sub * 16 + kq * 4]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < NPT; ++t)
                            acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[ct][t], 0, 0, 0);
                }
            }
        }
    }
    // D tile: row (lane>>4)*4 + r = cout, column lane&15 = position
    int mrow[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) mrow[t] = m0 + (wave * NPT + t) * 16 + j;
    conv_epilogue<COUT, CT, NPT, EPI, false, SM2>(acc, mrow, M, 0, kq, out, res, out2, par);
}

// LDS pointer type of the LDS-DMA builtins; counted wait on the in-order vector-memory counter (loads, stores and LDS-DMA share it)
typedef __attribute__((address_space(3))) void tg_lds_void;
#define TG_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// raw s_barrier fenced for the COMPILER on both sides: the LDS-DMA writes are invisible to it (inline assembly, see below), so
// nothing else stops it from scheduling a fragment read of the next stage above the barrier that publishes that stage
#define TG_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

// LDS-DMA pieces (64 lanes x 16 B -> 1 KB of LDS at `lds`, lane-linear) issued as INLINE ASSEMBLY on purpose.  Through the
// builtins hipcc knows that LDS is being written asynchronously, cannot see that the kernels' own counted s_waitcnt + s_barrier
// already order every fragment read behind the pieces it needs, and inserts `s_waitcnt vmcnt(0)` in front of the next ds_read --
// which makes every wave wait for the pieces it has JUST issued (two stages of landing slack thrown away, once per stage).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__device__ __forceinline__ u32x4 tg_rsrc(const void* base, unsigned bytes) {       // raw buffer: stride 0, bounds check on the byte offset
    const unsigned long long a = (unsigned long long)base;
    return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void tg_dma_global(const void* sbase /*wave-uniform*/, int voff_bytes, tg_lds_void* lds) {
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds);     // wave-uniform by construction; makes it an SGPR
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(voff_bytes), "s"(sbase) : "memory");
}
__device__ __forceinline__ void tg_dma_buffer(u32x4 rsrc, int voff_bytes, tg_lds_void* lds) {    // out-of-range offsets write zeros
    const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(la), "v"(voff_bytes), "s"(rsrc) : "memory");
}

// XOR swizzle of the 16-B chunk index for LDS images with 64-byte rows read as MFMA fragments by ds_read_b128 (lane = (j, kq):
// row base + j, chunk kq).  The instruction is served in four 16-lane groups, each holding all 16 values of j with kq = q0 for
// j in {0-3, 12-15} and q0^1 for j in {4-11} (MI355X_MICROARCH.md, LDS table); the 16-B slot of a lane is (row%4)*4 + chunk, so the
// four rows of one residue class, which use chunks (q0, q0^1, q0^1, q0) in row order, must land on four distinct chunks for every
// alignment of the base row: chunk ^ 2*((row>>2)&1) does that ({c, c^3, c^1, c^2}), the obvious chunk ^ ((row>>2)&3) is 2-way
// conflicted on every read (measured: SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE).
__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

#ifndef TG_H2_XCD
#define TG_H2_XCD 0        // experiment switch: XCD-contiguous row-tile order in k_conv3x3_h2
#endif
#ifndef TG_X2_3P
#define TG_X2_3P 1         // split precision: three partial products per stage pair (0: all four)
#endif
#ifndef TG_SG_PERSIST128
#define TG_SG_PERSIST128 0
#endif
#ifndef TG_SG_XCD
#define TG_SG_XCD 1        // XCD-contiguous tile order in k_conv3x3_sg (F = 128): HBM reads 1.21x -> 1.12x (EPI 0) / 1.10x -> 1.05x (EPI 1) of the input bytes, time unchanged; 0 = linear order
#endif
#ifndef TG_SG_PRIO
#define TG_SG_PRIO 0       // experiment switch: wave priority around the MFMA cluster (1), the epilogue (2), the weight-DMA issue (4)
#endif
#ifndef TG_SG_NG
#define TG_SG_NG 2           // k_conv3x3_sg at F=128: stages per barrier (ring = 2*NG slots of 8 KB); measured 2: 138.5, 3: 131.6, 4: 127.1 TFLOP/s
#endif
// ---- F->F 3x3 conv of the f32 tower (F = 128 / 256, attention-free; input already activated by its producer) -------------------
// Implicit GEMM on v_mfma_f32_16x16x4_f32: D[cout][pos] += W[tap][cout][cin] * X[pos + tap][cin].  F = 128: workgroup = 4 waves =
// 192 consecutive rows x all 128 couts (wave: 3 position x 8 cout tiles, 96 accumulator registers), <= 162 VGPRs and 34 KB of LDS,
// so THREE workgroups share a CU (three waves per SIMD fill each other's bubbles) and 16384 leaves are exactly 9 rounds of the 768
// resident slots.  F = 256: 128 rows x 256 couts (2 x 16 tiles), two workgroups per CU.
// Stage g = (16-channel slice, tap).  A (weights): the stage's tile arrives by LDS-DMA from a stage-ordered copy of the weights
// ([slice*9+tap][cout][16]) into a 4-slot ring, two stages per barrier, the next pair landing while this one is used; the LDS
// image is XOR-swizzled at the source (swz64) so ds_read_b128 fragments are conflict-free.  B (activations): inputs are
// slice-major, so a B fragment (16 rows x 64 B of one slice) is ONE contiguous KB in memory and every wave loads its own B
// fragments straight from L2 into registers, a stage ahead (out-of-board taps and rows outside the batch get an out-of-range
// buffer offset, i.e. zeros) -- no activation slab in LDS, no zero-row select.  Everything is at least a stage old when it is
// waited for, so each stage simply ends with s_waitcnt vmcnt(0).  An earlier design staged the activations as LDS slabs
// (k_conv3x3_sd, one barrier per stage, 53 KB): 1-3.5 % slower on every shape, removed.
// NPT = position tiles per wave (rows per workgroup = 64*NPT).  F = 128 is built for 3 (192 rows, three workgroups per CU: the best
// shape when the batch fills whole rounds of the 768 resident slots, e.g. exactly 16384 leaves = 9.0 rounds) and for 2 (128 rows,
// four per CU, 1024 slots): in steady state a wave evaluates ~15.7 k leaves = 8.63 rounds of the first shape, whose partial last
// round costs almost a full one; the host picks per launch whichever shape wastes less of its last round (+2 % in steady state).
template <int S, int F, int EPI, int NPT_ = (F == 128 ? 3 : 2)>
__global__ __launch_bounds__(256, (F == 128 ? (NPT_ == 3 ? 3 : 4) : 2)) void k_conv3x3_sg(const float* __restrict__ in, float* __restrict__ out,
                                                                       const float* __restrict__ res, const float* __restrict__ Ws,
                                                                       const float* __restrict__ bias, float* __restrict__ out2,
                                                                       const float* __restrict__ s2, const float* __restrict__ t2, int M,
                                                                       int* __restrict__ ctr) {
    constexpr int P = S * S, CT = F / 16, CC = 16;
    constexpr int NPT = NPT_, TM = 64 * NPT;
    constexpr int WPW = CT / 4;
    constexpr int NSL = F / CC, NST = NSL * 9, NG = TG_SG_NG, D = F == 128 ? 2 * NG : 4, NGS = D / 2, NGRP = NST / NGS;   // NGS stages per barrier, two groups resident
    static_assert((F == 128 || F == 256) && NST % NGS == 0, "tile geometry");
    __shared__ __attribute__((aligned(16))) float ws[D][F * CC];
    __shared__ __attribute__((aligned(16))) float par[3 * F];
    __shared__ int next_tile;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // XCD-aware tile order (one tile per workgroup, F = 128): workgroup b runs on XCD b % 8, so XCD x takes the CONTIGUOUS tile range
    // [x * tpx, (x + 1) * tpx) -- neighbouring tiles share their halo rows through one L2 instead of fetching them into two
    int bid = blockIdx.x;
    if (TG_SG_XCD && !(F == 256 || TG_SG_PERSIST128)) { const int tpx = ((int)gridDim.x + 7) >> 3; bid = (bid & 7) * tpx + (bid >> 3); }
    int m0 = bid * TM;
    if (TG_SG_XCD && m0 >= M) return;                                   // the grid is rounded up to a multiple of 8
    for (int i = tid; i < F; i += 256) { par[i] = bias[i]; par[F + i] = out2 ? s2[i] : 0.f; par[2 * F + i] = out2 ? t2[i] : 0.f; }

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, M * F * 4, 0x00020000);
    const int prow = lane >> 2, pchunk = (lane & 3) ^ swz64(lane >> 2);
    auto dma_w = [&](int g) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int pc = wave * WPW + i;
            tg_dma_global(Ws + (size_t)g * (F * CC) + pc * 16 * CC, (prow * CC + pchunk * 4) * 4, (tg_lds_void*)(&ws[g % D][pc * 256]));
        }
    };
    // F = 256 runs persistent with a DYNAMIC tile list: the first gridDim.x tiles are the block indices, every further tile index
    // comes from an atomic counter (fetched by one lane while the current tile computes, published through LDS behind the loop's
    // barriers), so the dispatcher-like balance is kept while the next tile's first weights go out before the current tile's stores
    // and its residual / first B fragments right after them (+1.2-1.7 %).  At F = 128 (shorter tiles, three workgroups per CU) the
    // same loop costs 2 %, so there every workgroup takes exactly one tile.
    constexpr bool PERSIST = F == 256 || TG_SG_PERSIST128;
    const int ntiles = (M + TM - 1) / TM;
    const int aoff = j * CC + ((kq ^ swz64(j)) << 2);
    unsigned vmask[NPT]; int boff[NPT];
    f32x4 acc[CT][NPT], b_cur[NPT], b_next[NPT];
    auto load_b = [&](f32x4* b, int g) {
        const int sl = g / 9, tap = g % 9;
        const int soff = (sl * M + (tap / 3 - 1) * S + (tap % 3 - 1)) * (CC * 4);       // wave-uniform: slice base + tap shift
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int voff = ((vmask[t] >> tap) & 1) ? boff[t] + soff : 0x7ffffff0;    // masked: out of range = zeros
            b[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, voff, 0, 0));
        }
    };
    auto tile_setup = [&]() {                                            // masks, accumulator start values and stage-0 fragments of tile m0
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int m = m0 + (wave * NPT + t) * 16 + j;
            unsigned mk = 0;
            if (m < M) {
                const int p = m % P, x = p % S, y = p / S;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                    if (yy >= 0 && yy < S && xx >= 0 && xx < S) mk |= 1u << tap;
                }
            }
            vmask[t] = mk; boff[t] = (m * CC + kq * 4) * 4;              // byte offset of this lane's 16 B inside slice 0
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {                            // EPI 1: start from the residual (no loads behind the epilogue's stores)
                acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                // always issued (rows past the batch re-read the last row; they are never stored): the wait at the loop head counts them
                if (EPI == 1) acc[ct][t] = *reinterpret_cast<const f32x4*>(res + (size_t)(m < M ? m : M - 1) * F + ct * 16 + kq * 4);
            }
        }
        load_b(b_cur, 0);
    };
#pragma unroll
    for (int g = 0; g < D; ++g) dma_w(g);
    tile_setup();
    for (;;) {
        if (PERSIST && tid == 0) next_tile = (int)gridDim.x + atomicAdd(ctr, 1);    // read by everybody after the loop's barriers
        // The barrier must see the weight DMAs of the first stages landed (and, in a persistent walk, the previous tile's stores
        // gone); the operations tile_setup() issued after them -- CT*NPT residual loads, NPT B fragments -- are the wave's youngest
        // and are waited for by the compiler, one by one, in front of the MFMA that first needs each.
        TG_VMCNT((EPI == 1 ? CT * NPT : 0) + NPT);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TG_BARRIER();

#pragma unroll 1
        for (int pp = 0; pp < NGRP; ++pp) {
#pragma unroll
            for (int h = 0; h < NGS; ++h) {
                const int g = NGS * pp + h;
                const float* wcur = ws[g % D];
                if (g + 1 < NST) load_b(b_next, g + 1);
                __builtin_amdgcn_sched_barrier(0);                       // keep the loads HERE: hipcc sinks them to their use, a stage later
#if TG_SG_PRIO & 1
                __builtin_amdgcn_s_setprio(1);
#endif
                f32x4 a_cur = *reinterpret_cast<const f32x4*>(wcur + aoff);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x4 a_next = a_cur;
                    if (ct + 1 < CT) a_next = *reinterpret_cast<const f32x4*>(wcur + (ct + 1) * 256 + aoff);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                        for (int t = 0; t < NPT; ++t)
                            acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s4], b_cur[t][s4], acc[ct][t], 0, 0, 0);
                    a_cur = a_next;
                }
#if TG_SG_PRIO & 1
                __builtin_amdgcn_s_setprio(0);
#endif
                TG_VMCNT(0);                                             // B fragments of stage g+1 (and any weight pieces): a stage old
#pragma unroll
                for (int t = 0; t < NPT; ++t) b_cur[t] = b_next[t];
            }
            TG_BARRIER();                                                // group pp+1 has landed for everybody; the slots of group pp are free
            if (NGS * (pp + 2) < NST) {
#if TG_SG_PRIO & 4
                __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
                for (int h = 0; h < NGS; ++h) dma_w(NGS * (pp + 2) + h);
#if TG_SG_PRIO & 4
                __builtin_amdgcn_s_setprio(0);
#endif
            }
        }
        int mrow[NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) mrow[t] = m0 + (wave * NPT + t) * 16 + j;
        const int next = PERSIST ? __builtin_amdgcn_readfirstlane(next_tile) : ntiles;     // written before the loop's first barrier
        const bool more = next < ntiles;
        if (more) {                                                      // every wave is past the last barrier: the ring is free
#pragma unroll
            for (int g = 0; g < D; ++g) dma_w(g);
        }
#if TG_SG_PRIO & 2
        __builtin_amdgcn_s_setprio(3);
#endif
        conv_epilogue<F, CT, NPT, (EPI == 1 ? 2 : EPI), EPI == 0, true>(acc, mrow, M, 0, kq, out, res, out2, par);
#if TG_SG_PRIO & 2
        __builtin_amdgcn_s_setprio(0);
#endif
        if (!more) break;
        m0 = next * TM;
        tile_setup();
        TG_BARRIER();                                                    // everybody has read next_tile before it is overwritten
    }
}

// fp16 activation tensors are CHUNK-MAJOR: [channels/32 slices][4 chunks of 8 channels][M rows][8 halfs].  A slab DMA piece (64 rows
// x 16 B of one chunk plane) is one contiguous KB in memory (a row-major piece is 16 half-used lines 2F bytes apart and costs the
// texture-address path ~3x as much), and the LDS image it builds -- plane after plane, rows 16 B apart -- is read by ds_read_b128
// MFMA fragments without bank conflicts and WITHOUT a swizzle: lane (row j, chunk kq) reads plane kq at row R + j, the 16-lane
// service groups of the instruction (MI355X_MICROARCH.md, LDS table) then cover 64 distinct banks as long as a plane is a multiple
// of 16 rows, so a tap is a constant byte offset (round 1 kept rows of 64 B and XOR-swizzled the chunk index by the row, which cost
// ~8 vector instructions of address arithmetic per fragment).  Element (row m, channel c):
__device__ __forceinline__ size_t h16_index(int m, int c, int M) { return (((size_t)(c >> 5) * 4 + ((c >> 3) & 3)) * M + m) * 8 + (c & 7); }

// s_waitcnt vmcnt(n) for a wave-uniform n in [0, N]: the instruction takes an immediate
template <int N>
__device__ __forceinline__ void vmcnt_uniform(int n) {
    if (n >= N) TG_VMCNT(N);
    else if constexpr (N > 0) vmcnt_uniform<N - 1>(n);
}

// Epilogue of k_conv3x3_h2 in the PAIRED cout mapping: the weights are restaged (k_restage_half, pair8) so that row i of A tile ct
// is physical cout (ct >> 1) * 32 + (i >> 2) * 8 + (ct & 1) * 4 + (i & 3); lane kq then holds, across the tile pair (2u, 2u + 1),
// the 8 CONSECUTIVE couts u * 32 + kq * 8 .. + 7 of its row -- one whole 16-B chunk of the chunk-major fp16 tensors.  Every fp16
// access of the epilogue (and the fp16 residual read that starts the accumulators) is one 16-B access per lane instead of two
// 8-B ones: half the memory instructions on the path that the DMA pieces already load.
// X2 (split precision, net_precision 3): the accumulators hold the conv of weights scaled by 2^s (wsc = 2^-s undoes it, exactly),
// and every fp16 output is written as TWO chunks -- hi = half(v), lo = half(v - hi) -- at the [16 hi | 16 lo] positions of the
// channel group (x2_index), which is what the next conv's K = 32 MFMA step consumes.
__device__ __forceinline__ int x2_index(int c) { return ((c >> 4) << 5) + (c & 15); }       // hi position of real channel c; lo = + 16
template <int F, int CT, int NPT, int EPI, bool R16, bool X2 = false>
__device__ __forceinline__ void conv_epilogue_h8(f32x4 (&acc)[CT][NPT], const int (&mrow)[NPT], int M, int co_base, int kq,
                                                 float* __restrict__ out32, _Float16* __restrict__ out16, const float* par, int pstride,
                                                 float wsc, float& amax) {
    // amax: largest |value| this lane rounded to fp16 (range guard: above 65504 the fp16 copy is inf and the next layer computes
    // NaN; the caller raises the network's sticky overflow counter, tg_net_range)
    static_assert(CT % 2 == 0, "tile pairs");
    static_assert(!(X2 && R16), "split precision keeps the f32 residual stream");
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        if (mrow[t] >= M) continue;
#pragma unroll
        for (int u = 0; u < CT / 2; ++u) {
            const int lc = u * 32 + kq * 8, col = co_base + lc;
            f32x4 v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) v[h] = (X2 ? acc[2 * u + h][t] * wsc : acc[2 * u + h][t]) + *reinterpret_cast<const f32x4*>(par + lc + 4 * h);
            h8 o;
            if constexpr (X2) {
                auto store_split = [&](const f32x4 (&w)[2]) {
                    h8 hi, lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float x = w[e >> 2][e & 3]; hi[e] = (_Float16)x; lo[e] = (_Float16)(x - (float)hi[e]); }
#pragma unroll
                    for (int e = 0; e < 8; e += 2) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(w[e >> 2][e & 3]), __builtin_fabsf(w[e >> 2][(e & 3) + 1])));
                    const int c2 = x2_index(col);
                    *reinterpret_cast<h8*>(out16 + h16_index(mrow[t], c2, M)) = hi;
                    *reinterpret_cast<h8*>(out16 + h16_index(mrow[t], c2 + 16, M)) = lo;
                };
                if (EPI == 0 || EPI == 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[h][e] = v[h][e] > 0.f ? v[h][e] : 0.f;
                }
                if (EPI == 0) { store_split(v); continue; }
                if (out32) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(out32 + (size_t)mrow[t] * F + col + 4 * h) = v[h];
                }
                if (out16) {
                    f32x4 w[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(par + pstride + lc + 4 * h);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(par + 2 * pstride + lc + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float y = v[h][e] * sc[e] + sh[e]; w[h][e] = y > 0.f ? y : 0.f; }
                    }
                    store_split(w);
                }
                continue;
            }
            if (EPI == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float w = v[e >> 2][e & 3]; o[e] = (_Float16)(w > 0.f ? w : 0.f); amax = __builtin_fmaxf(amax, w); }
                *reinterpret_cast<h8*>(out16 + h16_index(mrow[t], col, M)) = o;
            } else {
                if (EPI == 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[h][e] = v[h][e] > 0.f ? v[h][e] : 0.f;
                }
                if (out32) {                                             // null: the last block (only its activation feeds the head)
                    if (R16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { o[e] = (_Float16)v[e >> 2][e & 3]; amax = __builtin_fmaxf(amax, __builtin_fabsf(v[e >> 2][e & 3])); }
                        *reinterpret_cast<h8*>(reinterpret_cast<_Float16*>(out32) + h16_index(mrow[t], col, M)) = o;
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; ++h) *reinterpret_cast<f32x4*>(out32 + (size_t)mrow[t] * F + col + 4 * h) = v[h];
                    }
                }
                if (out16) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(par + pstride + lc + 4 * h);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(par + 2 * pstride + lc + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float w = v[h][e] * sc[e] + sh[e]; o[4 * h + e] = (_Float16)(w > 0.f ? w : 0.f); amax = __builtin_fmaxf(amax, w); }
                    }
                    *reinterpret_cast<h8*>(out16 + h16_index(mrow[t], col, M)) = o;
                }
            }
        }
    }
}

// Workgroup = 4 waves = 256 rows x 128 couts (the block index picks the cout half at F=256; wave tile 128 couts x 64 rows = 8x4
// accumulator tiles, 12 ds_read_b128 per 32 MFMAs).  Stage = (32-channel slice, tap); stages are handled in PAIRS per barrier:
// 4-slot weight ring (8 KB per slot, global_load_lds one pair ahead), two 32-channel slab buffers (rows + halo, buffer_load..lds,
// bounds check = zero fill) -- 70 KB of LDS and <= 256 VGPRs, so TWO workgroups share a CU and one computes while the other sits
// in its barrier, its slab wait or its epilogue.  The kernel can walk a (row tile, cout half) list with the next tile's first
// DMAs issued before the current tile's stores (gridDim.x a multiple of 16), but is launched with one block per tile.  Weight traffic: 1.18 MB per 256 rows (~4 TB/s of L2
// reads chip-wide at 1 PFLOP/s).  Measured shares of a 0.88-ms launch (8192 boards, F=256; ablation builds): MFMA loop alone
// 0.44, DMA issue + traffic 0.17, epilogue 0.22, B-fragment addressing + reads 0.09, barriers 0.03, A reads 0.03.
// An 8-wave 256x256 tile with one workgroup per CU (64-channel stages) measured 827 vs 888 TFLOP/s for this shape.
// X2 = split precision (net_precision 3, "f32x3"): the tensors hold every real channel as fp16 hi + lo, a 32-wide channel group
// being [16 real channels hi | the same 16 lo] in activations and weights alike (CIN counts those positions: 2 x the real input
// channels).  One K = 32 MFMA step with the B fragment [a_hi | a_lo] against the A fragment [w_hi | w_hi] gives w_hi*(a_hi + a_lo),
// a second one against [w_lo | w_lo] gives w_lo*(a_hi + a_lo): all four products of (w_hi + w_lo)(a_hi + a_lo) in f32
// accumulation from ONE read of the activations and two reads of the SAME weight tile (lane (j, kq) takes 16-B chunk kq & 1 of
// row j for the hi fragment, 2 + (kq & 1) for the lo one -- no duplicate storage, the swizzle and its conflict-freedom carry
// over).  4 x the MFMA work of the plain fp16 conv at ~22 significand bits per operand, against 16 x for the exact-f32 MFMA.
// With TG_X2_3P (default) only THREE of the four products are formed -- w_lo*a_lo is dropped and the two lo-weight products of a
// pair of stages share one K = 32 step (see the stage loop): 3 x the MFMA work of the plain fp16 conv.
template <int S, int CIN, int F, int EPI, bool R16 = false, bool X2 = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_h2(const _Float16* __restrict__ in, float* __restrict__ out32,
                                                       _Float16* __restrict__ out16, const float* __restrict__ res,
                                                       const _Float16* __restrict__ Ws, const float* __restrict__ bias,
                                                       const float* __restrict__ s2, const float* __restrict__ t2, int M, int nblk,
                                                       const float* __restrict__ wsc_p, unsigned* __restrict__ ovf) {
    constexpr int P = S * S, HALO = S + 1, KC = 32, NW = 4, NPT = 4, TM = 64 * NW, NCO = 128, CT = NCO / 16, COS = F / NCO;
    constexpr int NCHK = KC / 8, RPP = 64 / NCHK;                       // 4 chunks per 64-B row, 16 rows per 1-KB DMA piece
    constexpr int NSL = CIN / KC, NST = NSL * 9, NPAIR = NST / 2, NSLOT = 4;        // CIN input channels (row stride), F output channels
    constexpr int NROW = TM + 2 * HALO;
    constexpr int PLR = (NROW + 63) / 64 * 64;                          // rows of one chunk plane of the slab image (whole 1-KB pieces)
    constexpr int NXP = NCHK * (PLR / 64);                              // slab pieces: 4 planes x PLR/64
    constexpr int NXQ = (NXP + NW - 1) / NW;
    constexpr int WPW = NCO / RPP / NW;                                 // weight pieces per wave per stage (2)
    static_assert(NST % 2 == 0 && CIN % KC == 0 && F % NCO == 0 && NCO % (RPP * NW) == 0, "tile geometry");
    __shared__ __attribute__((aligned(16))) _Float16 xs[2][NCHK * PLR * 8];
    __shared__ __attribute__((aligned(16))) _Float16 ws[NSLOT][NCO * KC];
    __shared__ __attribute__((aligned(16))) float par[3 * NCO];
    __shared__ __attribute__((aligned(16))) float zrow[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    // the COS cout parts of one row tile are blocks b and b+8: same XCD (b % 8), dispatched together, so the slab's second read hits L2
    // Persistent: this workgroup takes blocks bid, bid + gridDim.x, ... (gridDim.x is a multiple of 8*COS, so the cout part co0
    // never changes); blocks whose rows lie past M (the block count is rounded up) are skipped.
    // TG_H2_XCD: XCD x (= b % 8) takes the contiguous row-tile range [x * tpx, (x + 1) * tpx), tpx = nblk / (8 * COS), so neighbouring
    // tiles share their slab halos through one L2 (linear order: consecutive tiles sit on consecutive XCDs and every halo is fetched twice)
    auto tile_m0 = [&](int b) {
        if (TG_H2_XCD) return ((b % 8) * (nblk / (8 * COS)) + b / (8 * COS)) * TM;
        return (COS == 1 ? b : (b / (8 * COS)) * 8 + b % 8) * TM;
    };
    int bid = blockIdx.x;
    const int co0 = COS == 1 ? 0 : ((bid / 8) % COS) * NCO;
    while (bid < nblk && tile_m0(bid) >= M) bid += gridDim.x;
    if (bid >= nblk) return;
    int m0 = tile_m0(bid);
    if (tid < 4) zrow[tid] = 0.f;
    for (int i = tid; i < NCO; i += 256) {
        par[i] = bias[co0 + i]; par[NCO + i] = out16 && s2 ? s2[co0 + i] : 0.f; par[2 * NCO + i] = out16 && t2 ? t2[co0 + i] : 0.f;
    }
    const u32x4 rin = tg_rsrc(in, (unsigned)M * CIN * 2);
    // weight pieces: 16 rows (couts) x 64 B; lane -> row prow, physical chunk pchk, which holds logical chunk pchk ^ swz64(row)
    const int prow = lane / NCHK, pchk = lane % NCHK;
    const int lane_w = prow * KC + (pchk ^ swz64(prow)) * 8;              // halfs, stage-tile rows are KC halfs apart
    // slab pieces: 64 rows x 16 B of one chunk plane, contiguous in memory and in LDS (lane = row)
    auto dma_x = [&](int sl) {
#pragma unroll
        for (int i = 0; i < NXQ; ++i) {
            const int q = wave * NXQ + i;
            if (q < NXP) {                                               // wave-uniform
                const int plane = q / (PLR / 64), rg = q % (PLR / 64);
                // the whole offset must travel in voffset: soffset is not range-checked, and rows outside the tensor must read 0
                // (before its first plane / past its last one: bounds check).  Rows m < 0 or >= M of an inner plane read the
                // neighbouring plane's rows instead -- finite values that only masked taps could ever select
                const int voff = (((sl * NCHK + plane) * M + m0 - HALO + rg * 64) * 8) * 2 + lane * 16;
                tg_dma_buffer(rin, voff, (tg_lds_void*)(&xs[sl & 1][(plane * PLR + rg * 64) * 8]));
            }
        }
    };
    auto dma_w = [&](int g) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int pc = wave * WPW + i;
            const _Float16* wb = Ws + ((size_t)g * F + co0 + pc * RPP) * KC;                       // wave-uniform
            tg_dma_global(wb, lane_w * 2, (tg_lds_void*)(&ws[g % NSLOT][pc * 512]));
        }
    };
    auto prologue = [&]() {                                              // order matters for the counted wait below
        dma_x(0); dma_w(0); dma_w(1);
        if (NSL > 1) dma_x(1);
        dma_w(2); dma_w(3);
    };
    prologue();
    bool first = true;
    const int aoff = j * KC + (((X2 ? (kq & 1) : kq) ^ swz64(j)) << 3);   // A fragment of cout tile ct: + ct*16*KC (X2: the hi half)
    const int aoff_lo = j * KC + (((2 + (kq & 1)) ^ swz64(j)) << 3);      // X2: the lo half of the same rows
    const float wsc = X2 ? wsc_p[0] : 1.f, wsc_inv = X2 ? 1.f / wsc : 1.f;   // powers of two: both exact
    for (;;) {
        unsigned vmask[NPT]; int vrow[NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int m = m0 + (wave * NPT + t) * 16 + j;
            unsigned mk = 0;
            if (m < M) {
                const int p = m % P, x = p % S, y = p / S;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                    if (yy >= 0 && yy < S && xx >= 0 && xx < S) mk |= 1u << tap;
                }
            }
            vmask[t] = mk; vrow[t] = (wave * NPT + t) * 16 + j + HALO;
        }
        auto read_b = [&](f32x4* b, int g) {
            const int sl = g / 9, tap = g % 9;
            const int toff = (tap / 3 - 1) * S + (tap % 3 - 1);
            const _Float16* base = xs[sl & 1];
#pragma unroll
            for (int t = 0; t < NPT; ++t) {
                const int R = vrow[t] + toff;
                const _Float16* src = ((vmask[t] >> tap) & 1) ? base + ((kq * PLR + R) << 3)
                                                              : reinterpret_cast<const _Float16*>(zrow);
                b[t] = *reinterpret_cast<const f32x4*>(src);
            }
        };
        // EPI 1: the accumulators START from the residual (32 loads per wave that fly while the first DMAs land), so the epilogue
        // has no load in it -- a load issued after stores waits for them (one in-order counter): 8 serial round trips per tile
        f32x4 acc[CT][NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) {
            const int m = m0 + (wave * NPT + t) * 16 + j;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (!(EPI == 1 && R16 && ct % 2 == 1)) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};   // (set by its pair's load)
                if (EPI == 1) {
                    // always issued (rows past the batch re-read the last row and are never stored): the number of residual loads in
                    // flight must be the same for every wave, because the wait below counts them.  Paired cout mapping (see
                    // conv_epilogue_h8): one 16-B load covers this lane's 8 couts of the tile pair.
                    const int mc = m < M ? m : M - 1;
                    if (R16) {
                        if (ct % 2 == 0) {
                            const h8 rh = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(res) + h16_index(mc, co0 + (ct >> 1) * 32 + kq * 8, M));
#pragma unroll
                            for (int e = 0; e < 4; ++e) { acc[ct][t][e] = (float)rh[e]; acc[ct + 1][t][e] = (float)rh[4 + e]; }
                        }
                    } else {
                        acc[ct][t] = *reinterpret_cast<const f32x4*>(res + (size_t)mc * F + co0 + (ct >> 1) * 32 + kq * 8 + (ct & 1) * 4);
                        if (X2) acc[ct][t] = acc[ct][t] * wsc_inv;              // the accumulators run in the scaled-weight domain
                    }
                }
            }
        }
        // first tile of a residual-free launch: only the prologue DMAs are in flight, so the wait can leave X(1), W(2), W(3) out.
        // Residual variant: the CT*NPT residual loads are the YOUNGEST operations of the wave (the prologue DMAs and, in a persistent
        // walk, the previous tile's stores are older), so waiting until only they are outstanding releases the first barrier as soon
        // as the DMAs have landed; each accumulator's own load is then waited for by the compiler, in order, right before its first
        // MFMA -- the 0.34-0.68 GB residual read no longer sits in front of the whole tile.
        if (EPI != 1 && first && NSL > 1) vmcnt_uniform<NXQ + 2 * WPW>((NXP - wave * NXQ < 0 ? 0 : NXP - wave * NXQ > NXQ ? NXQ : NXP - wave * NXQ) + 2 * WPW);
        else if (EPI != 1 && first) TG_VMCNT(2 * WPW);
        else if (EPI == 1) TG_VMCNT(R16 ? CT / 2 * NPT : CT * NPT);
        else TG_VMCNT(0);
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TG_BARRIER();
        f32x4 b_cur[NPT], b_next[NPT];
        read_b(b_cur, 0);

#pragma unroll 1                                                         // unrolled, hipcc keeps 4 pairs of address state live and spills
        for (int pp = 0; pp < NPAIR; ++pp) {
          const int g0 = 2 * pp;
          if constexpr (X2 && TG_X2_3P) {
            // THREE partial products per pair of stages instead of four: w_lo * a_lo (2^-22 of a product, below the f32 rounding of the
            // sum) is dropped.  Steps of a pair (s0, s1 = its two stages, same or neighbouring slice, both tiles in the ring):
            //   CT steps  [w_hi(s0) | w_hi(s0)] x [a_hi(s0) | a_lo(s0)]      (as before)
            //   CT steps  [w_hi(s1) | w_hi(s1)] x [a_hi(s1) | a_lo(s1)]      (as before)
            //   CT steps  [w_lo(s0) | w_lo(s1)] x [a_hi(s0) | a_hi(s1)]      -- both lo-weight products in ONE K = 32 step: the B operand is
            //             made in place from the two stages' fragments by v_permlane32_swap (the hi halves live in lanes 0-31), the A
            //             operand takes its lo chunks from tile s0 in lanes 0-31 and from tile s1 in lanes 32-63.
            // 3/4 of the MFMA steps and of the weight-fragment reads; b_cur holds s0's fragments, b_next s1's, and after the swap
            // b_next is free for the next pair's first stage.
            constexpr int NU = 3 * CT, DA = 4;
            const _Float16* const wmix = ws[(g0 + (kq >> 1)) % NSLOT] + aoff_lo;
            auto a_addr = [&](int un) -> const _Float16* {
                const int kind = un / CT, ct = un % CT;
                return kind == 2 ? wmix + ct * 16 * KC : ws[(g0 + kind) % NSLOT] + ct * 16 * KC + aoff;
            };
            f32x4 a[DA];
#pragma unroll
            for (int u = 0; u < DA - 1; ++u) a[u] = *reinterpret_cast<const f32x4*>(a_addr(u));
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + DA - 1 < NU) a[(u + DA - 1) % DA] = *reinterpret_cast<const f32x4*>(a_addr(u + DA - 1));
                if (u == CT / 2) read_b(b_next, g0 + 1);                                  // s1's fragments, needed from step CT on
                if (u == 2 * CT) {                                                        // b_cur <- [a_hi(s0) | a_hi(s1)]
#pragma unroll
                    for (int t = 0; t < NPT; ++t)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float fc = b_cur[t][e], fn = b_next[t][e];     // (scalar copies: a bit_cast of a vector ELEMENT miscompiles)
                            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(fc), __float_as_uint(fn), false, false);
                            b_cur[t][e] = __uint_as_float(r[0]);
                        }
                }
                if (u == 2 * CT + CT / 2 && g0 + 2 < NST) read_b(b_next, g0 + 2);         // the next pair's first stage (its slab is visible)
                const int ct = u % CT;
#pragma unroll
                for (int t = 0; t < NPT; ++t)
                    acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[u % DA]), __builtin_bit_cast(h8, u / CT == 1 ? b_next[t] : b_cur[t]),
                                                                        acc[ct][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < NPT; ++t) b_cur[t] = b_next[t];
          } else {
            constexpr int SPS = X2 ? 2 * CT : CT;                        // MFMA steps per stage (X2: hi and lo fragment of every cout tile)
            constexpr int NU = 2 * SPS, DA = 4;
            // A fragments run DA-1 steps ahead of their MFMAs (a step = 4 MFMAs = 64 cycles; an LDS read under load takes longer)
            auto a_addr = [&](int un) -> const _Float16* {
                const int st = un / SPS, w = un % SPS;
                const int ct = X2 ? w >> 1 : w;
                return ws[(g0 + st) % NSLOT] + ct * 16 * KC + ((X2 && (w & 1)) ? aoff_lo : aoff);
            };
            f32x4 a[DA];
#pragma unroll
            for (int u = 0; u < DA - 1; ++u) a[u] = *reinterpret_cast<const f32x4*>(a_addr(u));
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + DA - 1 < NU) a[(u + DA - 1) % DA] = *reinterpret_cast<const f32x4*>(a_addr(u + DA - 1));
                if (u % SPS == SPS / 2 && g0 + u / SPS + 1 < NST) read_b(b_next, g0 + u / SPS + 1);   // that stage's slab is visible (see below)
                const int ct = X2 ? (u % SPS) >> 1 : u % SPS;
#pragma unroll
                for (int t = 0; t < NPT; ++t)
                    acc[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[u % DA]), __builtin_bit_cast(h8, b_cur[t]),
                                                                        acc[ct][t], 0, 0, 0);
                if (u % SPS == SPS - 1) {
#pragma unroll
                    for (int t = 0; t < NPT; ++t) b_cur[t] = b_next[t];
                }
                __builtin_amdgcn_sched_barrier(0);                      // keep each step's reads where they were issued (hipcc sinks them to their use)
            }
          }
            // every DMA of this wave was issued at least one pair ago: wait for all of them, then meet the other waves.  After the
            // barrier the two slots of this pair are free, and so is the slab buffer of slice sl-1 once stage 9*sl-1 is behind us.
            TG_VMCNT(0);
            TG_BARRIER();
            const int gdone = g0 + 1;                                    // last finished stage
            // the slab of slice sl+1 goes into the buffer slice sl-1 used: issue it right after the pair that finished stage 9*sl-1
            // (it is fenced by the NEXT pair's barrier, i.e. before stage 9*sl+3; first read by the prefetch in stage 9*sl+8)
            {
                const int sl = (gdone + 1) / 9;
                if (sl >= 1 && sl + 1 < NSL && (gdone == 9 * sl - 1 || gdone == 9 * sl)) dma_x(sl + 1);
            }
            if (g0 + 4 < NST) { dma_w(g0 + 4); dma_w(g0 + 5); }
        }
        int mrow[NPT];
#pragma unroll
        for (int t = 0; t < NPT; ++t) mrow[t] = m0 + (wave * NPT + t) * 16 + j;
        if constexpr (X2) {
            // opaque here: computed before the stage loop, the epilogue's 64-bit addresses are spilled across it and reloaded BEHIND the
            // epilogue's stores (scratch reloads queue on the same in-order counter)
            static_assert(NPT == 4, "four row tiles per wave");
            asm volatile("" : "+v"(mrow[0]), "+v"(mrow[1]), "+v"(mrow[2]), "+v"(mrow[3]));
        }
        int Me = M, kqe = kq;                                             // likewise the row count and the lane's chunk index the plane offsets are made of
        if constexpr (X2) { asm volatile("" : "+s"(Me)); asm volatile("" : "+v"(kqe)); }
        // every wave is past the last barrier: ring and slab buffers are free, so the next tile's first loads go out before this
        // tile's results do and land while the epilogue runs
        int nb = bid + gridDim.x;
        while (nb < nblk && tile_m0(nb) >= M) nb += gridDim.x;
        const bool have_next = nb < nblk;
        if (have_next) { bid = nb; m0 = tile_m0(nb); prologue(); }
        float amax = 0.f;
        conv_epilogue_h8<F, CT, NPT, (EPI == 1 ? 2 : EPI), R16, X2>(acc, mrow, Me, co0, kqe, out32, out16, par, NCO, wsc, amax);   // EPI 4: the stem
        // range guard: one compare + ballot per tile and wave, an atomic only when a value left the fp16 range (sticky counter)
        if (__builtin_amdgcn_ballot_w64(!(amax <= 65504.f)) != 0 && lane == 0) atomicAdd(ovf, 1u);
        if (!have_next) break;
    }
}

// stage-ordered fp16 copy of a conv: dst[(slice*9 + tap)][cout][KC] = half(w[tap][cout][slice*KC + c]) for c < cin_src, else 0
// (w is [9][COUT][cin_src]; the destination covers cin_dst >= cin_src channels: the stem pads its 16 input planes to 64)
__global__ __launch_bounds__(256) void k_restage_half(const float* __restrict__ w, _Float16* __restrict__ dst, int COUT, int cin_src,
                                                      int cin_dst, int KC, int pair8) {
    const size_t total = (size_t)9 * COUT * cin_dst;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % KC);
        const size_t r = i / KC;
        int co = (int)(r % COUT);
        const int st = (int)(r / COUT);
        const int sl = st / 9, tap = st % 9;
        const int ci = sl * KC + c;
        // pair8 (k_conv3x3_h2): destination row r of a 32-row group holds physical cout ((r & 15) >> 2) * 8 + ((r >> 4) & 1) * 4 + (r & 3)
        if (pair8) co = (co & ~31) | ((((co & 15) >> 2) << 3) + (((co >> 4) & 1) << 2) + (co & 3));
        dst[i] = ci < cin_src ? (_Float16)w[((size_t)tap * COUT + co) * cin_src + ci] : (_Float16)0.f;
    }
}

// Split precision (net_precision 3).  k_absmax: largest |w| of one conv as float bits (non-negative floats order like unsigned ints).
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ w, size_t n, unsigned* __restrict__ out) {
    unsigned m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned b = __float_as_uint(w[i]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
// stage-ordered split copy: dst[(slice*9 + tap)][cout][32] = [16 channels hi | the same 16 lo] of w * 2^s, slice = 16 real input
// channels (zero beyond cin_src), hi = half(v), lo = half(v - hi); s puts the conv's largest weight in [2^14, 2^15), so that the
// lo halves of all but negligible weights are normal fp16 numbers; wsc_out = 2^-s for the epilogue.  pair8 as in k_restage_half.
__global__ __launch_bounds__(256) void k_restage_split(const float* __restrict__ w, _Float16* __restrict__ dst, float* __restrict__ wsc_out,
                                                       const unsigned* __restrict__ maxbits, int COUT, int cin_src, int nsl_dst, int pair8) {
    const unsigned mb = maxbits[0];
    int sh = mb ? 14 - ((int)((mb >> 23) & 0xffu) - 127) : 0;
    sh = sh < -60 ? -60 : sh > 60 ? 60 : sh;
    const float scale = ldexpf(1.f, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) wsc_out[0] = ldexpf(1.f, -sh);
    const size_t total = (size_t)nsl_dst * 9 * COUT * 32;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i & 31);
        const size_t r = i >> 5;
        int co = (int)(r % COUT);
        const int st = (int)(r / COUT);
        const int sl = st / 9, tap = st % 9;
        const int ci = sl * 16 + (k & 15);
        if (pair8) co = (co & ~31) | ((((co & 15) >> 2) << 3) + (((co >> 4) & 1) << 2) + (co & 3));
        const float v = ci < cin_src ? w[((size_t)tap * COUT + co) * cin_src + ci] * scale : 0.f;
        const _Float16 hi = (_Float16)v;
        dst[i] = (k & 16) ? (_Float16)(v - (float)hi) : hi;
    }
}

// stage-ordered f32 copy of an F->F conv for k_conv3x3_sg: dst[(slice*9 + tap)][cout][16] = w[tap][cout][slice*16 + c]
__global__ __launch_bounds__(256) void k_restage_f32(const float* __restrict__ w, float* __restrict__ dst, int F) {
    const size_t total = (size_t)9 * F * F;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 15);
        const size_t r = i >> 4;
        const int co = (int)(r % F);
        const int st = (int)(r / F);
        dst[i] = w[((size_t)(st % 9) * F + co) * F + (st / 9) * 16 + c];
    }
}

// network input for the fp16 stem: obs f32 [rows][C][P] (planes of 0/1, exact in fp16) -> x0h, 64 channels, chunk-major (h16_index)
template <int S>
__global__ __launch_bounds__(256) void k_obs_to_rows_h(const float* __restrict__ obs, _Float16* __restrict__ x0, int rows, int C) {
    constexpr int P = S * S;
    const size_t total = (size_t)rows * P * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i & 63);
        const size_t m = i >> 6;
        const int p = (int)(m % P);
        const size_t r = m / P;
        x0[h16_index((int)m, c, rows * P)] = c < C ? (_Float16)obs[(r * C + c) * P + p] : (_Float16)0.f;
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_bits_to_rows_h(const uint32_t* __restrict__ bits, const int32_t* __restrict__ row_slot,
                                                        _Float16* __restrict__ x0, int rows, int C, int W) {
    // one thread = one 16-B chunk (8 channels of one row) of the chunk-major tensor; consecutive threads walk a chunk plane, so the
    // stores are contiguous (round 2 wrote 2-B elements scattered over the planes: 97 us per 16 k leaves, now ~25)
    constexpr int P = S * S;
    const size_t M = (size_t)rows * P, total = M * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cp = (int)(i / M);
        const size_t m = i - (size_t)cp * M;
        h8 o = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (cp * 8 < C) {
            const size_t r = m / P;
            const int p = (int)(m - r * P);
            const uint32_t* base = bits + (size_t)row_slot[r] * W;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = cp * 8 + e, k = c * P + p;
                if (c < C) o[e] = (_Float16)(float)((base[k >> 5] >> (k & 31)) & 1u);
            }
        }
        *reinterpret_cast<h8*>(x0 + (i << 3)) = o;                       // h16_index(m, cp * 8, M) = (cp * M + m) * 8
    }
}

// fp16 head conv (F -> 16 couts: value/ownership and policy convs with their BNs folded, model.py:65-76): out[m][16] = relu(acc +
// bias) in f32 for k_heads.  One 16-cout tile makes this an LDS-read-bound kernel (one B fragment per MFMA), not an MFMA-bound
// one: the 9 A fragments of a 32-channel slice live in registers, slabs and the slice's 9-KB weight block are double-buffered
// LDS-DMA targets, one barrier per slice.  Same slab image, swizzle and masking as k_conv3x3_h2.
template <int S, int F>
__global__ __launch_bounds__(256, 2) void k_head_h(const _Float16* __restrict__ in, float* __restrict__ out,
                                                   const _Float16* __restrict__ Wh, const float* __restrict__ bias, int M) {
    constexpr int P = S * S, HALO = S + 1, KC = 32, NW = 4, NPT = 4, TM = 64 * NW, NCHK = 4, RPP = 16;
    constexpr int NSL = F / KC, NROW = TM + 2 * HALO, PLR = (NROW + 63) / 64 * 64, NXP = NCHK * (PLR / 64), NXQ = (NXP + NW - 1) / NW;
    static_assert(RPP == 16, "weight pieces are 16 couts x 64 B");
    __shared__ __attribute__((aligned(16))) _Float16 xs[2][NCHK * PLR * 8];      // chunk-major slab image (see h16_index)
    __shared__ __attribute__((aligned(16))) _Float16 wsl[2][9 * 16 * KC];
    __shared__ __attribute__((aligned(16))) float zrow[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.x * TM;
    if (tid < 4) zrow[tid] = 0.f;
    const u32x4 rin = tg_rsrc(in, (unsigned)M * F * 2);
    const int prow = lane / NCHK, pchk = lane % NCHK;
    const int lane_w = prow * KC + (pchk ^ swz64(prow)) * 8;
    auto dma = [&](int sl) {                                              // slab and weight block of slice sl
#pragma unroll
        for (int i = 0; i < NXQ; ++i) {
            const int q = wave * NXQ + i;
            if (q < NXP) {                                               // wave-uniform: a piece = 64 rows x 16 B of one chunk plane
                const int plane = q / (PLR / 64), rg = q % (PLR / 64);
                const int voff = (((sl * NCHK + plane) * M + m0 - HALO + rg * 64) * 8) * 2 + lane * 16;
                tg_dma_buffer(rin, voff, (tg_lds_void*)(&xs[sl & 1][(plane * PLR + rg * 64) * 8]));
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            if (tap % NW == wave)                                        // one 1-KB piece per tap: 16 couts x 32 channels
                tg_dma_global(Wh + ((size_t)sl * 9 + tap) * 16 * KC, lane_w * 2, (tg_lds_void*)(&wsl[sl & 1][tap * 512]));
    };
    unsigned vmask[NPT]; int vrow[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int m = m0 + (wave * NPT + t) * 16 + j;
        unsigned mk = 0;
        if (m < M) {
            const int p = m % P, x = p % S, y = p / S;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
                if (yy >= 0 && yy < S && xx >= 0 && xx < S) mk |= 1u << tap;
            }
        }
        vmask[t] = mk; vrow[t] = (wave * NPT + t) * 16 + j + HALO;
    }
    const int aoff = j * KC + ((kq ^ swz64(j)) << 3);
    const int nx_mine = NXP - wave * NXQ < 0 ? 0 : NXP - wave * NXQ > NXQ ? NXQ : NXP - wave * NXQ;
    const int cnt_mine = nx_mine + (wave == 0 ? 3 : 2);                  // DMA instructions this wave issues per slice (taps w, w+4, w+8 < 9)
    f32x4 acc[NPT];
#pragma unroll
    for (int t = 0; t < NPT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dma(0);
    if (NSL > 1) dma(1);
#pragma unroll 1
    for (int sl = 0; sl < NSL; ++sl) {
        // slice sl must have landed; the only younger DMAs of this wave are its pieces of slice sl+1
        if (sl + 1 < NSL) vmcnt_uniform<NXQ + 3>(cnt_mine); else TG_VMCNT(0);
        TG_BARRIER();
        f32x4 a[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) a[tap] = *reinterpret_cast<const f32x4*>(wsl[sl & 1] + tap * 512 + aoff);
        const _Float16* base = xs[sl & 1];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3 - 1) * S + (tap % 3 - 1);
#pragma unroll
            for (int t = 0; t < NPT; ++t) {
                const int R = vrow[t] + toff;
                const _Float16* src = ((vmask[t] >> tap) & 1) ? base + ((kq * PLR + R) << 3) : reinterpret_cast<const _Float16*>(zrow);
                const f32x4 b = *reinterpret_cast<const f32x4*>(src);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[tap]), __builtin_bit_cast(h8, b), acc[t], 0, 0, 0);
            }
        }
        TG_BARRIER();                                     // every wave is done with buffers sl & 1
        if (sl + 2 < NSL) dma(sl + 2);
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + kq * 4);
#pragma unroll
    for (int t = 0; t < NPT; ++t) {
        const int m = m0 + (wave * NPT + t) * 16 + j;
        if (m < M) {
            f32x4 v = acc[t] + bv;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            *reinterpret_cast<f32x4*>(out + (size_t)m * 16 + kq * 4) = v;
        }
    }
}

// ---- head conv (F -> the 6 real couts of the 16-wide head tensor: value/ownership conv 2 + policy conv 4, model.py:65-76) --------
// As an implicit GEMM a 6-cout conv fills 6 of the 16 columns of an MFMA tile: 0.44 ms per 16 k leaves at 9x9 / F = 128, a third of
// it useful.  Here it is "GEMM, then col2im": every INPUT row m is multiplied once by all 9 taps' weights,
//     D[m][tap*6 + co] = sum_c act(x[m][c]) * W[tap][co][c]          (54 useful of 64 columns = 4 MFMA N-tiles),
// and an output is out[p][co] = relu(bias[co] + sum_tap D[p + off(tap)][tap*6 + co]) over the taps that stay on the board.  MFMAs
// per output row: 4 N-tiles x F/4 steps x (1 + halo share) instead of 9 taps x F/4 -- 10.7 vs 18 per row at 9x9 / F = 128.
// No LDS operand staging and no barrier in the GEMM phase: a wave keeps its B fragments (the whole 64 x F weight matrix at F <= 128,
// two N-tiles per pass above) in registers and streams A fragments (16 rows x 64 B of the row-major stream) straight from L2, the
// next row tile's in flight while this one computes; D goes to LDS (row stride 68 floats: the four 16-lane groups of a store hit
// disjoint banks) for the col2im pass.  Workgroups walk the tile list (at F <= 128 the weights are fetched once per wave).  PRO: the
// input is the residual stream and relu(bn_end(.)) is applied to the A fragments in registers (rows outside the batch stay zero).
template <int S, int F, bool PRO>
__global__ __launch_bounds__(256, 2) void k_head_gemm(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ W2,
                                                      const float* __restrict__ bias, const float* __restrict__ ps,
                                                      const float* __restrict__ pt, int M, int ntiles) {
    constexpr int P = S * S, HALO = S + 1, NTW = S == 9 ? 3 : 4, NROW = 64 * NTW, TM = NROW - 2 * HALO, LDW = 68, NK = F / 16;
    constexpr int NTP = F <= 128 ? 4 : 2, NPASS = 4 / NTP;               // N-tiles whose B fragments a wave holds at once
    constexpr bool DBUF = F <= 128;                                     // room to keep the next row tile's A fragments in flight
    __shared__ __attribute__((aligned(16))) float dl[NROW * LDW];
    __shared__ __attribute__((aligned(16))) float pro[PRO ? 2 * F : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    if (PRO) for (int i = tid; i < F; i += 256) { pro[i] = ps[i]; pro[F + i] = pt[i]; }
    __syncthreads();
    auto load_a = [&](f32x4* a, int row) {                              // A fragments of one row tile: x[row][k16*16 + kq*4 .. +3]
        const bool ok = row >= 0 && row < M;
        const float* src = in + (size_t)(ok ? row : 0) * F + kq * 4;
#pragma unroll
        for (int k = 0; k < NK; ++k) a[k] = ok ? *reinterpret_cast<const f32x4*>(src + k * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 bw[NTP][NK];                                                  // B fragments: W2[n = nt*16 + j][k16*16 + kq*4 .. +3]
    auto load_b = [&](int pass) {
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int k = 0; k < NK; ++k)
                bw[nt][k] = *reinterpret_cast<const f32x4*>(W2 + (size_t)((pass * NTP + nt) * 16 + j) * F + k * 16 + kq * 4);
    };
    if (NPASS == 1) load_b(0);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TM, r0 = m0 - HALO;
#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            if (NPASS > 1) load_b(pass);
            f32x4 a[DBUF ? 2 : 1][NK];
            load_a(a[0], r0 + (wave * NTW) * 16 + j);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const int rt = wave * NTW + i, row = r0 + rt * 16 + j;
                f32x4* cur = a[DBUF ? (i & 1) : 0];
                if (DBUF && i + 1 < NTW) load_a(a[(i + 1) & 1], row + 16);
                const bool ok = row >= 0 && row < M;
                f32x4 acc[NTP];
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                int po = kq * 4;                                        // opaque per row tile: the scale / shift reads stay inside the loop
                asm volatile("" : "+v"(po));                            // (hoisted they are 2 * F / 4 registers per lane: spills at F = 256)
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    f32x4 v = cur[k];
                    if (PRO) {
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(pro + k * 16 + po), sh = *reinterpret_cast<const f32x4*>(pro + F + k * 16 + po);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { float u = v[e] * sc[e] + sh[e]; u = u > 0.f ? u : 0.f; v[e] = ok ? u : 0.f; }
                    }
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[s4], bw[nt][k][s4], acc[nt], 0, 0, 0);
                }
                // D tile: row = A's M index = kq*4 + r, column = B's N index = j
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dl[(rt * 16 + kq * 4 + r) * LDW + (pass * NTP + nt) * 16 + j] = acc[nt][r];
                if (!DBUF && i + 1 < NTW) load_a(a[0], row + 16);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < TM * 8; idx += 256) {
            const int o = idx >> 3, co = idx & 7, m = m0 + o;
            if (m >= M) continue;
            float v = 0.f;
            if (co < 6) {
                const int p = m % P, x = p % S, y = p / S;
                v = bias[co];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    if (y + dy >= 0 && y + dy < S && x + dx >= 0 && x + dx < S) v += dl[(o + HALO + dy * S + dx) * LDW + tap * 6 + co];
                }
                v = v > 0.f ? v : 0.f;
            }
            out[(size_t)m * 16 + co] = v;
        }
        __syncthreads();
    }
}

// the head conv's weights for k_head_gemm: dst[n = tap*6 + co][c] = w[tap][co][c] for co < 6 (w is [9][16][F]), rows 54..63 zero
__global__ __launch_bounds__(256) void k_restage_head(const float* __restrict__ w, float* __restrict__ dst, int F) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 64 * F; i += gridDim.x * 256) {
        const int n = i / F, c = i % F;
        dst[i] = n < 54 ? w[((size_t)(n / 6) * 16 + n % 6) * F + c] : 0.f;
    }
}

// The same head conv for the split-precision chain (net_precision 3, F = 128): the GEMM runs on v_mfma_f32_16x16x32_f16 with the
// activation rows taken from the split chunk-major conv input the last residual block wrote (relu(bn_end(x)) as fp16 hi | lo:
// a lane's A fragment of a 16-channel group is ONE 16-B load, chunk plane g*4 + kq) and the weights as [w_hi | w_hi] and
// [w_lo | w_lo] fragments in registers (dst of k_restage_head_split, scaled by a power of two): two K = 32 steps per (group,
// N-tile) give all four partial products of (w_hi + w_lo)(a_hi + a_lo) -- a quarter of the f32 kernel's MFMA time, so the kernel
// is bound by the 512 B per row it reads.  A wave holds the B fragments of two N-tiles (128 registers) for the whole kernel;
// D through LDS and col2im as in k_head_gemm, the power-of-two scale undone there (exact).
template <int S, int F>
__global__ __launch_bounds__(256, 2) void k_head_gemm_x2(const _Float16* __restrict__ in, float* __restrict__ out, const _Float16* __restrict__ W2,
                                                         const float* __restrict__ wsc_p, const float* __restrict__ bias, int M, int ntiles) {
    constexpr int P = S * S, HALO = S + 1, NTW = S == 9 ? 3 : 4, NROW = 64 * NTW, TM = NROW - 2 * HALO, LDW = 68, NG = F / 16;
    constexpr int NTP = 2;                                                
    __shared__ __attribute__((aligned(16))) float dl[NROW * LDW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kq = lane >> 4;
    const float wsc = wsc_p[0];
    auto load_a = [&](f32x4* a, int row) {                              // group g: 8 halfs = chunk plane g*4 + kq of row `row`
        const bool ok = row >= 0 && row < M;
        const _Float16* src = in + ((size_t)kq * M + (ok ? row : 0)) * 8;
#pragma unroll
        for (int g = 0; g < NG; ++g) a[g] = ok ? *reinterpret_cast<const f32x4*>(src + (size_t)g * 4 * M * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 bh[NTP][NG], bl[NTP][NG];                                     // B fragments of column n = nt*16 + j: hi and lo halves
    auto load_b = [&](int pass) {
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const _Float16* w = W2 + ((size_t)((pass * NTP + nt) * 16 + j) * NG + g) * 32;
                bh[nt][g] = *reinterpret_cast<const f32x4*>(w + (kq & 1) * 8);
                bl[nt][g] = *reinterpret_cast<const f32x4*>(w + 16 + (kq & 1) * 8);
            }
    };
    // waves 0, 1 take N-tiles {0, 1}, waves 2, 3 take {2, 3}: a wave's B fragments are loaded ONCE per kernel; the two waves of a
    // pair split the row tiles of a tile between them (every row tile is read by one wave of each pair)
    const int nh = wave >> 1, wr = wave & 1;
    load_b(nh);
    constexpr int RTW = 2 * NTW;                                        // row tiles per wave
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TM, r0 = m0 - HALO;
        {
            const int pass = nh;
            f32x4 cur[NG];                                              // one buffer: the next row tile is requested right behind this one's
            load_a(cur, r0 + (wr * RTW) * 16 + j);                      // MFMAs and lands while its D tile goes to LDS (128 registers are
#pragma unroll 1                                                         // the weights': a second A buffer spills)
            for (int i = 0; i < RTW; ++i) {
                const int rt = wr * RTW + i, row = r0 + rt * 16 + j;
                f32x4 acc[NTP];
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int nt = 0; nt < NTP; ++nt) {
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, cur[g]), __builtin_bit_cast(h8, bh[nt][g]), acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, cur[g]), __builtin_bit_cast(h8, bl[nt][g]), acc[nt], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < RTW) load_a(cur, row + 16);
                // D tile: row = A's M index = kq*4 + r, column = B's N index = j
#pragma unroll
                for (int nt = 0; nt < NTP; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dl[(rt * 16 + kq * 4 + r) * LDW + (pass * NTP + nt) * 16 + j] = acc[nt][r];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < TM * 8; idx += 256) {
            const int o = idx >> 3, co = idx & 7, m = m0 + o;
            if (m >= M) continue;
            float v = 0.f;
            if (co < 6) {
                const int p = m % P, x = p % S, y = p / S;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    if (y + dy >= 0 && y + dy < S && x + dx >= 0 && x + dx < S) v += dl[(o + HALO + dy * S + dx) * LDW + tap * 6 + co];
                }
                v = v * wsc + bias[co];
                v = v > 0.f ? v : 0.f;
            }
            out[(size_t)m * 16 + co] = v;
        }
        __syncthreads();
    }
}

// its weights: dst[n = tap*6 + co][group of 16 channels][w_hi 16 | w_lo 16] of w * 2^s (w is [9][16][F]; rows 54..63 zero; s puts the
// largest weight in [2^14, 2^15) like k_restage_split); wsc_out = 2^-s
__global__ __launch_bounds__(256) void k_restage_head_split(const float* __restrict__ w, _Float16* __restrict__ dst, float* __restrict__ wsc_out,
                                                            const unsigned* __restrict__ maxbits, int F) {
    const unsigned mb = maxbits[0];
    int sh = mb ? 14 - ((int)((mb >> 23) & 0xffu) - 127) : 0;
    sh = sh < -60 ? -60 : sh > 60 ? 60 : sh;
    const float scale = ldexpf(1.f, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) wsc_out[0] = ldexpf(1.f, -sh);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 64 * F; i += gridDim.x * 256) {
        const int n = i / F, c = i % F, g = c >> 4, e = c & 15;
        const float v = n < 54 ? w[((size_t)(n / 6) * 16 + n % 6) * F + c] * scale : 0.f;
        const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
        _Float16* d = dst + ((size_t)n * (F / 16) + g) * 32;
        d[e] = hi; d[16 + e] = lo;
    }
}

// Self_Attention core (model.py:301-315) for one board per workgroup, after the fused q/k/v 1x1 projection:
//   energy[i][j] = q_i . k_j ; attention = softmax_j(energy) ; out[:, j] = sum_i v[:, i] * attention[i][j]   (note: summed over
//   the softmaxed ROW index i, exactly as torch.bmm(proj_value, attention) does)
//   y = relu(bn(gamma * out + x))            -- x optionally pre-activated with relu(x*ps+pt) (policy head, model.py:94,106)
template <int S, int F>
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ qkv, const float* __restrict__ xin,
                                                   float* __restrict__ out, const float* __restrict__ gamma,
                                                   const float* __restrict__ bs, const float* __restrict__ bt,
                                                   const float* __restrict__ ps, const float* __restrict__ pt) {
    constexpr int P = S * S, FQ = F / 4, W = 2 * FQ + F;
    extern __shared__ float sm[];
    float* q = sm;                    // [P][W] rows: q | k | v
    float* at = sm + P * W;           // [P][P+1]
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* src = qkv + (size_t)b * P * W;
    for (int i = tid; i < P * W / 4; i += 256) reinterpret_cast<f32x4*>(q)[i] = reinterpret_cast<const f32x4*>(src)[i];
    __syncthreads();
    for (int e = tid; e < P * P; e += 256) {
        const int i = e / P, j = e % P;
        const float* qi = q + i * W; const float* kj = q + j * W + FQ;
        float a = 0.f;
        for (int c = 0; c < FQ; ++c) a = fmaf(qi[c], kj[c], a);
        at[i * (P + 1) + j] = a;
    }
    __syncthreads();
    for (int i = tid; i < P; i += 256) {
        float* r = at + i * (P + 1);
        float mx = r[0];
        for (int j = 1; j < P; ++j) mx = r[j] > mx ? r[j] : mx;
        float sum = 0.f;
        for (int j = 0; j < P; ++j) { float e = expf(r[j] - mx); r[j] = e; sum += e; }
        const float inv = 1.f / sum;
        for (int j = 0; j < P; ++j) r[j] *= inv;
    }
    __syncthreads();
    const float g = gamma[0];
    for (int e = tid; e < P * F; e += 256) {
        const int j = e / F, c = e % F;
        float a = 0.f;
        for (int i = 0; i < P; ++i) a = fmaf(q[i * W + 2 * FQ + c], at[i * (P + 1) + j], a);
        float x = xin[((size_t)b * P + j) * F + c];
        if (ps) { x = x * ps[c] + pt[c]; x = x > 0.f ? x : 0.f; }
        float y = (g * a + x) * bs[c] + bt[c];
        out[((size_t)b * P + j) * F + c] = y > 0.f ? y : 0.f;
    }
}

// one butterfly step of a reduction over the 16 lanes of a DPP row (= the 16 columns j of one lane group kq), without LDS traffic:
// steps 0, 1 exchange within quads (lane ^ 1, lane ^ 2); steps 2, 3 mirror the half row / the row, which pairs quad with quad and
// half with half -- the values are uniform within those by then, so the result equals the xor butterfly's, in the same order
__device__ __forceinline__ float row16_step(float v, int step) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (step) {
        case 0: r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false); break;      // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false); break;      // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false); break;     // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false); break;    // row_mirror
    }
    return __builtin_bit_cast(float, r);
}
// The same Self_Attention core on the matrix cores (F = 64, 128, 256 at 9x9): ONE WAVE PER BOARD, everything between the q/k/v
// projection and the block's output stays in registers.
//   GEMM 1  energy[i][j] = sum_c q[i][c] k[j][c]        (P x P x F/4; 6 x 6 tiles of 16 x 16, P = 81 padded to 96)
//   softmax over j, row by row, on the accumulator tiles themselves (a row lives in 16 lanes x 6 tiles: in-register max/sum
//           + 4 xor-shuffles); padded columns are excluded, padded rows zeroed
//   GEMM 2  out[c][j] = sum_i v[i][c] attention[i][j]   (F x P x P).  The D layout of a 16x16x4 MFMA tile (column = lane & 15,
//           row = 4*(lane >> 4) + r) IS the B-operand layout with k = 4*(lane >> 4) + s, so the softmaxed tiles feed GEMM 2
//           without leaving their registers; only v is loaded (each element once, 64-B runs).  i = 81 needs 21 k-steps, not 24.
//   y = relu(bn(gamma * out + x)), written row-major (the residual stream) and, when the next layer is a residual block of the
//   DMA-fed chain, also as its pre-activated slice-major input relu(bn1_next(y)) (out2).
// MFMA work per board: 288 + 1008 instructions of 16x16x4 (F = 128) = 1.3 % of an F->F conv's; the scalar-FMA kernel above took
// 8.3 ms per 16384 boards (3 x an F->F conv), this one is bounded by its 97 KB of q/k/v/x/y traffic per board.
#ifndef TG_ATT_OCC
#define TG_ATT_OCC 1
#endif
// X2O: out2 is the split-precision chain's conv input instead -- fp16 hi + lo of relu(bn1_next(y)), chunk-major (x2_index).
template <int S, int F, int CP, bool X2O = false>
__global__ __launch_bounds__(256, TG_ATT_OCC) void k_attention_mfma(const float* __restrict__ qkv, const float* __restrict__ xin,
                                                        float* __restrict__ out, float* __restrict__ out2,
                                                        const float* __restrict__ gamma, const float* __restrict__ bs,
                                                        const float* __restrict__ bt, const float* __restrict__ ps,
                                                        const float* __restrict__ pt, const float* __restrict__ s2,
                                                        const float* __restrict__ t2, int rows, unsigned* __restrict__ ovf = nullptr) {
    constexpr int P = S * S, FQ = F / 4, W = 2 * FQ + F, NT = (P + 15) / 16, CT = F / 16, NSUB = FQ / 16;
    static_assert(P <= 96 && FQ % 16 == 0 && CT % CP == 0, "attention tile geometry");
    float amax = 0.f;                                                                // X2O: range guard of the fp16 copies (see conv_epilogue_h8)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= rows) return;
    const int j = lane & 15, kq = lane >> 4;
    const float* base = qkv + (size_t)b * P * W;
    const int M = rows * P;
    // ---- GEMM 1 ----
    f32x4 e[NT][NT];
#pragma unroll
    for (int tm = 0; tm < NT; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) e[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        // padded rows (>= P) repeat the last one; their columns / rows are masked below.  k fragments of the slice stay resident,
        // q fragments stream through one register quad (one ahead), so the 36 energy tiles dominate the register budget
        auto row_of = [&](int t) { return t * 16 + j < P ? t * 16 + j : P - 1; };
        f32x4 kb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) kb[t] = *reinterpret_cast<const f32x4*>(base + (size_t)row_of(t) * W + FQ + sub * 16 + kq * 4);
        f32x4 qa = *reinterpret_cast<const f32x4*>(base + (size_t)row_of(0) * W + sub * 16 + kq * 4);
#pragma unroll
        for (int tm = 0; tm < NT; ++tm) {
            f32x4 qn = qa;
            if (tm + 1 < NT) qn = *reinterpret_cast<const f32x4*>(base + (size_t)row_of(tm + 1) * W + sub * 16 + kq * 4);
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
                    e[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s4], kb[tn][s4], e[tm][tn], 0, 0, 0);
            qa = qn;
        }
    }
    // ---- softmax over j (columns) for every row i = tm*16 + kq*4 + r ----
#pragma unroll
    for (int tm = 0; tm < NT; ++tm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) if (tn * 16 + j < P) mx = e[tm][tn][r] > mx ? e[tm][tn][r] : mx;
#pragma unroll
            for (int o = 0; o < 4; ++o) { const float t = row16_step(mx, o); mx = t > mx ? t : mx; }   // DPP: no LDS round trips
            float sum = 0.f;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const float v = tn * 16 + j < P ? __expf(e[tm][tn][r] - mx) : 0.f;        // v_exp_f32: ~1e-7 relative, tolerance is 1e-3
                e[tm][tn][r] = v; sum += v;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) sum += row16_step(sum, o);
            const float inv = (tm * 16 + kq * 4 + r < P) ? 1.f / sum : 0.f;     // rows past the board contribute nothing to GEMM 2
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) e[tm][tn][r] *= inv;
        }
    }
    // ---- GEMM 2 + epilogue, CP channel tiles per pass ----
    // One wave per SIMD (the 36 energy tiles alone are 144 registers), so latency is hidden inside the wave: ALL v operands of a
    // pass (NK k-steps x CP tiles, 4 B each) are requested while the previous pass computes -- with one step of lead the kernel
    // sat on load latency (84 dependent round trips per board, 1.14 ms per 16384 boards).
    const float g = gamma[0];
    const float* vb = base + 2 * FQ;
    constexpr int LASTS = P - (NT - 1) * 16, NS_LAST = LASTS < 4 ? LASTS : 4, NK = (NT - 1) * 4 + NS_LAST, NPASS = CT / CP;
    // k-step kk covers rows i = (kk/4)*16 + kq*4 + kk%4; of the last 16-row block only the steps that touch a row < P exist
    float av[2][NK][CP];
    auto load_pass = [&](int pass, float (*dst)[CP]) {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int i = (kk >> 2) * 16 + kq * 4 + (kk & 3);
            const int ic = i < P ? i : P - 1;                                         // the matching attention rows are zero
#pragma unroll
            for (int cp = 0; cp < CP; ++cp) dst[kk][cp] = vb[(size_t)ic * W + (pass * CP + cp) * 16 + j];
        }
    };
    load_pass(0, av[0]);
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int c0 = pass * CP;
        if (pass + 1 < NPASS) load_pass(pass + 1, av[(pass + 1) & 1]);
        f32x4 acc[CP][NT];
#pragma unroll
        for (int cp = 0; cp < CP; ++cp)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[cp][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the pass's residual rows are requested BEFORE its MFMAs: they queue behind the previous pass's stores (vmcnt is in order)
        // and have the whole output GEMM to get past them; requested in the epilogue they were waited for at once
        f32x4 x[CP][NT];
#pragma unroll
        for (int cp = 0; cp < CP; ++cp)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const int p = tn * 16 + j < P ? tn * 16 + j : P - 1;
                x[cp][tn] = *reinterpret_cast<const f32x4*>(xin + (size_t)(b * P + p) * F + (c0 + cp) * 16 + kq * 4);
            }
        // ... and so are the pass's per-channel parameters (at F = 256 the split-output variant has no registers left for that:
        // there the parameters are fetched behind the MFMAs, as before)
        constexpr bool PARAMS_EARLY = !(X2O && F > 128);
        f32x4 vbs[CP], vbt[CP], vps[CP], vpt[CP], vs2[CP], vt2[CP];
        auto load_params = [&]() {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int c = (c0 + cp) * 16 + kq * 4;
            vbs[cp] = *reinterpret_cast<const f32x4*>(bs + c); vbt[cp] = *reinterpret_cast<const f32x4*>(bt + c);
            vps[cp] = f32x4{1.f, 1.f, 1.f, 1.f}; vpt[cp] = f32x4{0.f, 0.f, 0.f, 0.f}; vs2[cp] = vps[cp]; vt2[cp] = vpt[cp];
            if (ps) { vps[cp] = *reinterpret_cast<const f32x4*>(ps + c); vpt[cp] = *reinterpret_cast<const f32x4*>(pt + c); }
            if (out2) { vs2[cp] = *reinterpret_cast<const f32x4*>(s2 + c); vt2[cp] = *reinterpret_cast<const f32x4*>(t2 + c); }
        }
        };
        if constexpr (PARAMS_EARLY) load_params();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
#pragma unroll
            for (int cp = 0; cp < CP; ++cp)
#pragma unroll
                for (int tn = 0; tn < NT; ++tn)
                    acc[cp][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pass & 1][kk][cp], e[kk >> 2][tn][kk & 3], acc[cp][tn], 0, 0, 0);
        // D tile: row = channel (c0+cp)*16 + kq*4 + r, column = position tn*16 + j
        if constexpr (!PARAMS_EARLY) load_params();
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int c = (c0 + cp) * 16 + kq * 4;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const int p = tn * 16 + j;
                if (p >= P) continue;
                const int m = b * P + p;
                f32x4 y, u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float xv = x[cp][tn][q];
                    if (ps) { xv = xv * vps[cp][q] + vpt[cp][q]; xv = xv > 0.f ? xv : 0.f; }
                    const float w = (g * acc[cp][tn][q] + xv) * vbs[cp][q] + vbt[cp][q];
                    y[q] = w > 0.f ? w : 0.f;
                    const float z = y[q] * vs2[cp][q] + vt2[cp][q];
                    u[q] = z > 0.f ? z : 0.f;
                }
                *reinterpret_cast<f32x4*>(out + (size_t)m * F + c) = y;
                if constexpr (X2O) {
                    if (out2) {
                        h4 hi, lo;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { hi[q] = (_Float16)u[q]; lo[q] = (_Float16)(u[q] - (float)hi[q]); amax = __builtin_fmaxf(amax, u[q]); }
                        _Float16* const o16 = reinterpret_cast<_Float16*>(out2);
                        const int c2 = x2_index(c);
                        *reinterpret_cast<h4*>(o16 + h16_index(m, c2, M)) = hi;
                        *reinterpret_cast<h4*>(o16 + h16_index(m, c2 + 16, M)) = lo;
                    }
                } else {
                    if (out2) *reinterpret_cast<f32x4*>(out2 + f32_sm_index(m, c, M)) = u;
                }
            }
        }
    }
    if constexpr (X2O) {
        if (__builtin_amdgcn_ballot_w64(!(amax <= 65504.f)) != 0 && lane == 0) atomicAdd(ovf, 1u);
    }
}

// Self_Attention as ONE kernel for the split-precision chain (net_precision 3, F = 128, 9x9): the q/k/v 1x1 projection is computed
// inside the per-board core, on the fp16 matrix cores in split precision, and never goes through memory (the two-kernel form
// moves 62 KB of q|k|v per board out and back in: at 16 k boards the pair took 0.61 + 0.70 ms, 23 % of a MainNetwork wave).
//   * the split weights of the layer ([F/16 groups][W couts][hi 16 | lo 16], 64-B rows XOR-swizzled like the conv's stage tiles,
//     scaled by 2^s; k_restage_att_split) sit in LDS for the life of a persistent workgroup (96 KB, one workgroup of 4 waves per CU);
//   * one wave per board, as in k_attention_mfma.  x fragments are built from the f32 residual stream on the fly: lane (j, kq) loads
//     8 channels of row j and keeps their fp16 hi (kq < 2) or lo (kq >= 2) halves -- the [a_hi | a_lo] operand of a K = 32 step;
//   * phase A: q^T and k^T for the whole board, as D[c][pos] tiles -- which ARE the A / B operand layouts of the energy GEMM's f32
//     16x16x4 steps (k index = 4*(lane >> 4) + r).  A three-stage stream over the 16-channel groups: group g on the matrix cores,
//     group g+1 being split (VALU work in the shadow of the MFMAs), group g+2's loads in flight;
//   * phase B, per block of 16 rows i (tm): v for those rows and all channels as D[pos][c] tiles -- the A operand layout of the
//     output GEMM -- from the block's x rows (second pass over x; measured, it comes from HBM again: a board's 41 KB do not survive
//     in an L2 shared by 128 boards in flight); the block's 16 x 96 energies; softmax on the accumulator tiles with DPP row
//     reductions (a row is complete within the block); out[c][j] += (gamma v)[i][c] attention[i][j] into accumulators that cover all
//     channels and also take the residual x of column tile tm in block tm (from the lines the block's row loads just fetched).
//     Nothing but those accumulators outlives a block, so the 36 energy tiles of k_attention_mfma never exist at once and x is
//     split twice per board, not once per channel pass.
//   * epilogue: y = relu(bn(acc)) row-major and the next residual block's split input -- stores only (a load in here would queue
//     behind the stores: vmcnt is in order); the next board's first two groups are requested before the first store, its lines
//     were touched towards L2 (LDS-DMA into a scratch KB) at the start of phase B.
// Per board: 1152 K=32 fp16 steps (projection) + 1296 f32 16x16x4 steps (exact-f32 energy and output GEMMs) = 59 k cycles of MFMA;
// measured 130 k cycles per board (0.98 ms per 16 k boards; MFMA pipe 40 % busy).  PRO: x is relu(x*ps + pt) first (attention in
// the policy head, model.py:94,106).  Diagnostics: -DTG_ATT_STAMP (phase stamps, scripts/stamp_att.py), TG_ATT_X3=0 (the two-kernel
// form), -DTG_ATT_TOUCH=0, -DTG_ATT_IGLP=0.
#ifndef TG_ATT_IGLP
#define TG_ATT_IGLP 1
#endif
#if TG_ATT_IGLP
// NM MFMAs with KV VALU instructions (the split of the next group) in the shadow of each
#define TG_ATT_SCHED(NM, KV) do { _Pragma("unroll") for (int i_ = 0; i_ < (NM); ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
                                  __builtin_amdgcn_sched_group_barrier(0x002, (KV), 0); } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TG_ATT_SCHED(NM, KV) __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef TG_ATT_STAMP
// diagnostic build: phase stamps (s_memtime, 100 MHz) of the third board of wave 0 of workgroup 0, PRO = false launches
__device__ unsigned long long g_att_stamp[32];
#define TG_ASTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (!PRO && blockIdx.x == 0 && wave == 0 && nboard == 2 && lane == 0) g_att_stamp[i] = __builtin_readcyclecounter(); \
                          __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TG_ASTAMP(i) do { } while (0)
#endif
#ifndef TG_ATT_TOUCH
#define TG_ATT_TOUCH 1
#endif
#ifndef TG_ATT_RESID_PERM
#define TG_ATT_RESID_PERM 1
#endif
template <int S, int F, bool PRO>
__global__ __launch_bounds__(256, 1) void k_attention_x3(const float* __restrict__ xin, float* __restrict__ out, _Float16* __restrict__ out2,
                                                         const _Float16* __restrict__ wimg, const float* __restrict__ qb,
                                                         const float* __restrict__ wsc_p, const float* __restrict__ gamma,
                                                         const float* __restrict__ bs, const float* __restrict__ bt,
                                                         const float* __restrict__ ps, const float* __restrict__ pt,
                                                         const float* __restrict__ s2, const float* __restrict__ t2, int rows,
                                                         unsigned* __restrict__ ovf) {
    constexpr int P = S * S, FQ = F / 4, W = 2 * FQ + F, NT = (P + 15) / 16, CT = F / 16, NG = F / 16, NSUB = FQ / 16;
    static_assert(P <= 96 && FQ % 16 == 0 && NG % 2 == 0, "attention tile geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
    _Float16* const wl = reinterpret_cast<_Float16*>(att_smem);                       // [NG][W][32]
    for (int i = threadIdx.x; i < NG * W * 4; i += 256) reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(wimg)[i];
    // per-channel parameters behind the image: q|k|v bias [W], then bn scale / shift, next block's bn1 scale / shift, prologue scale /
    // shift [F each].  Read from LDS, they neither pin registers across the board loop nor queue behind the epilogue's stores (vector
    // memory loads and stores share the in-order vmcnt)
    float* const prm = reinterpret_cast<float*>(att_smem + (size_t)NG * W * 64);
    for (int i = threadIdx.x; i < W; i += 256) prm[i] = qb[i];
    for (int i = threadIdx.x; i < F; i += 256) {
        prm[W + i] = bs[i]; prm[W + F + i] = bt[i];
        prm[W + 2 * F + i] = out2 ? s2[i] : 1.f; prm[W + 3 * F + i] = out2 ? t2[i] : 0.f;
        prm[W + 4 * F + i] = PRO ? ps[i] : 1.f; prm[W + 5 * F + i] = PRO ? pt[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: board pointers stay in SGPRs
    const int j = lane & 15, kq = lane >> 4;
    const float lo_neg = kq >> 1 ? -1.f : -0.f;                                      // -1: this lane carries the lo halves of x
    const unsigned chx = (kq & 1) * 8;
    const int whi = j * 32 + (((kq & 1) ^ swz64(j)) << 3), wlo = j * 32 + (((2 + (kq & 1)) ^ swz64(j)) << 3);
    // ds_read offsets are 16-bit: the image's upper half gets its own (opaque) base so that no address needs a register of its own
    int upper = NG / 2 * W * 32;
    asm volatile("" : "+v"(upper));
    const _Float16* const wl1 = wl + upper;
    auto wgroup = [&](int g) { return g < NG / 2 ? wl + g * W * 32 : wl1 + (g - NG / 2) * W * 32; };
    const float wsc = wsc_p[0];
    const int M = rows * P;
    // Addresses: tile t of a board starts 16 rows = 16*F floats further (a scalar add on the board pointer); within the tile lane j
    // takes row j -- except in the last tile, where rows past the board clamp to its last row.  Four per-lane byte offsets serve
    // every access: {first five tiles, last tile} x {8-channel column of the projection reads, 4-channel column of the D tiles}.
    constexpr int LASTR = P - 1 - (NT - 1) * 16;                                     // last valid row of the last tile
    const unsigned rowA = (unsigned)j * F * 4u, rowB = (unsigned)(j <= LASTR ? j : LASTR) * F * 4u;
    const unsigned xoA = rowA + chx * 4u, xoB = rowB + chx * 4u, eoA = rowA + kq * 16u;
    [[maybe_unused]] const unsigned eoB = rowB + kq * 16u;
    const unsigned hoA = ((unsigned)(kq >> 1) * M + j) * 16u + (kq & 1) * 8u;        // split chunk-major output, + t*256 per tile
    auto tile = [](const float* board, int t) { return board + t * 16 * F; };
    auto ld16 = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off); };
    // 8 channels of one row -> this lane's half of the split fragment
    const float* l_ps = nullptr; const float* l_pt = nullptr;                          // set per board (LDS parameter block)
    struct Pro { f32x4 sc[2], sh[2]; };                                               // prologue scale / shift of this lane's 8 channels of a group
    auto pro_of = [&](int g) {
        Pro p{};
        if constexpr (PRO) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { p.sc[h] = *reinterpret_cast<const f32x4*>(l_ps + g * 16 + chx + 4 * h); p.sh[h] = *reinterpret_cast<const f32x4*>(l_pt + g * 16 + chx + 4 * h); }
        }
        return p;
    };
    // range guard (see conv_epilogue_h8): the largest |x| this wave split in phase A (phase B splits the same values again) and the
    // largest value the epilogue rounds to fp16; checked once per board
    float amax = 0.f;
    auto split8 = [&](const f32x4 (&src)[2], const Pro& pr, bool track = false) -> h8 {
        h8 xf;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 v = {src[e >> 2][e & 3], src[e >> 2][(e & 3) + 1]};
            if constexpr (PRO) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { const float w = v[q] * pr.sc[e >> 2][(e & 3) + q] + pr.sh[e >> 2][(e & 3) + q]; v[q] = w > 0.f ? w : 0.f; }
            }
            if (track) amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
            const f32x2 hi = __builtin_convertvector(__builtin_convertvector(v, h2), f32x2);
            // hi lanes (lo_neg = -0): half(v); lo lanes (lo_neg = -1): half(v - hi), the product is exact either way
            const f32x2 d = {__builtin_fmaf(lo_neg, hi[0], v[0]), __builtin_fmaf(lo_neg, hi[1], v[1])};
            const h2 o = __builtin_convertvector(d, h2);
            xf[e] = o[0]; xf[e + 1] = o[1];
        }
        return xf;
    };
    f32x4 rawA[NT][2], rawB[NT][2];                                                  // phase A stream: one group of all six tiles each
    h8 xfA[NT], xfB[NT];
    // (the lane offsets are made opaque where they are used: hoisted out of the board loop in their 64-bit form they are spilled, and the
    // loads then take full VGPR addresses reloaded from scratch -- behind whatever the wave has in flight)
    auto issue = [&](const float* xb, int g, f32x4 (&dst)[NT][2]) {
        unsigned oa = xoA, ob = xoB;
        asm volatile("" : "+v"(oa), "+v"(ob));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            dst[t][0] = ld16(tile(xb, t), (t + 1 < NT ? oa : ob) + g * 64);
            dst[t][1] = ld16(tile(xb, t), (t + 1 < NT ? oa : ob) + g * 64 + 16);
        }
    };
    auto split = [&](const f32x4 (&src)[NT][2], int g, h8 (&xf)[NT]) {
        const Pro pr = pro_of(g);
#pragma unroll
        for (int t = 0; t < NT; ++t) xf[t] = split8(src[t], pr, true);
    };
    int b = blockIdx.x * 4 + wave;
    if (b >= rows) return;
    { const float* xb0 = xin + (size_t)b * P * F; issue(xb0, 0, rawB); issue(xb0, 1, rawA); }
    int nboard = 0;
    for (; b < rows; b += gridDim.x * 4, ++nboard) {
        const float* xb = xin + (size_t)b * P * F;
        // opaque per board: LDS is read-only from here on, so every parameter read would otherwise be hoisted out of the board loop
        // (and pin, then spill, ~150 registers)
        unsigned popq = 0;
        asm volatile("" : "+v"(popq));                                                // (an opaque OFFSET: the pointer keeps its LDS address space)
        const float* const l_qb = prm + popq;
        const float* const l_bs = l_qb + W; const float* const l_bt = l_bs + F;
        const float* const l_s2 = l_bt + F; const float* const l_t2 = l_s2 + F; l_ps = l_t2 + F; l_pt = l_ps + F;
        TG_ASTAMP(0);
        // ---- phase A: q^T and k^T, [c][pos] tiles ----
        f32x4 qk[2 * NSUB][NT];
#pragma unroll
        for (int ct = 0; ct < 2 * NSUB; ++ct)
#pragma unroll
            for (int t = 0; t < NT; ++t) qk[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto mfma_qk = [&](int g, const h8 (&xf)[NT]) {
#pragma unroll
            for (int ct = 0; ct < 2 * NSUB; ++ct) {
                const _Float16* wr = wgroup(g) + ct * 16 * 32;
                const h8 ah = *reinterpret_cast<const h8*>(wr + whi), al = *reinterpret_cast<const h8*>(wr + wlo);
#pragma unroll
                for (int t = 0; t < NT; ++t) qk[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xf[t], qk[ct][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) qk[ct][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xf[t], qk[ct][t], 0, 0, 0);
            }
        };
        split(rawB, 0, xfA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            if (g + 2 < NG) issue(xb, g + 2, rawB);
            mfma_qk(g, xfA);
            split(rawA, g + 1, xfB);
            TG_ATT_SCHED(2 * NSUB * NT * 2, 3);
            if (g + 3 < NG) issue(xb, g + 3, rawA);
            mfma_qk(g + 1, xfB);
            if (g + 2 < NG) split(rawB, g + 2, xfA);
            TG_ATT_SCHED(2 * NSUB * NT * 2, 3);
            if (g == 0) TG_ASTAMP(29);
        }
        TG_ASTAMP(1);
        // ---- phase B, block tm = rows i in [16 tm, 16 tm + 16) ----
        f32x4 rv[NG][2];                                                             // the block's x rows, all groups
        auto issue_rows = [&](int tm) {
            unsigned o = tm + 1 < NT ? xoA : xoB;
            asm volatile("" : "+v"(o));
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                rv[g][0] = ld16(tile(xb, tm), o + g * 64);
                rv[g][1] = ld16(tile(xb, tm), o + g * 64 + 16);
            }
        };
        issue_rows(0);
#pragma unroll
        for (int ct = 0; ct < 2 * NSUB; ++ct) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(l_qb + ct * 16 + kq * 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) qk[ct][t] = qk[ct][t] * wsc + bq;
        }
        // The residual x (PRO: relu(x*ps + pt)) is ADDED INTO the output accumulators, column tile tn in row block tm = tn, and v
        // carries the factor gamma, so that they end as gamma * out + x and the epilogue has no loads at all: a load queued behind
        // the epilogue's stores would wait for their acknowledgements (vmcnt is in order), and did -- 30 % of a board's time in the
        // first version.  Taken in block tn, the residual rows are the lines that block's own row loads have just brought in (read
        // up front for all tiles they were a third trip to HBM: 2.17 GB of reads per launch against 0.67 GB of x).
        const float gam = gamma[0];
        f32x4 acc[CT][NT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[ct][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
        // next board's rows towards L2 now (one dword per 128-B line, results unused): phase A is the first touch of a board's x and
        // its loads run only one group ahead of their use -- from HBM that was 17 % of a board's time
        const int bnx = b + gridDim.x * 4;
        const float* const xnext = xin + (size_t)(bnx < rows ? bnx : b) * P * F;
#if TG_ATT_TOUCH
        // as LDS-DMA into a scratch KB of this wave: no destination registers, nothing ever waits for them
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const unsigned lo = (unsigned)(lane + 64 * i) * 128u, off = lo < (unsigned)(P * F * 4 - 16) ? lo : (unsigned)(P * F * 4 - 16);
            tg_dma_global(xnext, (int)off, (tg_lds_void*)(&att_smem[(size_t)NG * W * 64 + (W + 6 * F) * 4 + wave * 1024]));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        TG_ASTAMP(27);
#pragma unroll
        for (int tm = 0; tm < NT; ++tm) {
            // v for the block: D[pos][c], all channel tiles
            f32x4 va[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) va[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            // weight fragments: eight in flight, each slot refilled (for the next half-group) right behind the MFMA that used it; the
            // fences and group barriers keep that distance (left alone, the scheduler sinks every read next to its MFMA and the wave
            // sits on LDS latency 128 times per block)
            h8 wf[CT];
            auto wslot = [&](int hg, int ct) {                                       // half-group hg = 2*g + (0: hi, 1: lo)
                return *reinterpret_cast<const h8*>(wgroup(hg >> 1) + (2 * FQ + ct * 16) * 32 + ((hg & 1) ? wlo : whi));
            };
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) wf[ct] = wslot(0, ct);
            h8 xf = split8(rv[0], pro_of(0)), xfn = xf;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int hg = 0; hg < 2 * NG; ++hg) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    va[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, wf[ct], va[ct], 0, 0, 0);
                    if (hg + 1 < 2 * NG) wf[ct] = wslot(hg + 1, ct);
                }
                if ((hg & 1) == 0 && hg + 2 < 2 * NG) xfn = split8(rv[hg / 2 + 1], pro_of(hg / 2 + 1));   // next group's fragment, in the MFMA shadow
#if TG_ATT_IGLP
#pragma unroll
                for (int i_ = 0; i_ < CT; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
                if (hg & 1) xf = xfn;
            }
            TG_ASTAMP(2 + 4 * tm);
            __builtin_amdgcn_sched_barrier(0);
            // the residual rows of column tile tm, D layout (row = channel ct*16 + kq*4 + r, column = position tm*16 + j).
            // TG_ATT_RESID_PERM (round 4, default): taken from the block's ROW registers, which hold exactly these values in the
            // projection layout -- lane (j, kq') has channels g*16 + (kq' & 1)*8 .. +7 of row j, the lane pairs kq' and kq' + 2
            // hold the same eight -- so lanes kq' < 2 offer their first four, lanes kq' >= 2 their last four, and destination
            // (j, kq) pulls from (j, (kq >> 1) + 2*(kq & 1)): one ds_bpermute per register, no memory access at all.  (Round 3
            // read them from memory again: meant to be cache hits on the lines the row loads had just fetched, they were the third
            // trip to HBM -- the XCD's L2 turns over within a row block, profiles/r3_pmc_attention_x3.json: 2.19 GB fetched.)
            f32x4 xa[CT];
#if TG_ATT_RESID_PERM
            {
                const int src = ((((kq >> 1) + 2 * (kq & 1)) << 4) + j) << 2;          // byte address of the source lane
                const bool hi_half = kq >= 2;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    // (scalar copies: a bit_cast of a vector ELEMENT sends the whole vector through scratch)
                    const float s0 = hi_half ? rv[ct][1][0] : rv[ct][0][0], s1 = hi_half ? rv[ct][1][1] : rv[ct][0][1];
                    const float s2 = hi_half ? rv[ct][1][2] : rv[ct][0][2], s3 = hi_half ? rv[ct][1][3] : rv[ct][0][3];
                    xa[ct] = f32x4{__int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s0))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s1))),
                                   __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s2))), __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(s3)))};
                }
            }
#else
            {
                unsigned o = tm + 1 < NT ? eoA : eoB;
                asm volatile("" : "+v"(o));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) xa[ct] = ld16(tile(xb, tm), o + ct * 64);
            }
#endif
            if (tm + 1 < NT) issue_rows(tm + 1);                                      // lands during the block's energy / output GEMMs
            __builtin_amdgcn_sched_barrier(0);                                       // (kept here: sunk to the block's end they are waited for at once)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) va[ct] = (va[ct] * wsc + l_qb[2 * FQ + ct * 16 + j]) * gam;
            // energies of the block (exact f32): e[tn] = q[tm] . k[tn]
            f32x4 e[NT];
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) e[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn)
                        e[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(qk[sub][tm][s4], qk[NSUB + sub][tn][s4], e[tn], 0, 0, 0);
            TG_ASTAMP(3 + 4 * tm);
            // softmax over j (columns) for the rows i = tm*16 + kq*4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = -INFINITY;
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) if (tn * 16 + j < P) mx = e[tn][r] > mx ? e[tn][r] : mx;
#pragma unroll
                for (int o = 0; o < 4; ++o) { const float t = row16_step(mx, o); mx = t > mx ? t : mx; }
                float sum = 0.f;
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    const float v = tn * 16 + j < P ? __expf(e[tn][r] - mx) : 0.f;        // v_exp_f32: ~1e-7 relative, tolerance is 1e-3
                    e[tn][r] = v; sum += v;
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) sum += row16_step(sum, o);
                const float inv = (tm * 16 + kq * 4 + r < P) ? 1.f / sum : 0.f;     // rows past the board contribute nothing below
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) e[tn][r] *= inv;
            }
            TG_ASTAMP(4 + 4 * tm);
            // out[c][j] += v[i][c] attention[i][j]: k-step r covers rows i = tm*16 + kq*4 + r; of the last block only the steps that
            // touch a row < P exist
            constexpr int LASTS = P - (NT - 1) * 16, NS_LAST = LASTS < 4 ? LASTS : 4;
            const int ns = tm + 1 < NT ? 4 : NS_LAST;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= ns) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn)
                        acc[ct][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[ct][r], e[tn][r], acc[ct][tn], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x4 xv = xa[ct];
                if constexpr (PRO) {
                    const f32x4 vps = *reinterpret_cast<const f32x4*>(l_ps + ct * 16 + kq * 4), vpt = *reinterpret_cast<const f32x4*>(l_pt + ct * 16 + kq * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const float w = xv[q] * vps[q] + vpt[q]; xv[q] = w > 0.f ? w : 0.f; }
                }
                acc[ct][tm] = acc[ct][tm] + xv;
            }
            __builtin_amdgcn_sched_barrier(0);
            TG_ASTAMP(5 + 4 * tm);
        }
        // ---- epilogue.  D tile: row = channel ct*16 + kq*4 + r, column = position tn*16 + j ----
        // every address = a per-board scalar base + one of six per-lane byte offsets + a compile-time constant.  Loads are kept
        // AHEAD of the stores in issue order (a load queued behind stores waits for their acknowledgements): the next board's first
        // two groups go out before the first store, and the residual is already inside the accumulators.
        {
            issue(xnext, 0, rawB); issue(xnext, 1, rawA);                             // unconditional (last board: its own rows again)
        }
        char* const yb = reinterpret_cast<char*>(out + (size_t)b * P * F);
        // opaque per board: hoisted out of the board loop, the 64-bit forms of these offsets would be spilled -- and a spill reload in
        // here queues behind the stores
        unsigned eoE = eoA, hoE = hoA;
        asm volatile("" : "+v"(eoE), "+v"(hoE));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int c = ct * 16 + kq * 4;
            const f32x4 vbs = *reinterpret_cast<const f32x4*>(l_bs + c), vbt = *reinterpret_cast<const f32x4*>(l_bt + c);
            const f32x4 vs2 = *reinterpret_cast<const f32x4*>(l_s2 + c), vt2 = *reinterpret_cast<const f32x4*>(l_t2 + c);
            // split output: element (m, x2_index(c)) of the chunk-major tensor = chunk plane ct*4 + (kq >> 1) (lo: + 2), row m, half (kq & 1)*4
            char* const hb = reinterpret_cast<char*>(out2) + ((size_t)ct * 4 * M + (size_t)b * P) * 16;
            char* const lb = hb + (size_t)2 * M * 16;
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                if (tn * 16 + j >= P) continue;
                f32x4 y, u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float w = acc[ct][tn][q] * vbs[q] + vbt[q];
                    y[q] = w > 0.f ? w : 0.f;
                    const float z = y[q] * vs2[q] + vt2[q];
                    u[q] = z > 0.f ? z : 0.f;
                }
                *reinterpret_cast<f32x4*>(yb + tn * 16 * F * 4 + (eoE + ct * 64)) = y;
                if (out2) {
                    h4 hi, lo;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi[q] = (_Float16)u[q]; lo[q] = (_Float16)(u[q] - (float)hi[q]); amax = __builtin_fmaxf(amax, u[q]); }
                    *reinterpret_cast<h4*>(hb + (hoE + tn * 256)) = hi;
                    *reinterpret_cast<h4*>(lb + (hoE + tn * 256)) = lo;
                }
            }
            if (ct == 3) TG_ASTAMP(28);
        }
        TG_ASTAMP(26);
        if (__builtin_amdgcn_ballot_w64(!(amax <= 65504.f)) != 0) { if (lane == 0) atomicAdd(ovf, 1u); amax = 0.f; }
    }
    TG_VMCNT(0);                                                                     // the last board's touches and stores
}

// the LDS image of k_attention_x3: dst[g][cout][32] = [hi of channels g*16..+15 | lo of the same] of w[cout][F] * 2^s, the four
// 16-B chunks of a row XOR-swizzled by swz64(cout) (the fragment reads of the kernel undo it); wsc_out = 2^-s
__global__ __launch_bounds__(256) void k_restage_att_split(const float* __restrict__ w, _Float16* __restrict__ dst, float* __restrict__ wsc_out,
                                                           const unsigned* __restrict__ maxbits, int W, int F) {
    const unsigned mb = maxbits[0];
    int sh = mb ? 14 - ((int)((mb >> 23) & 0xffu) - 127) : 0;
    sh = sh < -60 ? -60 : sh > 60 ? 60 : sh;
    const float scale = ldexpf(1.f, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) wsc_out[0] = ldexpf(1.f, -sh);
    const int total = F / 16 * W * 32;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int k = i & 31, r = (i >> 5) % W, g = (i >> 5) / W;
        const int kk = (((k >> 3) ^ swz64(r)) << 3) + (k & 7);                       // logical position of this physical slot
        const float v = w[(size_t)r * F + g * 16 + (kk & 15)] * scale;
        const _Float16 hi = (_Float16)v;
        dst[i] = (kk & 16) ? (_Float16)(v - (float)hi) : hi;
    }
}

// Heads after the 3x3 head convs (hc[row][p][16]: channels 0-1 value/own, 2-5 policy; BN+ReLU already applied).
// HR rows per workgroup (8 at 9x9, 2 at 19x19): the dense weights (fc_act alone is 4P x A floats = 106 KB at 9x9) are read once per workgroup and used
// for all its rows -- with one row per workgroup the kernel moved 1.7 GB of L2 -> CU traffic per 16384 rows.  Every output is the
// same k-ordered fmaf chain as before, whatever the batching, so a row's results do not depend on which rows share its workgroup.
template <int S> struct HeadRows { static constexpr int N = S == 9 ? 8 : 2; };      // LDS: 6P + A + 64 floats per row
template <int S>
__global__ __launch_bounds__(256) void k_heads(const float* __restrict__ hc, const float* __restrict__ hca, const float* __restrict__ w_vo,
                                               const float* __restrict__ b_vo, const float* __restrict__ w_v,
                                               const float* __restrict__ b_v, const float* __restrict__ w_o,
                                               const float* __restrict__ b_o, const float* __restrict__ w_a,
                                               const float* __restrict__ b_a, float* __restrict__ policy,
                                               float* __restrict__ value, float* __restrict__ own, int rows) {
    constexpr int P = S * S, A = P + 1, HR = HeadRows<S>::N;
    __shared__ float hin[HR][6 * P];
    __shared__ float hid[HR][64];
    __shared__ float logit[HR][A];
    __shared__ float red[HR][2];
    const int row0 = blockIdx.x * HR, tid = threadIdx.x;
    const int nr = rows - row0 < HR ? rows - row0 : HR;
    for (int i = tid; i < HR * 6 * P; i += 256) {      // channels 0-1 from the value conv, 2-5 from the policy conv
        const int r = i / (6 * P), k = i % (6 * P), c = k / P, p = k % P;
        hin[r][k] = r < nr ? (c < 2 ? hc : hca)[((size_t)(row0 + r) * P + p) * 16 + c] : 0.f;
    }
    __syncthreads();
    // (spreading the rows of the workgroup over all 256 threads -- two / four row groups, each thread half / a quarter of the rows,
    // the same fmaf chains -- was measured: 135 us per launch against 92; the loops are bound by their weight loads, whose number per
    // thread stays the same and whose total doubles.  Splitting the input range instead would change the summation order.)
    if (tid < 64) {                                                   // fc_val_own + ReLU (model.py:99)
        float a[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) a[r] = b_vo[tid];
        for (int i = 0; i < 2 * P; ++i) {
            const float w = w_vo[i * 64 + tid];
#pragma unroll
            for (int r = 0; r < HR; ++r) a[r] = fmaf(hin[r][i], w, a[r]);
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) hid[r][tid] = a[r] > 0.f ? a[r] : 0.f;
    }
    for (int o = tid; o < A; o += 256) {                              // fc_act (model.py:110)
        float a[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) a[r] = b_a[o];
        for (int i = 0; i < 4 * P; ++i) {
            const float w = w_a[(size_t)i * A + o];
#pragma unroll
            for (int r = 0; r < HR; ++r) a[r] = fmaf(hin[r][2 * P + i], w, a[r]);
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) logit[r][o] = a[r];
    }
    __syncthreads();
    if (tid < nr) {
        const int r = tid;
        float a = b_v[0];
        for (int k = 0; k < 64; ++k) a = fmaf(hid[r][k], w_v[k], a);
        value[row0 + r] = tanhf(a);                                   // model.py:101
        float mx = logit[r][0];
        for (int o = 1; o < A; ++o) mx = logit[r][o] > mx ? logit[r][o] : mx;
        red[r][0] = mx;
    }
    if (own)
        for (int o = tid; o < P; o += 256) {                          // fc_own (model.py:102)
            float a[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r) a[r] = b_o[o];
            for (int k = 0; k < 64; ++k) {
                const float w = w_o[k * P + o];
#pragma unroll
                for (int r = 0; r < HR; ++r) a[r] = fmaf(hid[r][k], w, a[r]);
            }
#pragma unroll
            for (int r = 0; r < HR; ++r) if (r < nr) own[(size_t)(row0 + r) * P + o] = tanhf(a[r]);
        }
    __syncthreads();
    for (int i = tid; i < HR * A; i += 256) { const int r = i / A, o = i % A; logit[r][o] = expf(logit[r][o] - red[r][0]); }
    __syncthreads();
    if (tid < nr) { float s2 = 0.f; for (int o = 0; o < A; ++o) s2 += logit[tid][o]; red[tid][1] = s2; }
    __syncthreads();
    for (int i = tid; i < nr * A; i += 256) {                          // softmax, model.py:111
        const int r = i / A, o = i % A;
        policy[(size_t)(row0 + r) * A + o] = logit[r][o] * (1.f / red[r][1]);
    }
}

struct ProfScope {
    Net* n; hipStream_t st; bool on;
    ProfScope(Net* net, hipStream_t s, double flops) : n(net), st(s), on(net->prof && net->ev_used + 2 <= net->ev.size()) {
        if (on) { (void)hipEventRecord(n->ev[n->ev_used], st); n->conv_flops += flops; n->conv_launches++; }
        else if (net->prof) n->conv_skipped++;
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(n->ev[n->ev_used + 1], st); n->ev_used += 2; } }
};

template <int S, int F>
int forward_t(tg_ctx* ctx, Net* n, const float* obs, int rows, float* policy, float* value, float* own) {
    constexpr int P = S * S;
    hipStream_t st = ctx->stream;
    const int M = rows * P;
    const int grid = (M + 127) / 128;                 // stem / head / 1x1 convs: 128 rows per workgroup
    // position tiles per wave in the F->F convs (workgroup = 64*NPT rows): 3 while the accumulators (F/16*NPT*4 registers)
    // still leave room for two waves per SIMD, else 2.  Measured at F=128: 128.0 -> 132.3 TFLOP/s.
    constexpr int NPT = F <= 128 ? 3 : 2;
    const int grid_f = (M + 64 * NPT - 1) / (64 * NPT);
    const double conv_flops = 2.0 * 9.0 * (double)F * (double)F * (double)M;
    // head conv F -> 16 (6 real couts): GEMM + col2im (k_head_gemm); TG_HEAD_GEMM=0 selects the implicit-GEMM kernel it replaced
    auto head_conv = [&](const float* in, float* outp, const float* Wg, const ConvW& cw, const float* ps, const float* pt) {
        const char* hg = getenv("TG_HEAD_GEMM");
        if (hg && atoi(hg) == 0) {
            if (ps) hipLaunchKernelGGL((k_conv3x3<S, F, 16, true, 0>), dim3(grid), dim3(256), 0, st, in, outp, (const float*)nullptr, cw.w, cw.b, ps, pt, M);
            else hipLaunchKernelGGL((k_conv3x3<S, F, 16, false, 0>), dim3(grid), dim3(256), 0, st, in, outp, (const float*)nullptr, cw.w, cw.b,
                                    (const float*)nullptr, (const float*)nullptr, M);
            return;
        }
        constexpr int TMH = 64 * (S == 9 ? 3 : 4) - 2 * (S + 1);
        const int nt = (M + TMH - 1) / TMH, g = nt < 512 ? nt : 512;      // two workgroups per CU walk the tile list
        if (ps) hipLaunchKernelGGL((k_head_gemm<S, F, true>), dim3(g), dim3(256), 0, st, in, outp, Wg, cw.b, ps, pt, M, nt);
        else hipLaunchKernelGGL((k_head_gemm<S, F, false>), dim3(g), dim3(256), 0, st, in, outp, Wg, cw.b, (const float*)nullptr, (const float*)nullptr, M, nt);
    };
    int g0 = (int)(((size_t)M * 16 + 255) / 256); if (g0 > 65535) g0 = 65535;
    if (n->prec == 0) {
        if (n->in_bits) hipLaunchKernelGGL((k_bits_to_rows<S>), dim3(g0), dim3(256), 0, st, n->in_bits, n->in_slot, n->x0, rows, n->C, n->in_words);
        else hipLaunchKernelGGL((k_obs_to_rows<S>), dim3(g0), dim3(256), 0, st, obs, n->x0, rows, n->C);
    }
    float* x = n->bufA; float* y = n->bufB;
    if constexpr (F == 128 || F == 256) {
        if (n->prec == 3) {
            // Split-precision chain ("f32x3"): the f32 tower's arithmetic with every conv operand carried as fp16 hi + lo and all
            // three of the four partial products on the fp16 matrix cores (k_conv3x3_h2<..., X2>): stem and tower convs; the residual stream
            // stays f32 row-major, the dense heads are the f32 kernel, the narrow head conv runs split as well at 128 filters
            // (k_head_gemm_x2).  Opt-in: not the default path.
            if ((long long)M * F * 4 >= (1ll << 31)) TG_FAIL(ctx, TG_ERR_ARG, "split-precision path: rows * P * F * 4 bytes must stay below 2 GiB per activation buffer");
            const int grid_h = (M + 255) / 256;
            constexpr int COS = F / 128;
            const int nblk_h2 = (COS == 1 && !TG_H2_XCD) ? grid_h : (grid_h + 7) / 8 * 8 * COS;
            const size_t nb = n->blocks.size();
            int g0h = (int)(((size_t)M * 8 + 255) / 256); if (g0h > 65535) g0h = 65535;
            if (n->in_bits) hipLaunchKernelGGL((k_bits_to_rows_h<S>), dim3(g0h), dim3(256), 0, st, n->in_bits, n->in_slot, n->x0h, rows, n->C, n->in_words);
            else hipLaunchKernelGGL((k_obs_to_rows_h<S>), dim3(g0h), dim3(256), 0, st, obs, n->x0h, rows, n->C);
            const float* wsc = n->wsc;
            // The layer program (attention layers allowed at 9x9): residual blocks on the split-precision convs; a Self_Attention
            // layer reads and writes the f32 residual stream with the f32 kernels (1x1 q/k/v projection + k_attention_mfma) and,
            // when a residual block follows, hands it relu(bn1_next(y)) already split (X2O).
            const size_t nl = n->layers.size();
            // the head conv on the split fp16 MFMA too (k_head_gemm_x2, F = 128): the LAST residual block then writes relu(bn_end(.))
            // split, as it would for a block behind it, and -- when nothing else reads the f32 stream (no attention in the policy
            // head) -- leaves the stream itself unwritten (TG_HEAD_X2=0: the f32 head conv)
            static const bool head_x2_on = !(getenv("TG_HEAD_X2") && atoi(getenv("TG_HEAD_X2")) == 0);
            const bool x2_head = F == 128 && head_x2_on && n->head_x2 && nl && n->layers[nl - 1].kind == 0;
            auto next_bn = [&](size_t i, const float** sn, const float** tn) -> bool {   // layer i+1 is a residual block?
                if (i + 1 < nl && n->layers[i + 1].kind == 0) { const BlockW& nb2 = n->blocks[n->layers[i + 1].ridx]; *sn = nb2.s1; *tn = nb2.t1; return true; }
                if (i + 1 == nl && x2_head) { *sn = n->s_end; *tn = n->t_end; return true; }
                *sn = nullptr; *tn = nullptr; return false;
            };
            constexpr int WQ = F / 4 + F / 4 + F;
            auto attention_x2 = [&](const AttW& a, const float* xin, float* xout, const float* ps, const float* pt, _Float16* o2,
                                    const float* sn, const float* tn) {
                if constexpr (S == 9 && F == 128) {
                    // one kernel per attention block: q/k/v projected inside the core, in split precision (TG_ATT_X3=0: the pair below)
                    static const bool fused = !(getenv("TG_ATT_X3") && atoi(getenv("TG_ATT_X3")) == 0);
                    if (fused && a.x3w) {
                        constexpr int lds3 = (int)(sizeof(_Float16) * F / 16 * WQ * 32 + sizeof(float) * (WQ + 6 * F) + 4096);   // image, parameters, touch scratch
                        const int nwg = (rows + 3) / 4 < 256 ? (rows + 3) / 4 : 256;
                        if (ps) hipLaunchKernelGGL((k_attention_x3<S, F, true>), dim3(nwg), dim3(256), lds3, st, xin, xout, o2, a.x3w, a.qkv.b, a.x3sc,
                                                   a.gamma, a.s, a.t, ps, pt, sn, tn, rows, n->range);
                        else hipLaunchKernelGGL((k_attention_x3<S, F, false>), dim3(nwg), dim3(256), lds3, st, xin, xout, o2, a.x3w, a.qkv.b, a.x3sc,
                                                a.gamma, a.s, a.t, ps, pt, sn, tn, rows, n->range);
                        return;
                    }
                }
                if constexpr (S == 9) {
                    if (ps)
                        hipLaunchKernelGGL((k_conv3x3<S, F, WQ, true, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                                           (const float*)nullptr, a.qkv.w, a.qkv.b, ps, pt, M);
                    else
                        hipLaunchKernelGGL((k_conv3x3<S, F, WQ, false, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                                           (const float*)nullptr, a.qkv.w, a.qkv.b, (const float*)nullptr, (const float*)nullptr, M);
                    hipLaunchKernelGGL((k_attention_mfma<S, F, 2, true>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float*)n->bufQ, xin, xout,
                                       reinterpret_cast<float*>(o2), a.gamma, a.s, a.t, ps, pt, sn, tn, rows, n->range);
                }
            };
            const float* s0 = nullptr; const float* t0 = nullptr;
            const bool act0 = nl && n->layers[0].kind == 0;
            if (act0) { s0 = n->blocks[n->layers[0].ridx].s1; t0 = n->blocks[n->layers[0].ridx].t1; }
            hipLaunchKernelGGL((k_conv3x3_h2<S, 64, F, 4, false, true>), dim3(nblk_h2), dim3(256), 0, st, (const _Float16*)n->x0h, x,
                               act0 ? n->act16 : (_Float16*)nullptr, (const float*)nullptr, (const _Float16*)n->stem_h, n->stem.b, s0, t0,
                               M, nblk_h2, wsc + 2 * nb, n->range);
            for (size_t i = 0; i < nl; ++i) {
                const Layer& L = n->layers[i];
                const float* sn; const float* tn;
                const bool act = next_bn(i, &sn, &tn);
                if (L.kind == 1) {
                    attention_x2(L.a, x, y, nullptr, nullptr, act ? n->act16 : (_Float16*)nullptr, sn, tn);
                    float* t = x; x = y; y = t;
                    continue;
                }
                const BlockW& b = n->blocks[L.ridx];
                { ProfScope ps(n, st, conv_flops);
                  hipLaunchKernelGGL((k_conv3x3_h2<S, 2 * F, F, 0, false, true>), dim3(nblk_h2), dim3(256), 0, st, (const _Float16*)n->act16,
                                     (float*)nullptr, n->h16, (const float*)nullptr, b.h1, b.c1.b, (const float*)nullptr, (const float*)nullptr, M, nblk_h2,
                                     wsc + 2 * L.ridx, n->range); }
                float* const y32 = (i + 1 == nl && x2_head && !n->pol_att) ? (float*)nullptr : y;     // nothing reads the last block's f32 stream
                { ProfScope ps(n, st, conv_flops);
                  hipLaunchKernelGGL((k_conv3x3_h2<S, 2 * F, F, 1, false, true>), dim3(nblk_h2), dim3(256), 0, st, (const _Float16*)n->h16,
                                     y32, act ? n->act16 : (_Float16*)nullptr, (const float*)x, b.h2, b.c2.b, sn, tn, M, nblk_h2, wsc + 2 * L.ridx + 1, n->range); }
                float* t = x; x = y; y = t;
            }
            if (x2_head) {
                if constexpr (F == 128) {
                    constexpr int TMH = 64 * (S == 9 ? 3 : 4) - 2 * (S + 1);
                    const int nt = (M + TMH - 1) / TMH, g = nt < 512 ? nt : 512;
                    hipLaunchKernelGGL((k_head_gemm_x2<S, F>), dim3(g), dim3(256), 0, st, (const _Float16*)n->act16, n->hc, n->head_x2, n->head_x2sc, n->head.b, M, nt);
                }
            } else {
                // the head conv is 16 couts wide (one MFMA tile): the f32 kernel reads the f32 residual stream and activates while staging
                head_conv(x, n->hc, n->head_g, n->head, n->s_end, n->t_end);
            }
            const float* hca3 = n->hc;
            if (n->pol_att) {                              // attention in the policy head (model.py:72,106-107), f32 head conv behind it
                // (the policy head conv on the split MFMA as well was measured: what it saves the attention block's extra split write
                // of its output costs -- 723.3 vs 723.2 k sims/s -- so it reads the f32 output)
                attention_x2(n->patt, x, y, n->s_end, n->t_end, (_Float16*)nullptr, (const float*)nullptr, (const float*)nullptr);
                head_conv(y, n->hca, n->head_ag, n->head_a, nullptr, nullptr);
                hca3 = n->hca;
            }
            hipLaunchKernelGGL((k_heads<S>), dim3((rows + HeadRows<S>::N - 1) / HeadRows<S>::N), dim3(256), 0, st, (const float*)n->hc, hca3, n->w_vo, n->b_vo,
                               n->w_v, n->b_v, n->w_o, n->b_o, n->w_a, n->b_a, policy, value, own, rows);
            TG_HIP(ctx, hipGetLastError());
            return TG_OK;
        }
        if (n->prec >= 1) {
            // fp16 chain: every conv (stem, tower, head conv) takes fp16 operands and accumulates in f32; the small dense heads
            // (k_heads) stay f32, and so does the residual stream x/y unless net_precision is 2 (then x/y are fp16, slice-major, and
            // live in the same buffers; a block's output is still rounded once, from the f32 accumulator)
            const bool r16 = n->prec == 2;
            if ((long long)M * F * 2 >= (1ll << 31)) TG_FAIL(ctx, TG_ERR_ARG, "fp16 path: rows * P * F * 2 bytes must stay below 2 GiB per activation buffer");
            const int grid_h = (M + 255) / 256;                                      // 256-row tiles
            constexpr int COS = F / 128;
            const int nblk_h2 = (COS == 1 && !TG_H2_XCD) ? grid_h : (grid_h + 7) / 8 * 8 * COS;      // (row tile, cout part) blocks
            // one workgroup per block: the kernel can walk a tile list (grid < nblk), but the hardware dispatcher balances
            // better than a static list -- 512 persistent workgroups measured 0.6-2 % slower
            const int grid_h2 = nblk_h2;
            const size_t nb = n->blocks.size();
            // stem on the fp16 matrix cores too (input planes are 0/1, exact in fp16; 16 planes padded to 64 channels = 18 stages):
            // writes the f32 residual stream x and the first conv input relu(bn_next(x)) as fp16
            int g0h = (int)(((size_t)M * 8 + 255) / 256); if (g0h > 65535) g0h = 65535;
            if (n->in_bits) hipLaunchKernelGGL((k_bits_to_rows_h<S>), dim3(g0h), dim3(256), 0, st, n->in_bits, n->in_slot, n->x0h, rows, n->C, n->in_words);
            else hipLaunchKernelGGL((k_obs_to_rows_h<S>), dim3(g0h), dim3(256), 0, st, obs, n->x0h, rows, n->C);
            if (r16)
                hipLaunchKernelGGL((k_conv3x3_h2<S, 64, F, 4, true>), dim3(grid_h2), dim3(256), 0, st, (const _Float16*)n->x0h, x, n->act16,
                                   (const float*)nullptr, (const _Float16*)n->stem_h, n->stem.b, nb ? n->blocks[0].s1 : n->s_end,
                                   nb ? n->blocks[0].t1 : n->t_end, M, nblk_h2, (const float*)nullptr, n->range);
            else
                hipLaunchKernelGGL((k_conv3x3_h2<S, 64, F, 4>), dim3(grid_h2), dim3(256), 0, st, (const _Float16*)n->x0h, x, n->act16,
                                   (const float*)nullptr, (const _Float16*)n->stem_h, n->stem.b, nb ? n->blocks[0].s1 : n->s_end,
                                   nb ? n->blocks[0].t1 : n->t_end, M, nblk_h2, (const float*)nullptr, n->range);
            for (size_t i = 0; i < nb; ++i) {
                const BlockW& b = n->blocks[i];
                const bool last = i + 1 == nb;
                const float* sn = last ? n->s_end : n->blocks[i + 1].s1;          // the last block activates for the head conv
                const float* tn = last ? n->t_end : n->blocks[i + 1].t1;
                _Float16* const a16 = n->act16;
                { ProfScope ps(n, st, conv_flops);
                  hipLaunchKernelGGL((k_conv3x3_h2<S, F, F, 0>), dim3(grid_h2), dim3(256), 0, st, (const _Float16*)n->act16,
                                     (float*)nullptr, n->h16, (const float*)nullptr, b.h1, b.c1.b, (const float*)nullptr, (const float*)nullptr, M, nblk_h2,
                                     (const float*)nullptr, n->range); }
                { ProfScope ps(n, st, conv_flops);
                  if (r16)
                      hipLaunchKernelGGL((k_conv3x3_h2<S, F, F, 1, true>), dim3(grid_h2), dim3(256), 0, st, (const _Float16*)n->h16,
                                         last ? (float*)nullptr : y, a16, (const float*)x, b.h2, b.c2.b, sn, tn, M, nblk_h2, (const float*)nullptr, n->range);
                  else
                      hipLaunchKernelGGL((k_conv3x3_h2<S, F, F, 1>), dim3(grid_h2), dim3(256), 0, st, (const _Float16*)n->h16,
                                         last ? (float*)nullptr : y, a16, (const float*)x, b.h2, b.c2.b, sn, tn, M, nblk_h2, (const float*)nullptr, n->range); }
                float* t = x; x = y; y = t;
            }
            hipLaunchKernelGGL((k_head_h<S, F>), dim3(grid_h), dim3(256), 0, st, (const _Float16*)n->act16, n->hc,
                               (const _Float16*)n->head_h, n->head.b, M);
            hipLaunchKernelGGL((k_heads<S>), dim3((rows + HeadRows<S>::N - 1) / HeadRows<S>::N), dim3(256), 0, st, (const float*)n->hc, (const float*)n->hc, n->w_vo, n->b_vo,
                               n->w_v, n->b_v, n->w_o, n->b_o, n->w_a, n->b_a, policy, value, own, rows);
            TG_HIP(ctx, hipGetLastError());
            return TG_OK;
        }
        if (n->dma && (long long)M * F * 4 < (1ll << 31)) {
            // Prologue-free chain over the layer program: every producer (stem, the second conv of a residual block, an attention
            // block) also writes relu(bn1_next(.)) slice-major for a residual block that follows, so the DMA-fed F->F kernels never
            // activate anything.  Attention layers read and write the row-major residual stream (the reference's shipped
            // MainNetwork, "RARRRARRRRAR+P", runs its nine residual blocks on k_conv3x3_sg this way).
            const size_t nl = n->layers.size();
            auto next_bn = [&](size_t i, const float** sn, const float** tn) -> bool {   // layer i+1 is a residual block?
                if (i + 1 < nl && n->layers[i + 1].kind == 0) { const BlockW& nb2 = n->blocks[n->layers[i + 1].ridx]; *sn = nb2.s1; *tn = nb2.t1; return true; }
                *sn = nullptr; *tn = nullptr; return false;
            };
            const float* s0 = nullptr; const float* t0 = nullptr;
            const bool act0 = nl && n->layers[0].kind == 0;
            if (act0) { s0 = n->blocks[n->layers[0].ridx].s1; t0 = n->blocks[n->layers[0].ridx].t1; }
            if (n->NB) (void)hipMemsetAsync(n->tile_ctr, 0, sizeof(int) * 2 * n->NB, st);     // dynamic tile counters of the conv launches
            hipLaunchKernelGGL((k_conv3x3<S, 16, F, false, 0, 9, 2, true>), dim3(grid), dim3(256), 0, st, (const float*)n->x0, x,
                               (const float*)nullptr, n->stem.w, n->stem.b, (const float*)nullptr, (const float*)nullptr, M,
                               act0 ? n->bufAct : (float*)nullptr, s0, t0);
            constexpr int WQ = F / 4 + F / 4 + F;
            auto attention_fast = [&](const AttW& a, const float* xin, float* xout, const float* ps, const float* pt, float* o2,
                                      const float* sn, const float* tn) {
                if (ps)
                    hipLaunchKernelGGL((k_conv3x3<S, F, WQ, true, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                                       (const float*)nullptr, a.qkv.w, a.qkv.b, ps, pt, M);
                else
                    hipLaunchKernelGGL((k_conv3x3<S, F, WQ, false, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                                       (const float*)nullptr, a.qkv.w, a.qkv.b, (const float*)nullptr, (const float*)nullptr, M);
                if constexpr (S == 9)
                    hipLaunchKernelGGL((k_attention_mfma<S, F, 2>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float*)n->bufQ, xin, xout,
                                       o2, a.gamma, a.s, a.t, ps, pt, sn, tn, rows);
            };
            for (size_t i = 0; i < nl; ++i) {
                const Layer& L = n->layers[i];
                const float* sn; const float* tn;
                const bool act = next_bn(i, &sn, &tn);
                if (L.kind == 1) {
                    attention_fast(L.a, x, y, nullptr, nullptr, act ? n->bufAct : (float*)nullptr, sn, tn);
                    float* t = x; x = y; y = t;
                    continue;
                }
                const BlockW& b = n->blocks[L.ridx];
                // tile shape per launch (F = 128): rows one full round of resident workgroups covers = 768 x 192 or 1024 x 128; take the
                // shape whose rounded-up round count wastes fewer rows (ties: the larger tile)
                bool small_tiles = false;
                if constexpr (F == 128) {
                    const long long r3 = 768LL * 192, r2 = 1024LL * 128;
                    small_tiles = ((M + r2 - 1) / r2) * r2 < ((M + r3 - 1) / r3) * r3;
                }
                const int SD_TM = F == 128 ? (small_tiles ? 128 : 192) : 128;
                const int ntile_sd = (M + SD_TM - 1) / SD_TM, slots_sd = F == 128 ? ntile_sd : 512;  // F=256: 2 resident workgroups x 256 CUs walk a dynamic tile list
                int grid_sd = ntile_sd < slots_sd ? ntile_sd : slots_sd;
                if (TG_SG_XCD && F == 128) grid_sd = (grid_sd + 7) / 8 * 8;
                int* const ctr1 = n->tile_ctr + 2 * L.ridx; int* const ctr2 = ctr1 + 1;              // zeroed at the top of the forward
                float* const actn = act ? n->bufAct : (float*)nullptr;
                { ProfScope ps(n, st, conv_flops);
                  if (small_tiles)
                      hipLaunchKernelGGL((k_conv3x3_sg<S, F, 0, 2>), dim3(grid_sd), dim3(256), 0, st, (const float*)n->bufAct, n->bufH,
                                         (const float*)nullptr, b.g1, b.c1.b, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, ctr1);
                  else
                      hipLaunchKernelGGL((k_conv3x3_sg<S, F, 0>), dim3(grid_sd), dim3(256), 0, st, (const float*)n->bufAct, n->bufH,
                                         (const float*)nullptr, b.g1, b.c1.b, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, ctr1); }
                { ProfScope ps(n, st, conv_flops);
                  if (small_tiles)
                      hipLaunchKernelGGL((k_conv3x3_sg<S, F, 1, 2>), dim3(grid_sd), dim3(256), 0, st, (const float*)n->bufH, y,
                                         (const float*)x, b.g2, b.c2.b, actn, sn, tn, M, ctr2);
                  else
                      hipLaunchKernelGGL((k_conv3x3_sg<S, F, 1>), dim3(grid_sd), dim3(256), 0, st, (const float*)n->bufH, y,
                                         (const float*)x, b.g2, b.c2.b, actn, sn, tn, M, ctr2); }
                float* t = x; x = y; y = t;
            }
            // bufAct / bufH are slice-major; the head convs read the row-major residual stream and activate it while staging
            head_conv(x, n->hc, n->head_g, n->head, n->s_end, n->t_end);
            const float* hca = n->hc;
            if (n->pol_att) {                              // attention in the policy head (model.py:72,106-107) on relu(bn_end(x))
                attention_fast(n->patt, x, y, n->s_end, n->t_end, (float*)nullptr, (const float*)nullptr, (const float*)nullptr);
                head_conv(y, n->hca, n->head_ag, n->head_a, nullptr, nullptr);
                hca = n->hca;
            }
            hipLaunchKernelGGL((k_heads<S>), dim3((rows + HeadRows<S>::N - 1) / HeadRows<S>::N), dim3(256), 0, st, (const float*)n->hc, hca, n->w_vo, n->b_vo,
                               n->w_v, n->b_v, n->w_o, n->b_o, n->w_a, n->b_a, policy, value, own, rows);
            TG_HIP(ctx, hipGetLastError());
            return TG_OK;
        }
    }
    hipLaunchKernelGGL((k_conv3x3<S, 16, F, false, 0>), dim3(grid), dim3(256), 0, st, (const float*)n->x0, x,
                       (const float*)nullptr, n->stem.w, n->stem.b, (const float*)nullptr, (const float*)nullptr, M);
    // attention block: fused q|k|v 1x1 projection on the matrix cores, then the per-board core (model.py:301-315)
    constexpr int W = F / 4 + F / 4 + F;
    constexpr size_t att_lds = sizeof(float) * ((size_t)P * W + (size_t)P * (P + 1));
    auto attention = [&](const AttW& a, const float* xin, float* xout, const float* ps, const float* pt) -> int {
        if (att_lds > 160 * 1024) return -1;
        if (ps)
            hipLaunchKernelGGL((k_conv3x3<S, F, W, true, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                               (const float*)nullptr, a.qkv.w, a.qkv.b, ps, pt, M);
        else
            hipLaunchKernelGGL((k_conv3x3<S, F, W, false, 2, 1>), dim3(grid), dim3(256), 0, st, xin, n->bufQ,
                               (const float*)nullptr, a.qkv.w, a.qkv.b, (const float*)nullptr, (const float*)nullptr, M);
        hipLaunchKernelGGL((k_attention<S, F>), dim3(rows), dim3(256), att_lds, st, (const float*)n->bufQ, xin, xout,
                           a.gamma, a.s, a.t, ps, pt);
        return 0;
    };
    for (const Layer& L : n->layers) {
        if (L.kind == 1) {
            if (attention(L.a, x, y, nullptr, nullptr)) TG_FAIL(ctx, TG_ERR_ARG, "attention blocks need P*(1.5F+P+1) floats of LDS: not available at this board size");
            float* t = x; x = y; y = t;
            continue;
        }
        const BlockW& b = n->blocks[L.ridx];
        { ProfScope ps(n, st, conv_flops);
          hipLaunchKernelGGL((k_conv3x3<S, F, F, true, 0, 9, NPT>), dim3(grid_f), dim3(256), 0, st, (const float*)x, n->bufH,
                             (const float*)nullptr, b.c1.w, b.c1.b, b.s1, b.t1, M);
        }
        { ProfScope ps(n, st, conv_flops);
          hipLaunchKernelGGL((k_conv3x3<S, F, F, false, 1, 9, NPT>), dim3(grid_f), dim3(256), 0, st, (const float*)n->bufH, y,
                             (const float*)x, b.c2.w, b.c2.b, (const float*)nullptr, (const float*)nullptr, M);
        }
        float* t = x; x = y; y = t;
    }
    // heads: value/ownership conv reads relu(bn_end(x)) (model.py:94,97); the policy conv reads the same tensor, or its
    // Self_Attention when the architecture has attention_act (model.py:72,106-107)
    head_conv(x, n->hc, n->head_g, n->head, n->s_end, n->t_end);
    const float* hca = n->hc;
    if (n->pol_att) {
        if (attention(n->patt, x, y, n->s_end, n->t_end)) TG_FAIL(ctx, TG_ERR_ARG, "attention policy head: not enough LDS at this board size");
        head_conv(y, n->hca, n->head_ag, n->head_a, nullptr, nullptr);
        hca = n->hca;
    }
    hipLaunchKernelGGL((k_heads<S>), dim3((rows + HeadRows<S>::N - 1) / HeadRows<S>::N), dim3(256), 0, st, (const float*)n->hc, hca, n->w_vo, n->b_vo, n->w_v, n->b_v,
                       n->w_o, n->b_o, n->w_a, n->b_a, policy, value, own, rows);
    TG_HIP(ctx, hipGetLastError());
    return TG_OK;
}

int adopt_pending(tg_ctx* ctx, Net* n, bool wait);

// A finished background refresh takes over only where the CALLER is at a boundary: tg_net_predict (every call stands alone) and
// tg_sp_begin_move (tg_net_adopt_ready) -- never in the middle of a move's search, so one search tree and one recorded pi never
// mix evaluations of two weight sets and the switch point does not depend on upload timing within a move.
int forward(tg_ctx* ctx, Net* n, const float* obs, int rows, float* policy, float* value, float* own) {
    if (rows <= 0) return TG_OK;
    if (rows > n->rows_cap) TG_FAIL(ctx, TG_ERR_ARG, "network batch larger than the allocated activation buffers");
#define TG_NET_CASE(SZ, FF) if (n->S == SZ && n->F == FF) return forward_t<SZ, FF>(ctx, n, obs, rows, policy, value, own)
    TG_NET_CASE(9, 32); TG_NET_CASE(9, 64); TG_NET_CASE(9, 128); TG_NET_CASE(9, 256);
    TG_NET_CASE(19, 128); TG_NET_CASE(19, 256);
#undef TG_NET_CASE
    TG_FAIL(ctx, TG_ERR_ARG, "unsupported (board_size, net_filters): built are 9x{32,64,128,256}, 19x{128,256}");
}

// arch: one letter per trunk layer, 'R' residual block / 'A' Self_Attention block, optional "+P" = attention in the policy
// head.  The reference MainNetwork (model.py:49-76) is "RARRRARRRRAR+P"; BASELINE's N-block tower is N x 'R'.
bool parse_arch(const std::string& arch, std::string* trunk, bool* pol) {
    *pol = false; *trunk = arch;
    const size_t plus = arch.find('+');
    if (plus != std::string::npos) {
        if (arch.substr(plus) != "+P") return false;
        *pol = true; *trunk = arch.substr(0, plus);
    }
    for (char c : *trunk) if (c != 'R' && c != 'A') return false;
    return true;
}

size_t expected_floats(int S, int C, int F, const std::string& arch) {
    const size_t P = (size_t)S * S, A = P + 1, Wq = (size_t)F / 4 * 2 + F;
    (void)C;
    std::string trunk; bool pol;
    if (!parse_arch(arch, &trunk, &pol)) return 0;
    const size_t att = Wq * F + Wq + 1 + 2 * (size_t)F;
    size_t n = 9 * (size_t)F * 16 + F;
    for (char c : trunk) n += (c == 'R') ? (2 * (size_t)F + 2 * (9 * (size_t)F * F + F)) : att;
    n += 2 * (size_t)F;
    if (pol) n += att + 9 * 16 * (size_t)F + 16;
    n += 9 * 16 * (size_t)F + 16;
    n += 2 * P * 64 + 64 + 64 + 1 + 64 * P + P + 4 * P * A + A;
    return n;
}

// Bind the pointer fields of `n` to weight set `k` (blob sections in the order transgo_amd/model.py:pack_weights writes them).
void bind_weights(Net* n, int k) {
    const int F = n->F; const size_t P = n->P, A = n->A, Wq = (size_t)F / 4 * 2 + F;
    const Net::WeightSet& w = n->sets[k];
    n->blob = w.blob; n->wstage = w.wstage; n->wh = w.wh; n->stem_h = w.stem_h; n->head_h = w.head_h; n->wsc = w.wsc; n->head_g = w.head_g; n->head_ag = w.head_ag;
    n->head_x2 = w.head_x2; n->head_x2sc = w.head_x2sc;
    std::string trunk; bool pol = false;
    parse_arch(n->arch, &trunk, &pol);
    const float* p = w.blob;
    auto take = [&](size_t c) { const float* q = p; p += c; return q; };
    auto take_att = [&](AttW& a) { a.qkv.w = take(Wq * F); a.qkv.b = take(Wq); a.gamma = take(1); a.s = take(F); a.t = take(F); };
    n->stem.w = take(9 * (size_t)F * 16); n->stem.b = take(F);
    n->blocks.assign(n->NB, BlockW{});
    n->layers.clear();
    const size_t per = 9 * (size_t)F * F;
    int ri = 0, ai = 0;
    int n_att = pol ? 1 : 0; for (char c : trunk) n_att += c == 'A';
    const size_t att_img = (size_t)F / 16 * Wq * 32;                  // halfs per attention layer (k_attention_x3's LDS image)
    auto bind_x3 = [&](AttW& a) { if (w.att_h) { a.x3w = w.att_h + (size_t)ai * att_img; a.x3sc = w.att_sc + ai; } ++ai; };
    for (char c : trunk) {
        Layer L; L.kind = c == 'A'; L.ridx = -1; L.a = AttW{};
        if (c == 'R') {
            BlockW& b = n->blocks[ri];
            b.s1 = take(F); b.t1 = take(F);
            b.c1.w = take(per); b.c1.b = take(F);
            b.c2.w = take(per); b.c2.b = take(F);
            b.g1 = w.wstage ? w.wstage + (size_t)(2 * ri) * per : nullptr; b.g2 = w.wstage ? w.wstage + (size_t)(2 * ri + 1) * per : nullptr;
            const size_t perh = n->prec == 3 ? 2 * per : per;           // split precision: hi + lo halves of every weight
            b.h1 = w.wh ? w.wh + (size_t)(2 * ri) * perh : nullptr; b.h2 = w.wh ? w.wh + (size_t)(2 * ri + 1) * perh : nullptr;
            L.ridx = ri++;
        } else {
            take_att(L.a); bind_x3(L.a);
        }
        n->layers.push_back(L);
    }
    n->s_end = take(F); n->t_end = take(F);
    if (pol) { take_att(n->patt); bind_x3(n->patt); n->head_a.w = take(9 * 16 * (size_t)F); n->head_a.b = take(16); }
    (void)n_att;
    n->head.w = take(9 * 16 * (size_t)F); n->head.b = take(16);
    n->w_vo = take(2 * P * 64); n->b_vo = take(64); n->w_v = take(64); n->b_v = take(1);
    n->w_o = take(64 * P); n->b_o = take(P); n->w_a = take(4 * P * A); n->b_a = take(A);
}

// Upload `blob` into weight set k and rebuild its stage-ordered copies, everything on `st` (no synchronisation here).
int fill_weight_set(tg_ctx* ctx, Net* n, int k, const float* blob, hipStream_t st, bool blob_on_device = false) {
    const int F = n->F;
    TG_HIP(ctx, hipMemcpyAsync(n->sets[k].blob, blob, sizeof(float) * n->blob_floats,
                               blob_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (blob_on_device) {
        // the library keeps no caller pointer after a call returns: the copy out of the caller's buffer is over before the
        // (asynchronous) restaging starts
        TG_HIP(ctx, hipStreamSynchronize(st));
    }
    TG_HIP(ctx, hipMemsetAsync(n->range + 1 + k, 0, sizeof(unsigned), st));
    hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, (const float*)n->sets[k].blob, n->blob_floats, n->range + 1 + k);
    Net view = *n;                                        // pointer fields of set k without disturbing the live binding
    bind_weights(&view, k);
    hipLaunchKernelGGL(k_restage_head, dim3(64), dim3(256), 0, st, view.head.w, n->sets[k].head_g, F);
    if (n->pol_att) hipLaunchKernelGGL(k_restage_head, dim3(64), dim3(256), 0, st, view.head_a.w, n->sets[k].head_ag, F);
    if (n->prec == 3) {
        // split copies (hi | lo of w * 2^s per conv), made on the device from the blob just uploaded; the scale words double as the
        // scratch of the abs-max reduction (bit patterns), overwritten with 2^-s by the restaging kernel that follows in order
        const size_t nconv = 2 * view.blocks.size() + 1;
        unsigned* mx = reinterpret_cast<unsigned*>(n->sets[k].wsc) + nconv;          // second half of the allocation
        TG_HIP(ctx, hipMemsetAsync(mx, 0, sizeof(unsigned) * nconv, st));
        size_t ci = 0;
        for (const BlockW& b : view.blocks) {
            for (int h = 0; h < 2; ++h) {
                const float* w = h ? b.c2.w : b.c1.w;
                hipLaunchKernelGGL(k_absmax, dim3(256), dim3(256), 0, st, w, (size_t)9 * F * F, mx + ci);
                hipLaunchKernelGGL(k_restage_split, dim3(1024), dim3(256), 0, st, w, const_cast<_Float16*>(h ? b.h2 : b.h1), n->sets[k].wsc + ci,
                                   (const unsigned*)(mx + ci), F, F, F / 16, 1);
                ++ci;
            }
        }
        if (n->sets[k].att_h) {                                                       // fused attention layers: split q|k|v weights
            std::vector<const AttW*> atts;
            for (const Layer& L : view.layers) if (L.kind == 1) atts.push_back(&L.a);
            if (n->pol_att) atts.push_back(&view.patt);
            const int Wq = F / 4 * 2 + F;
            unsigned* amx = reinterpret_cast<unsigned*>(n->sets[k].att_sc) + atts.size();
            TG_HIP(ctx, hipMemsetAsync(amx, 0, sizeof(unsigned) * atts.size(), st));
            for (size_t a = 0; a < atts.size(); ++a) {
                hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, st, atts[a]->qkv.w, (size_t)Wq * F, amx + a);
                hipLaunchKernelGGL(k_restage_att_split, dim3(256), dim3(256), 0, st, atts[a]->qkv.w, const_cast<_Float16*>(atts[a]->x3w),
                                   const_cast<float*>(atts[a]->x3sc), (const unsigned*)(amx + a), Wq, F);
            }
        }
        hipLaunchKernelGGL(k_absmax, dim3(64), dim3(256), 0, st, view.stem.w, (size_t)9 * F * 16, mx + ci);
        hipLaunchKernelGGL(k_restage_split, dim3(256), dim3(256), 0, st, view.stem.w, view.stem_h, n->sets[k].wsc + ci, (const unsigned*)(mx + ci),
                           F, 16, 2, 1);                                              // 16 planes = one group, padded to two (an even stage count)
        if (n->sets[k].head_x2) {
            unsigned* hmx = reinterpret_cast<unsigned*>(n->sets[k].head_x2sc) + 1;
            TG_HIP(ctx, hipMemsetAsync(hmx, 0, sizeof(unsigned), st));
            hipLaunchKernelGGL(k_absmax, dim3(16), dim3(256), 0, st, view.head.w, (size_t)9 * 16 * F, hmx);
            hipLaunchKernelGGL(k_restage_head_split, dim3(32), dim3(256), 0, st, view.head.w, n->sets[k].head_x2, n->sets[k].head_x2sc, (const unsigned*)hmx, F);
        }
        TG_HIP(ctx, hipGetLastError());
    } else if (n->prec >= 1) {
        // stage-ordered fp16 copies, converted on the device from the blob just uploaded (round to nearest even)
        for (const BlockW& b : view.blocks) {
            hipLaunchKernelGGL(k_restage_half, dim3(1024), dim3(256), 0, st, b.c1.w, const_cast<_Float16*>(b.h1), F, F, F, 32, 1);   // k_conv3x3_h2: 32-channel stages, paired couts
            hipLaunchKernelGGL(k_restage_half, dim3(1024), dim3(256), 0, st, b.c2.w, const_cast<_Float16*>(b.h2), F, F, F, 32, 1);
        }
        hipLaunchKernelGGL(k_restage_half, dim3(256), dim3(256), 0, st, view.stem.w, view.stem_h, F, 16, 64, 32, 1);
        hipLaunchKernelGGL(k_restage_half, dim3(256), dim3(256), 0, st, view.head.w, view.head_h, 16, F, F, 32, 0);
        TG_HIP(ctx, hipGetLastError());
    } else if (n->dma) {
        // stage-ordered copy for k_conv3x3_sg, [slice*9 + tap][cout][16 channels of the slice], made on the device
        for (const BlockW& b : view.blocks) {
            hipLaunchKernelGGL(k_restage_f32, dim3(1024), dim3(256), 0, st, b.c1.w, const_cast<float*>(b.g1), F);
            hipLaunchKernelGGL(k_restage_f32, dim3(1024), dim3(256), 0, st, b.c2.w, const_cast<float*>(b.g2), F);
        }
        TG_HIP(ctx, hipGetLastError());
    }
    return TG_OK;
}

// A completed asynchronous refresh becomes the live set.  Called at boundaries only -- tg_sp_begin_move (tg_net_adopt_ready), every
// stand-alone tg_net_predict, tg_net_load_poll, and at the top of every load -- never from forward(): a host that drives
// tg_sp_collect / tg_net_forward / tg_sp_absorb itself adopts a refresh through tg_net_load_poll (include/transgo_hip.h).
int adopt_pending(tg_ctx* ctx, Net* n, bool wait) {
    if (!n->pending) return TG_OK;
    if (wait) TG_HIP(ctx, hipEventSynchronize(n->loaded));
    else if (hipEventQuery(n->loaded) != hipSuccess) return TG_OK;     // still in flight: keep searching on the old weights
    n->active ^= 1;
    bind_weights(n, n->active);
    n->pending = false;
    TG_HIP(ctx, hipEventRecord(n->swapped, ctx->stream));               // kernels already queued may still read the retired set
    return TG_OK;
}

}  // namespace

extern "C" {

size_t tg_net_blob_floats(int board_size, int encode_dim, int filters, int blocks) {
    return expected_floats(board_size, encode_dim, filters, std::string((size_t)(blocks > 0 ? blocks : 0), 'R'));
}

size_t tg_net_blob_floats_arch(int board_size, int encode_dim, int filters, const char* arch) {
    return arch ? expected_floats(board_size, encode_dim, filters, arch) : 0;
}

int tg_net_load(tg_ctx* ctx, const float* blob, size_t n_floats, int rows_cap) {
    if (!ctx) return TG_ERR_ARG;
    const std::string arch((size_t)(ctx->cfg.net_blocks > 0 ? ctx->cfg.net_blocks : 0), 'R');
    return tg_net_load_arch(ctx, arch.c_str(), blob, n_floats, rows_cap);
}

int tg_net_load_arch(tg_ctx* ctx, const char* arch_c, const float* blob, size_t n_floats, int rows_cap) {
    if (!ctx || !blob || !arch_c) return TG_ERR_ARG;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int S = ctx->S, C = ctx->cfg.encode_dim, F = ctx->cfg.net_filters;
    const std::string arch(arch_c);
    std::string trunk; bool pol = false;
    if (!parse_arch(arch, &trunk, &pol)) TG_FAIL(ctx, TG_ERR_ARG, "bad architecture string (letters R/A, optional +P)");
    if (C > 16) TG_FAIL(ctx, TG_ERR_ARG, "encode_dim > 16 not supported by the stem kernel");
    if (F % 32 != 0) TG_FAIL(ctx, TG_ERR_ARG, "net_filters must be a multiple of 32");
    if (n_floats != expected_floats(S, C, F, arch)) TG_FAIL(ctx, TG_ERR_ARG, "weight blob size does not match (board_size, filters, architecture)");
    if (!ctx->eng) { ctx->eng = new Engine(); }          // rules-only context + network: a bare evaluator
    Engine* e = ctx->eng;
    if (rows_cap <= 0) rows_cap = e->rows_cap > 0 ? e->rows_cap : 256;
    if (e->rows_cap > rows_cap) rows_cap = e->rows_cap;
    Net* n = e->net;
    const int prec = ctx->cfg.net_precision;
    if (prec < 0 || prec > 3) TG_FAIL(ctx, TG_ERR_ARG, "net_precision: 0 (f32), 1 (fp16 storage, f32 accumulate, f32 residual stream), 2 (fp16 residual stream too) or 3 (split precision: fp16 hi + lo operands, f32 accumulate)");
    if (n && (n->rows_cap < rows_cap || n->arch != arch || n->prec != prec)) { tg_net_destroy(ctx); n = nullptr; }
    const size_t P = (size_t)S * S, A = P + 1, Wq = (size_t)F / 4 * 2 + F;
    const bool any_att = pol || trunk.find('A') != std::string::npos;
    int NB = 0; for (char c : trunk) NB += c == 'R';
    if (prec >= 1 && ((any_att && !(prec == 3 && S == 9)) || (F != 128 && F != 256)))
        TG_FAIL(ctx, TG_ERR_ARG, "net_precision 1 / 2 (fp16 matrix cores) is built for attention-free towers with 128 or 256 filters; 3 (split precision) also takes attention layers at 9x9");
    if (!n) {
        n = new Net();
        e->net = n;
        n->S = S; n->P = (int)P; n->A = (int)A; n->C = C; n->F = F; n->NB = NB; n->rows_cap = rows_cap; n->arch = arch; n->pol_att = pol;
        n->blob_floats = n_floats; n->prec = prec;
        const size_t act = sizeof(float) * (size_t)rows_cap * P * F;
        TG_HIP(ctx, hipMalloc((void**)&n->bufA, act));
        TG_HIP(ctx, hipMalloc((void**)&n->bufB, act));
        TG_HIP(ctx, hipMalloc((void**)&n->bufH, act));
        TG_HIP(ctx, hipMalloc((void**)&n->x0, sizeof(float) * (size_t)rows_cap * P * 16));
        TG_HIP(ctx, hipMalloc((void**)&n->hc, sizeof(float) * (size_t)rows_cap * P * 16));
        TG_HIP(ctx, hipMalloc((void**)&n->own, sizeof(float) * (size_t)rows_cap * P));
        TG_HIP(ctx, hipMalloc((void**)&n->range, sizeof(unsigned) * 4));
        TG_HIP(ctx, hipMemset(n->range, 0, sizeof(unsigned) * 4));

        if (any_att) TG_HIP(ctx, hipMalloc((void**)&n->bufQ, sizeof(float) * (size_t)rows_cap * P * Wq));
        if (pol) TG_HIP(ctx, hipMalloc((void**)&n->hca, sizeof(float) * (size_t)rows_cap * P * 16));
        // DMA-fed F->F chain: f32 towers of 128 / 256 filters, with or without attention layers (those need the 9x9 MFMA kernel)
        n->dma = ((F == 128 || F == 256) && (!any_att || S == 9)) ? (getenv("TG_DMA_CONV") ? (atoi(getenv("TG_DMA_CONV")) != 0) : 1) : 0;
        if (prec >= 1) n->dma = 0;
        const size_t wcopy = (size_t)(NB > 0 ? 2 * NB : 1) * 9 * F * F;
        if (n->dma) TG_HIP(ctx, hipMalloc((void**)&n->bufAct, act));
        if (n->dma) TG_HIP(ctx, hipMalloc((void**)&n->tile_ctr, sizeof(int) * (size_t)(NB > 0 ? 2 * NB : 1)));
        for (Net::WeightSet& w : n->sets) {
            TG_HIP(ctx, hipMalloc((void**)&w.blob, sizeof(float) * n_floats));
            TG_HIP(ctx, hipMalloc((void**)&w.head_g, sizeof(float) * 64 * (size_t)F));
            if (pol) TG_HIP(ctx, hipMalloc((void**)&w.head_ag, sizeof(float) * 64 * (size_t)F));
            if (n->dma) TG_HIP(ctx, hipMalloc((void**)&w.wstage, sizeof(float) * wcopy));
            if (prec >= 1) {
                TG_HIP(ctx, hipMalloc((void**)&w.wh, sizeof(_Float16) * wcopy * (prec == 3 ? 2 : 1)));
                if (prec == 3) TG_HIP(ctx, hipMalloc((void**)&w.wsc, sizeof(float) * 2 * (size_t)(2 * NB + 1)));
                if (prec == 3 && F == 128) {                                         // split-precision head conv (k_head_gemm_x2)
                    TG_HIP(ctx, hipMalloc((void**)&w.head_x2, sizeof(_Float16) * 64 * (size_t)F * 2));
                    TG_HIP(ctx, hipMalloc((void**)&w.head_x2sc, sizeof(float) * 2));
                }
                if (prec == 3 && any_att && F == 128 && S == 9) {                    // k_attention_x3 (its weight image must fit LDS)
                    size_t n_att = pol ? 1 : 0; for (char c : trunk) n_att += c == 'A';
                    TG_HIP(ctx, hipMalloc((void**)&w.att_h, sizeof(_Float16) * n_att * ((size_t)F / 16 * Wq * 32)));
                    TG_HIP(ctx, hipMalloc((void**)&w.att_sc, sizeof(float) * 2 * n_att));
                }
                TG_HIP(ctx, hipMalloc((void**)&w.stem_h, sizeof(_Float16) * 9 * (size_t)F * 64));
                TG_HIP(ctx, hipMalloc((void**)&w.head_h, sizeof(_Float16) * 9 * 16 * (size_t)F));
            }
        }
        if (prec >= 1) {
            TG_HIP(ctx, hipMalloc((void**)&n->act16, prec == 3 ? act : act / 2));     // split precision: hi + lo per element
            TG_HIP(ctx, hipMalloc((void**)&n->h16, prec == 3 ? act : act / 2));
            TG_HIP(ctx, hipMalloc((void**)&n->x0h, sizeof(_Float16) * (size_t)rows_cap * P * 64));
        }
        TG_HIP(ctx, hipStreamCreateWithFlags(&n->side, hipStreamNonBlocking));
        TG_HIP(ctx, hipEventCreateWithFlags(&n->loaded, hipEventDisableTiming));
        TG_HIP(ctx, hipEventCreateWithFlags(&n->swapped, hipEventDisableTiming));
        TG_HIP(ctx, hipEventRecord(n->swapped, ctx->stream));
        TG_HIP(ctx, hipHostMalloc((void**)&n->pinned, sizeof(float) * n_floats, hipHostMallocDefault));
        n->active = 1;                                    // the load below fills set 0 and makes it the live one
        if (prec == 3 && any_att && F == 128 && S == 9) {
            const int lds3 = (int)(sizeof(_Float16) * (size_t)F / 16 * Wq * 32 + sizeof(float) * (Wq + 6 * (size_t)F) + 4096);
            TG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_x3<9, 128, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
            TG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_x3<9, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds3));
        }
        if (any_att) {
            const size_t lds = sizeof(float) * (P * Wq + P * (P + 1));
            if (lds <= 160 * 1024) {
                hipError_t er = hipSuccess;
#define TG_ATT_ATTR(SZ, FF) if (S == SZ && F == FF) er = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention<SZ, FF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                TG_ATT_ATTR(9, 32); TG_ATT_ATTR(9, 64); TG_ATT_ATTR(9, 128); TG_ATT_ATTR(9, 256);
#undef TG_ATT_ATTR
                TG_HIP(ctx, er);
            }
        }
    }
    // Synchronous refresh (trainer.py:76-79 -> self_play.py:913): an asynchronous one still in flight is adopted first, then the
    // retired set is filled on the context's own stream (in order behind every kernel that reads it) and becomes the live set.
    { int rc = adopt_pending(ctx, n, /*wait=*/true); if (rc) return rc; }
    const int k = n->active ^ 1;
    { int rc = fill_weight_set(ctx, n, k, blob, ctx->stream); if (rc) return rc; }
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    n->active = k;
    bind_weights(n, k);
    return TG_OK;
}

// The refresh that never stalls a search: the blob is copied to pinned memory, uploaded and re-staged into the
// retired weight set on a side stream while the context's stream keeps searching on the live set; the switch happens at the next
// move boundary (tg_sp_begin_move), stand-alone tg_net_predict or tg_net_load_poll that finds the upload complete -- never inside a
// search.  Needs a network already loaded with the same architecture (else: synchronous).
int tg_net_load_async(tg_ctx* ctx, const char* arch_c, const float* blob, size_t n_floats) {
    if (!ctx || !blob || !arch_c) return TG_ERR_ARG;
    Net* n = ctx->eng ? ctx->eng->net : nullptr;
    if (!n || n->arch != arch_c || n->blob_floats != n_floats || n->prec != ctx->cfg.net_precision)
        return tg_net_load_arch(ctx, arch_c, blob, n_floats, 0);
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    { int rc = adopt_pending(ctx, n, /*wait=*/true); if (rc) return rc; }     // at most one refresh in flight
    memcpy(n->pinned, blob, sizeof(float) * n_floats);
    TG_HIP(ctx, hipStreamWaitEvent(n->side, n->swapped, 0));                 // nothing queued before the last switch reads this set now
    { int rc = fill_weight_set(ctx, n, n->active ^ 1, n->pinned, n->side); if (rc) return rc; }
    TG_HIP(ctx, hipEventRecord(n->loaded, n->side));
    n->pending = true;
    return TG_OK;
}

// The same with the blob already in THIS GPU's memory (what an RCCL broadcast delivered): device -> device into the retired set,
// no host bounce.  The contents of d_blob must be complete when the call is made (synchronise the producing stream first); the
// buffer may be reused as soon as the call returns.
int tg_net_load_async_dev(tg_ctx* ctx, const char* arch_c, const float* d_blob, size_t n_floats) {
    if (!ctx || !d_blob || !arch_c) return TG_ERR_ARG;
    Net* n = ctx->eng ? ctx->eng->net : nullptr;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (!n || n->arch != arch_c || n->blob_floats != n_floats || n->prec != ctx->cfg.net_precision) {
        // nothing to refresh yet (first weights of this context, or another architecture): the synchronous load, through the host
        std::vector<float> host(n_floats);
        TG_HIP(ctx, hipMemcpy(host.data(), d_blob, sizeof(float) * n_floats, hipMemcpyDeviceToHost));
        return tg_net_load_arch(ctx, arch_c, host.data(), n_floats, 0);
    }
    { int rc = adopt_pending(ctx, n, /*wait=*/true); if (rc) return rc; }
    TG_HIP(ctx, hipStreamWaitEvent(n->side, n->swapped, 0));
    { int rc = fill_weight_set(ctx, n, n->active ^ 1, d_blob, n->side, /*blob_on_device=*/true); if (rc) return rc; }
    TG_HIP(ctx, hipEventRecord(n->loaded, n->side));
    n->pending = true;
    return TG_OK;
}

// pending = 1 while an asynchronous refresh has not been adopted yet; wait != 0 blocks until it is (and adopts it).
int tg_net_load_poll(tg_ctx* ctx, int wait, int* pending) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_ERR_STATE;
    Net* n = ctx->eng->net;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    int rc = adopt_pending(ctx, n, wait != 0);
    if (pending) *pending = n->pending ? 1 : 0;
    return rc;
}

#ifdef TG_ATT_STAMP
int tg_debug_att_stamps(unsigned long long* out, int n) {
    if (n > 32) n = 32;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_att_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

void tg_net_destroy(tg_ctx* ctx) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return;
    Net* n = ctx->eng->net;
    if (n->pending) (void)hipEventSynchronize(n->loaded);
    void* ptrs[] = {n->bufA, n->bufB, n->bufH, n->x0, n->hc, n->own, n->bufQ, n->hca, n->bufAct, n->act16, n->h16, n->x0h, n->tile_ctr, n->range};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (Net::WeightSet& w : n->sets) { void* q[] = {w.blob, w.wstage, w.wh, w.stem_h, w.head_h, w.wsc, w.head_g, w.head_ag, w.att_h, w.att_sc, w.head_x2, w.head_x2sc}; for (void* p : q) if (p) (void)hipFree(p); }
    if (n->side) (void)hipStreamDestroy(n->side);
    if (n->loaded) (void)hipEventDestroy(n->loaded);
    if (n->swapped) (void)hipEventDestroy(n->swapped);
    if (n->pinned) (void)hipHostFree(n->pinned);
    for (hipEvent_t ev : n->ev) (void)hipEventDestroy(ev);
    delete n;
    ctx->eng->net = nullptr;
}

// Move boundary of the engine (tg_sp_begin_move): a completed background refresh becomes the live set here, and only here.
int tg_net_adopt_ready(tg_ctx* ctx) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_OK;
    return adopt_pending(ctx, ctx->eng->net, /*wait=*/false);
}

int tg_net_forward(tg_ctx* ctx, int rows) {
    Engine* e = ctx->eng;
    if (!e || !e->net) TG_FAIL(ctx, TG_ERR_STATE, "no network weights loaded (tg_net_load)");
    Net* n = e->net;
    n->in_bits = e->dev.obs_bits; n->in_slot = e->dev.row_slot; n->in_words = e->dev.obs_words;
    const int rc = forward(ctx, n, nullptr, rows, e->dev.policy, e->dev.value, nullptr);
    n->in_bits = nullptr; n->in_slot = nullptr;
    return rc;
}

// main_prediction on host buffers (model.py:17-20): obs f32[n][C][S][S] -> policy f32[n][A], value f32[n], own f32[n][P]
int tg_net_predict(tg_ctx* ctx, const float* obs, int n_rows, float* policy, float* value, float* own) {
    if (!ctx || !ctx->eng || !ctx->eng->net) { if (ctx) ctx->err = "no network weights loaded (tg_net_load)"; return TG_ERR_STATE; }
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    Net* n = ctx->eng->net;
    { int rc = adopt_pending(ctx, n, /*wait=*/false); if (rc) return rc; }      // a finished background refresh takes over between calls
    const size_t P = n->P, A = n->A, C = n->C;
    int done = 0;
    tg::DevBuf& din = ctx->env_f32; tg::DevBuf& dout = ctx->env_in;
    while (done < n_rows) {
        const int k = (n_rows - done) < n->rows_cap ? (n_rows - done) : n->rows_cap;
        if (din.reserve(sizeof(float) * k * C * P) || dout.reserve(sizeof(float) * k * (A + 1)))
            TG_FAIL(ctx, TG_ERR_HIP, "hipMalloc failed (predict scratch)");
        float* d_pol = (float*)dout.p; float* d_val = d_pol + (size_t)k * A;
        TG_HIP(ctx, hipMemcpyAsync(din.p, obs + (size_t)done * C * P, sizeof(float) * k * C * P, hipMemcpyHostToDevice, ctx->stream));
        int rc = forward(ctx, n, (const float*)din.p, k, d_pol, d_val, own ? n->own : nullptr);
        if (rc) return rc;
        TG_HIP(ctx, hipMemcpyAsync(policy + (size_t)done * A, d_pol, sizeof(float) * k * A, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipMemcpyAsync(value + done, d_val, sizeof(float) * k, hipMemcpyDeviceToHost, ctx->stream));
        if (own) TG_HIP(ctx, hipMemcpyAsync(own + (size_t)done * P, n->own, sizeof(float) * k * P, hipMemcpyDeviceToHost, ctx->stream));
        TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        done += k;
    }
    return TG_OK;
}


// Range report (the reference runs f32 end to end, model.py:79-114, and hands over arbitrary trained checkpoints, model.py:23-27;
// the fp16-carrying precisions do not have f32's range).  fp16_overflows: sticky count, since the network was created, of epilogue
// tiles (convs) / boards (attention) in which a value beyond +-65504 was rounded to fp16 -- from there on the network computes
// inf / NaN; 0 = every forward pass so far stayed in range.  Always 0 with net_precision 0.  weight_absmax: largest |w| of the
// live weight set's BN-folded blob (inf / NaN if the blob holds one).  A non-zero count also leaves a message in tg_last_error.
int tg_net_range(tg_ctx* ctx, uint64_t* fp16_overflows, float* weight_absmax) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_ERR_STATE;
    Net* n = ctx->eng->net;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    unsigned host[4] = {0, 0, 0, 0};
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    TG_HIP(ctx, hipMemcpy(host, n->range, sizeof(host), hipMemcpyDeviceToHost));
    if (fp16_overflows) *fp16_overflows = host[0];
    if (weight_absmax) { float f; memcpy(&f, &host[1 + n->active], sizeof(f)); *weight_absmax = f; }
    if (host[0])
        ctx->err = "fp16 overflow: " + std::to_string(host[0]) + " epilogue tiles rounded a value beyond +-65504 to fp16 (net_precision " +
                   std::to_string(n->prec) + "); outputs since then may be inf/NaN -- use net_precision 0 for these weights";
    return TG_OK;
}

// HIP-event timing of the dominant kernel (3x3 conv F->F), measured on the stream the kernels are launched on.
int tg_prof_enable(tg_ctx* ctx, int on, int max_launches) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_ERR_STATE;
    Net* n = ctx->eng->net;
    TG_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (on) {
        const size_t want = 2 * (size_t)(max_launches > 0 ? max_launches : 4096);
        while (n->ev.size() < want) { hipEvent_t ev; TG_HIP(ctx, hipEventCreate(&ev)); n->ev.push_back(ev); }
        n->ev_used = 0; n->conv_ms = 0; n->conv_launches = 0; n->conv_flops = 0; n->conv_skipped = 0;
    }
    n->prof = on != 0;
    return TG_OK;
}

int tg_prof_read(tg_ctx* ctx, double* conv_ms, int64_t* conv_launches, double* conv_flops) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_ERR_STATE;
    Net* n = ctx->eng->net;
    TG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0;
    for (size_t i = 0; i + 1 < n->ev_used; i += 2) {
        float t = 0; TG_HIP(ctx, hipEventElapsedTime(&t, n->ev[i], n->ev[i + 1])); ms += t;
    }
    n->conv_ms += ms; n->ev_used = 0;
    if (conv_ms) *conv_ms = n->conv_ms;
    if (conv_launches) *conv_launches = n->conv_launches;
    if (conv_flops) *conv_flops = n->conv_flops;
    return TG_OK;
}

// Launches of the dominant kernel that were NOT timed since tg_prof_enable(ctx, 1, n) because the event pool was full
// (tg_prof_read drains the pool into the running totals; 0 here = the totals cover every launch).
int tg_prof_skipped(tg_ctx* ctx, int64_t* launches) {
    if (!ctx || !ctx->eng || !ctx->eng->net) return TG_ERR_STATE;
    if (launches) *launches = ctx->eng->net->conv_skipped;
    return TG_OK;
}

}  // extern "C"
