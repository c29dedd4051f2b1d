// rng_host.cpp -- host-side MT19937 in NumPy's legacy RandomState semantics (tg_host_* of include/transgo_hip.h).
//
// The reference draws all randomness from the global legacy NumPy stream: np.random.dirichlet([0.03]*n) once per move
// (self_play.py:93), np.random.choice(list) for PUCT ties (self_play.py:709) and np.random.choice(A, p=) for the move
// (self_play.py:683).  Bit-exact visit counts need the same stream, so this file restates, from NumPy's published
// algorithm (numpy/random/src/mt19937/mt19937.c, src/legacy/legacy-distributions.c, _bounded_integers: masked
// rejection on 32-bit words), the four primitives involved.  tests/test_rng_host.py pins them against draw sequences
// recorded from NumPy 2.2.6 (tests/golden/rng_mt19937.npz) and against the live NumPy of whatever box runs the tests.
// log()/pow() come from the same libm NumPy links against, which is what makes the gamma variates bit-identical.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/transgo_hip.h"

namespace {

constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;

void twist(tg_mt19937* s) {
    uint32_t* mt = s->key;
    int kk;
    uint32_t y;
    for (kk = 0; kk < N - M; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < N - 1; kk++) {
        y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
        mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    s->pos = 0;
}

inline uint32_t next32(tg_mt19937* s) {
    if (s->pos == N) twist(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

inline double next_double(tg_mt19937* s) {
    int32_t a = (int32_t)(next32(s) >> 5), b = (int32_t)(next32(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

inline double std_exponential(tg_mt19937* s) { return -log(1.0 - next_double(s)); }

// legacy_standard_gamma for shape < 1 (the Dirichlet concentrations used are 0.03); == 1 is the exponential.
double std_gamma_small(tg_mt19937* s, double shape) {
    if (shape == 1.0) return std_exponential(s);
    if (shape == 0.0) return 0.0;
    for (;;) {
        double U = next_double(s);
        double V = std_exponential(s);
        if (U <= 1.0 - shape) {
            double X = pow(U, 1. / shape);
            if (X <= V) return X;
        } else {
            double Y = -log((1 - U) / shape);
            double X = pow(1.0 - shape + shape * Y, 1. / shape);
            if (X <= (V + Y)) return X;
        }
    }
}

}  // namespace

extern "C" {

void tg_host_mt_seed(tg_mt19937* s, uint32_t seed) {     // RandomState(seed) / np.random.seed(seed) for an integer
    for (int pos = 0; pos < N; pos++) {
        s->key[pos] = seed;
        seed = (1812433253u * (seed ^ (seed >> 30)) + pos + 1);
    }
    s->pos = N;
}

uint32_t tg_host_mt_next32(tg_mt19937* s) { return next32(s); }

double tg_host_mt_random_sample(tg_mt19937* s) { return next_double(s); }

// RandomState.choice(list_of_k) -> index (randint(0,k): masked rejection on 32-bit words; no draw when k == 1)
int32_t tg_host_mt_choice_index(tg_mt19937* s, int32_t k) {
    if (k <= 1) return 0;
    uint32_t rng = (uint32_t)(k - 1), mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = next32(s) & mask; } while (v > rng);
    return (int32_t)v;
}

// RandomState.dirichlet([alpha]*n) with alpha <= 1 -> out[n].  Returns 0, or -1 for unsupported alpha.
int tg_host_mt_dirichlet(tg_mt19937* s, double alpha, int32_t n, double* out) {
    if (!(alpha > 0.0) || alpha > 1.0) return -1;
    double acc = 0.0;
    for (int i = 0; i < n; i++) { out[i] = std_gamma_small(s, alpha); acc += out[i]; }
    double inv = 1.0 / acc;
    for (int i = 0; i < n; i++) out[i] = out[i] * inv;
    return 0;
}

}  // extern "C"
