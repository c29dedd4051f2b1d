// AddressSanitizer + UBSan driver for the host MT19937 helpers of the C ABI (transgo_amd/csrc/rng_host.cpp), built and run by
// tests/test_sanitizers.py on the CPU (GPU sanitizers are not available on the pool).
#include <cstdio>
#include <vector>

#include "../../include/transgo_hip.h"

int main() {
    tg_mt19937 st;
    double acc = 0;
    for (unsigned seed = 0; seed < 40; ++seed) {
        tg_host_mt_seed(&st, seed * 2654435761u + 1);
        for (int i = 0; i < 700; ++i) acc += tg_host_mt_next32(&st) * 1e-10;        // crosses the 624-word refill
        for (int i = 0; i < 50; ++i) acc += tg_host_mt_random_sample(&st);
        for (int n = 1; n < 40; ++n) acc += tg_host_mt_choice_index(&st, n);
        for (int n : {1, 2, 5, 81, 82, 361, 362}) {
            std::vector<double> out(n);
            tg_host_mt_dirichlet(&st, 0.03, n, out.data());
            for (double v : out) acc += v;
        }
    }
    std::printf("san_rng_host: checksum %.6f, no sanitizer report\n", acc);
    return 0;
}
