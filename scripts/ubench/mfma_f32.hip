// Microbenchmark: what does the fp32 MFMA stream of k_conv3x3 cost step by step? (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int MODE, int NACC>   // MODE 0: pure MFMA from registers; 1: + ds_read_b128 fragments as in the conv; 2: + barrier per 128 MFMAs
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[148 * 36 + 128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 148 * 36 + 128 * 36; i += 256) lds[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f32x4 a = *reinterpret_cast<f32x4*>(&lds[j * 36 + kq * 4]);
    f32x4 b0 = *reinterpret_cast<f32x4*>(&lds[(j + 16) * 36 + kq * 4]);
    f32x4 b1 = *reinterpret_cast<f32x4*>(&lds[(j + 32) * 36 + kq * 4]);
    for (int it = 0; it < iters; ++it) {
        // one "stage": 2 subs x 8 ct x (4 k-steps x 2 pt) = 128 MFMAs
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (MODE >= 1 && !(MODE == 4 && sub == 0)) {
                b0 = *reinterpret_cast<f32x4*>(&lds[(j + 16 + (it & 7)) * 36 + sub * 16 + kq * 4]);
                b1 = *reinterpret_cast<f32x4*>(&lds[(j + 32 + (it & 7)) * 36 + sub * 16 + kq * 4]);
            }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                if (MODE >= 1 && !(MODE == 4 && sub == 0 && ct == 0)) a = *reinterpret_cast<f32x4*>(&lds[148 * 36 + (ct * 16 + j) * 36 + sub * 16 + kq * 4]);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[(ct * 2) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc[(ct * 2) % NACC], 0, 0, 0);
                    acc[(ct * 2 + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc[(ct * 2 + 1) % NACC], 0, 0, 0);
                }
            }
        }
        if (MODE == 2) __syncthreads();
        if (MODE == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        if (MODE == 4) {   // fragments of the next stage fetched BEFORE a raw barrier (3-deep ring makes that legal)
            a = *reinterpret_cast<f32x4*>(&lds[148 * 36 + j * 36 + kq * 4]);
            b0 = *reinterpret_cast<f32x4*>(&lds[(j + 16 + ((it + 1) & 7)) * 36 + kq * 4]);
            b1 = *reinterpret_cast<f32x4*>(&lds[(j + 32 + ((it + 1) & 7)) * 36 + kq * 4]);
            __builtin_amdgcn_s_barrier();
        }
    }
    f32x4 r = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = r[0] + r[1] + r[2] + r[3];
}

template <int MODE, int NACC> void run(const char* name, int wgs_per_cu, float* d_out) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), 0, 0, d_out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(grid), dim3(256), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 128.0 * 2048.0;
    printf("%-28s wgs/cu %d: %.3f ms  %.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 8 * 256);
    run<0, 16>("pure mfma, 16 acc", 1, d); run<0, 16>("pure mfma, 16 acc", 2, d);
    run<0, 2>("pure mfma, 2 acc", 1, d);  run<0, 2>("pure mfma, 2 acc", 2, d);
    run<1, 16>("+lds frags", 1, d); run<1, 16>("+lds frags", 2, d);
    run<2, 16>("+lds frags +barrier", 1, d); run<2, 16>("+lds frags +barrier", 2, d);
    run<2, 16>("+lds frags +barrier", 3, d);
    run<3, 16>("raw s_barrier", 1, d); run<3, 16>("raw s_barrier", 2, d);
    run<4, 16>("raw barrier, frags prefetched", 1, d); run<4, 16>("raw barrier, frags prefetched", 2, d);
    return 0;
}
