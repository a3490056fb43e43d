// Practical fp32 MFMA ceiling on this GPU: back-to-back v_mfma_f32_32x32x2f32 on independent accumulators,
// no memory traffic.  Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0, int noise)
{
    f32x16 acc[NACC];
    for (int m = 0; m < NACC; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    // `noise` != 0: operands change every step with full-mantissa pseudo-random values (realistic
    // toggle rate, so the sustained clock under power limits shows up); 0: constant operands.
    unsigned h = 0x9E3779B9u * (threadIdx.x + 1u + blockIdx.x * 256u), g = 0;
    for (int i = 0; i < iters; ++i) {
        if (noise == 4) {  // the same VALU work, not feeding the MFMA operands
            h = h * 1664525u + 1013904223u;
            g ^= 0x3F000000u | (h >> 9);
            g += 0x3F000000u | ((h * 2654435761u) >> 9);
        }
        if (noise == 1 || (noise == 2 && i == 0) || (noise == 3 && (i & 15) == 0)) {
            h = h * 1664525u + 1013904223u;
            a = __uint_as_float(0x3F000000u | (h >> 9));
            b = __uint_as_float(0x3F000000u | ((h * 2654435761u) >> 9));
        }
#pragma unroll
        for (int m = 0; m < NACC; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
    }
    float s = 0.f;
    for (int m = 0; m < NACC; ++m)
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(g);
}

template <int NACC>
void run(int blocks, int iters, int noise)
{
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f, noise);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f, noise);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        const double flops = 2.0 * 32 * 32 * 2 * double(NACC) * iters * 4.0 * blocks;
        printf("acc=%d blocks=%d iters=%d noise=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks, iters, noise, ms, flops / ms * 1e-9);
    }
    (void)hipFree(out);
}

int main()
{
    run<4>(512, 20000, 0);    // constant operands, 2 workgroups per CU
    run<4>(512, 200000, 0);   // ~50 ms launches: shows the clock settling under load
    run<4>(512, 200000, 1);   // pseudo-random operands, new every step
    run<4>(512, 200000, 2);   // pseudo-random per lane, constant in time
    run<4>(512, 200000, 3);   // pseudo-random, new every 16 steps
    run<4>(512, 200000, 4);   // constant operands + the same integer VALU work on unrelated registers
    run<1>(1024, 40000, 0);   // one dependent chain per wave, 4 workgroups per CU
    return 0;
}
