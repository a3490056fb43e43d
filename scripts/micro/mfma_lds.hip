// How much of the fp32 MFMA ceiling survives when every 4 MFMAs are fed by LDS fragment reads?
//   mode 0: 1 ds_read_b32 (B) + 1 ds_read_b128 (A, 4 M-tiles) per step, immediate offsets only
//   mode 1: the same data through 1 ds_read_b32 + 2 ds_read2_b32 and per-step address arithmetic
//           (what a runtime tap table costs)
// Build: hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, const int *tab, int rnd)
{
    __shared__ float lds[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) {
        unsigned h = (i + 1u + blockIdx.x * 12288u) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // rnd: full-mantissa pseudo-random values in [-1, 1) (realistic toggle rate); else a smooth ramp
        lds[i] = rnd ? (__uint_as_float(0x3F800000u | (h >> 9)) - 1.5f) * 2.f : 1.0f / (1 + i);
    }
    __shared__ int tl[64];
    if (threadIdx.x < 64) tl[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m)
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float *pa = lds + lane * 4;
    const float *pb = lds + 8192 + lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f32x4 a[2]; float b[2];
            a[0] = *reinterpret_cast<const f32x4 *>(pa); b[0] = pb[0];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int cur = s & 1, nxt = cur ^ 1;
                a[nxt] = *reinterpret_cast<const f32x4 *>(pa + ((s + 1) & 15) * 256);
                b[nxt] = pb[((s + 1) & 15) * 67];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m], b[cur], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float a[2][4], b[2];
            int off = tl[0];
            for (int m = 0; m < 4; ++m) a[0][m] = lds[lane + m * 32];
            b[0] = pb[off];
            for (int t = 0; t < 8; ++t) {
                const int offn = tl[(t + 1) & 7];
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    const int cur = cp & 1, nxt = cur ^ 1;
                    const float *wn = lds + (((cp ? t + 1 : t) & 7) * 4 + (cp ? 0 : 2)) * 128 + lane;
                    b[nxt] = pb[(cp ? offn : off) + (cp ? 0 : 2) * 300];
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[nxt][m] = wn[m * 32];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][m], b[cur], acc[m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                off = offn;
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < 4; ++m)
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks, int iters, int rnd)
{
    float *out; int *tab;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&tab, 256);
    int h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i * 37) % 200;
    (void)hipMemcpy(tab, h, 256, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, tab, rnd);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 32 * 32 * 2 * 4.0 * 16 * iters * 4.0 * blocks;
        if (rep == 3) printf("mode=%d rnd=%d blocks=%d  %.3f ms  %.1f TFLOP/s\n", MODE, rnd, blocks, ms, flops / ms * 1e-9);
    }
    (void)hipFree(out); (void)hipFree(tab);
}

int main()
{
    run<0>(512, 20000, 0);
    run<0>(512, 20000, 1);
    run<1>(512, 20000, 0);
    run<1>(512, 20000, 1);
    return 0;
}
