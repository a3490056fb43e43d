// Semantics check for LDS-DMA (global_load_lds): lane l of a wave-instruction lands at lds_base + l*size,
// for the 4-byte (per-lane gather) and the 16-byte form.  Prints OK/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__global__ __launch_bounds__(256) void k(const float *src, const int *gather, float *out)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = tid >> 6;
    // 16-byte form: 2 slots of 256 threads x 16 B = 8 KB contiguous copy
    for (int sl = 0; sl < 2; ++sl) {
        const float *g = src + (sl * 256 + tid) * 4;
        float *l = lds + (sl * 256 + wave * 64) * 4;  // wave-uniform base
        __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 16, 0, 0);
    }
    // 4-byte form with a per-lane source address (gather), 3 slots
    for (int sl = 0; sl < 3; ++sl) {
        const float *g = src + gather[sl * 256 + tid];
        float *l = lds + 2048 + sl * 256 + wave * 64;
        __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)l, 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 2048 + 768; i += 256) out[i] = lds[i];
}

int main()
{
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i;
    std::vector<int> gi(768);
    for (int i = 0; i < 768; ++i) gi[i] = (i * 2654435761u) % 4096;
    float *src, *out; int *gat;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&out, 2816 * 4); (void)hipMalloc(&gat, 768 * 4);
    (void)hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(gat, gi.data(), 768 * 4, hipMemcpyHostToDevice);
    k<<<1, 256, 2816 * 4>>>(src, gat, out);
    std::vector<float> o(2816);
    (void)hipMemcpy(o.data(), out, 2816 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 2048; ++i) bad += o[i] != h[i];
    for (int i = 0; i < 768; ++i) bad += o[2048 + i] != h[gi[i]];
    printf("%s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
