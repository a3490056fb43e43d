// Semantics check for the BUFFER form of LDS-DMA (buffer_load_dword ... offen lds): lane l lands at lds_base + 4 l, the
// address is base + voffset (per lane) + soffset (SGPR), and a lane whose voffset is out of range (0x80000000 with
// num_records < 2 GB) writes ZERO to its LDS slot -- the masked-convolution staging relies on exactly that for masked /
// padded elements (no zero page, no per-lane select, the channel offset on the scalar unit).  Prints OK/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

__global__ __launch_bounds__(256) void k(const float *src, int n, const int *voff, int soff, float *out)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 1024; i += 256) lds[i] = -7.f;   // stale data the DMA must overwrite (zeros included)
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n * 4, 0x00020000);
    for (int sl = 0; sl < 4; ++sl)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(lds + sl * 256 + wave * 64), 4, voff[sl * 256 + tid], soff * sl, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = lds[i];
}

int main()
{
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
    std::vector<int> vo(1024);
    for (int i = 0; i < 1024; ++i) vo[i] = (i % 5 == 3) ? (int)0x80000000u : (int)(((i * 2654435761u) % 8192) * 4);
    const int soff = 4 * 9000;
    float *src, *out; int *dv;
    (void)hipMalloc(&src, n * 4); (void)hipMalloc(&out, 1024 * 4); (void)hipMalloc(&dv, 1024 * 4);
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dv, vo.data(), 1024 * 4, hipMemcpyHostToDevice);
    k<<<1, 256, 4096>>>(src, n, dv, soff, out);
    std::vector<float> o(1024);
    (void)hipMemcpy(o.data(), out, 1024 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) {
        const float want = (i % 5 == 3) ? 0.f : h[vo[i] / 4 + (soff / 4) * (i / 256)];
        bad += o[i] != want;
    }
    printf("buffer lds-dma: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
