// FETCH_SIZE calibration on gfx950 (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly half of the bytes of a wide
// coalesced streaming read"): every kernel below reads the SAME 1 GiB exactly once with a different access shape;
// run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and compare the counter (KiB) with 1,048,576 KiB.
//   k_reg16   : global_load_dwordx4, lane-contiguous (1 KiB per wave-instruction)
//   k_lds16   : global_load_lds 16 B per lane, lane-contiguous (the weight / output-layer row DMA)
//   k_lds4    : global_load_lds 4 B per lane, lane-contiguous (256 B per wave-instruction)
//   k_lds4_rows: global_load_lds 4 B per lane, lanes walk 35-float rows of a 2-D patch (row pitch 128 floats): the
//               activation-patch gather of conv_tap_mfma_kernel with 4-byte pieces
//   k_lds16_rows: 16 B per lane, 9 pieces per 36-float row (the 16-byte "patch4" pieces of the 8-wave kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
constexpr size_t kBytes = 1ull << 30;

__global__ __launch_bounds__(256) void k_reg16(const float4 *src, float *sink, size_t n16)
{
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_lds16(const float *src, float *sink, size_t n16)
{
    __shared__ float lds[8 * 256 * 4];
    const int wave = threadIdx.x >> 6;
    int slot = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull, slot = (slot + 1) & 7)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + i * 4), (lds_void *)(lds + slot * 1024 + wave * 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 123.456f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_lds4(const float *src, float *sink, size_t n4)
{
    __shared__ float lds[8 * 256];
    const int wave = threadIdx.x >> 6;
    int slot = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += gridDim.x * 256ull, slot = (slot + 1) & 7)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + i), (lds_void *)(lds + slot * 256 + wave * 64), 4, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 123.456f) sink[0] = 1.f;
}
// the buffer as rows of 128 floats; a "patch" = 35 (or 36) floats of each row; 128/35 -> every row is visited by 4 patches
// at x = 0, 32, 64, 93 so that all 128 floats are touched (overlaps = halo re-reads, as in a real tile grid)
__global__ __launch_bounds__(256) void k_lds4_rows(const float *src, float *sink, size_t rows)
{
    __shared__ float lds[8 * 256];
    const int wave = threadIdx.x >> 6;
    int slot = 0;
    const size_t total = rows * 4 * 35;   // elements requested
    for (size_t e = blockIdx.x * 256ull + threadIdx.x; e < total; e += gridDim.x * 256ull, slot = (slot + 1) & 7) {
        const size_t patch = e / 35, col = e % 35, row = patch / 4;
        const int x0 = (patch % 4) == 3 ? 93 : 32 * static_cast<int>(patch % 4);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + row * 128 + x0 + col), (lds_void *)(lds + slot * 256 + wave * 64), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 123.456f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void k_lds16_rows(const float *src, float *sink, size_t rows)
{
    __shared__ float lds[8 * 256 * 4];
    const int wave = threadIdx.x >> 6;
    int slot = 0;
    const size_t total = rows * 4 * 9;   // 16-byte pieces requested: 4 patches x 9 pieces (36 floats) per row
    for (size_t e = blockIdx.x * 256ull + threadIdx.x; e < total; e += gridDim.x * 256ull, slot = (slot + 1) & 7) {
        const size_t patch = e / 9, piece = e % 9, row = patch / 4;
        const int x0 = (patch % 4) == 3 ? 92 : 32 * static_cast<int>(patch % 4);
        __builtin_amdgcn_global_load_lds((glb_void *)(src + row * 128 + x0 + piece * 4), (lds_void *)(lds + slot * 1024 + wave * 256), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lds[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

int main()
{
    float *src, *sink;
    (void)hipMalloc(&src, kBytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(src, 0, kBytes);
    float *flush;   // 1 GiB written between kernels: evicts the Infinity Cache so that every kernel streams from HBM
    (void)hipMalloc(&flush, kBytes);
    const int grid = 256 * 8;
    const size_t rows = kBytes / 512;
#define RUN(K, ...) do { (void)hipMemset(flush, 1, kBytes); (void)hipDeviceSynchronize(); K<<<grid, 256>>>(__VA_ARGS__); (void)hipDeviceSynchronize(); } while (0)
    RUN(k_reg16, reinterpret_cast<const float4 *>(src), sink, kBytes / 16);
    RUN(k_lds16, src, sink, kBytes / 16);
    RUN(k_lds4, src, sink, kBytes / 4);
    RUN(k_lds4_rows, src, sink, rows);
    RUN(k_lds16_rows, src, sink, rows);
    printf("each kernel touched every byte of a 1 GiB buffer (1048576 KiB); *_rows request 4x35 (4x36) floats of every 128-float row = 1.094x (1.125x) the bytes\n");
    return 0;
}
