// What does v_mfma_f32_32x32x2_f32 compute, bit for bit?  D = C + a0*b0 + a1*b1 -- as which sequence of roundings?
// Candidates checked on random full-mantissa operands (incl. cancelling cases):
//   A  fma(a1, b1, fma(a0, b0, c))        (k = 0 first)
//   B  fma(a0, b0, fma(a1, b1, c))        (k = 1 first)
//   C  c + (a0*b0 + a1*b1) with rounded products
//   D  fma(a1, b1, a0*b0) + c
// and a chain of N MFMAs against the equivalent chain of 2N FMAs.  The answer decides whether a VALU dot product (the
// persistent scan-line kernel) can reproduce the masked convolution's MFMA sums exactly.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_arith.hip -o mfma_arith ; run: ./mfma_arith
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// one wave: A[32 rows][2], B[2][32 cols], C[32][32] -> D.  Lane l: row/col = l & 31, k = l >> 5.
__global__ void mfma_once(const float *A, const float *B, const float *C, float *D, int chain)
{
    const int lane = threadIdx.x, col = lane & 31, kh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + col];
    for (int s = 0; s < chain; ++s) {
        const float a = A[(s * 32 + col) * 2 + kh], b = B[(s * 2 + kh) * 32 + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) D[(8 * (r >> 2) + 4 * kh + (r & 3)) * 32 + col] = acc[r];
}

static uint32_t rng_state = 12345u;
static float rnd(int mode)
{
    rng_state = rng_state * 1664525u + 1013904223u;
    const uint32_t h = rng_state;
    if (mode == 0) return (float)((int32_t)h) / 2147483648.0f;                     // uniform (-1, 1), full mantissa
    uint32_t bits = (h & 0x807FFFFFu) | ((uint32_t)(120 + (h >> 23) % 14) << 23);  // wide exponent range
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

int main()
{
    const int chain_lens[] = {1, 1, 16, 64};
    const int modes[] = {0, 1, 0, 1};
    for (int trial = 0; trial < 4; ++trial) {
        const int chain = chain_lens[trial], mode = modes[trial];
        std::vector<float> A(chain * 64), B(chain * 64), C(1024), D(1024);
        long cnt[5] = {0, 0, 0, 0, 0}, total = 0;
        for (int rep = 0; rep < 64; ++rep) {
            for (auto &v : A) v = rnd(mode);
            for (auto &v : B) v = rnd(mode);
            for (auto &v : C) v = chain == 1 ? rnd(mode) : 0.f;
            if (rep % 4 == 3 && chain == 1)   // cancellation: c ~ -(a0 b0)
                for (int r = 0; r < 32; ++r)
                    for (int c = 0; c < 32; ++c) C[r * 32 + c] = -(A[r * 2] * B[c]);
            float *dA, *dB, *dC, *dD;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_once, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, chain);
            hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
            hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
            for (int r = 0; r < 32; ++r)
                for (int c = 0; c < 32; ++c) {
                    float vA = C[r * 32 + c], vB = vA, vC = vA, vD = vA;
                    for (int s = 0; s < chain; ++s) {
                        const float a0 = A[(s * 32 + r) * 2], a1 = A[(s * 32 + r) * 2 + 1];
                        const float b0 = B[(s * 2) * 32 + c], b1 = B[(s * 2 + 1) * 32 + c];
                        vA = std::fmaf(a1, b1, std::fmaf(a0, b0, vA));
                        vB = std::fmaf(a0, b0, std::fmaf(a1, b1, vB));
                        volatile float p0 = a0 * b0, p1 = a1 * b1;
                        volatile float ps = p0 + p1;
                        vC = vC + ps;
                        volatile float q = std::fmaf(a1, b1, p0);
                        vD = q + vD;
                    }
                    const float got = D[r * 32 + c];
                    ++total;
                    cnt[0] += std::memcmp(&got, &vA, 4) == 0;
                    cnt[1] += std::memcmp(&got, &vB, 4) == 0;
                    cnt[2] += std::memcmp(&got, &vC, 4) == 0;
                    cnt[3] += std::memcmp(&got, &vD, 4) == 0;
                }
        }
        std::printf("chain %3d mode %d: of %ld outputs equal to  A fma(k1, fma(k0, c)) %ld | B fma(k0, fma(k1, c)) %ld | C c + (p0 + p1) %ld | D fma(k1, p0) + c %ld\n",
                    chain, mode, total, cnt[0], cnt[1], cnt[2], cnt[3]);
    }
    return 0;
}
