// Entropy-parameter kernels: fused quantise + table-index evaluation, coalesced over HBM.
// These are pure streaming kernels (a few bytes in, 8 bytes out per latent): the only rules
// that matter are full-width coalesced accesses and enough workgroups to cover 256 CUs.
//
// Reference semantics:
//   GaussianConditional.build_indexes / quantize  (CompressAI 1.2.3, quoted at
//       modules/prior_model/prior_coder/pgm_coder.py:814-818; call site compressai_coder.py:377-393)
//   EntropyBottleneck.compress/decompress         (call site compressai_coder.py:230-245)
//   GaussianPGMPriorCoderImpl._select_best_indexes pgm_coder.py:802-821
//   TopoGroupPGMPriorCoder group step              pgm_coder.py:921-941, 962-978
#include "common.h"

using namespace basic;

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > 256 * 8) g = 256 * 8;  // grid-stride the rest
    if (g < 1) g = 1;
    return static_cast<int>(g);
}

// torch.round / rintf: round half to even (v_rndne_f32).
__device__ __forceinline__ float round_even(float v) { return rintf(v); }

__global__ void gc_quantize_index_kernel(const float *__restrict__ y, const float *__restrict__ scales, int64_t n,
                                         const float *__restrict__ table, int table_len, float bound,
                                         int32_t *__restrict__ symbols, int32_t *__restrict__ indexes,
                                         float *__restrict__ yhat)
{
    extern __shared__ float s_table[];
    __shared__ int s_unsorted;
    if (threadIdx.x == 0) s_unsorted = 0;
    for (int i = threadIdx.x; i < table_len; i += blockDim.x) s_table[i] = table[i];
    __syncthreads();
    for (int i = threadIdx.x; i + 1 < table_len; i += blockDim.x)
        if (!(s_table[i] <= s_table[i + 1])) s_unsorted = 1;
    __syncthreads();
    // build_indexes (compressai GaussianConditional): idx = (len - 1) - #{ j < len - 1 : s <= table[j] }.  On a non-decreasing
    // table (the scale table is: checked above) that count is the number of entries from the first one >= s on, so idx is
    // that entry's position, found in log2(len) probes instead of len - 1 compares per element (the kernel was VALU-bound at
    // ten times its HBM time).  NaN and unsorted tables take the counting loop.
    const bool sorted = s_unsorted == 0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float s = fmaxf(scales[i], bound);  // LowerBound
        int idx;
        if (sorted && s == s) {
            int lo = 0, hi = table_len - 1;   // first j in [0, len - 1) with table[j] >= s, len - 1 if none
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_table[mid] < s) lo = mid + 1; else hi = mid;
            }
            idx = lo;
        } else {
            idx = table_len - 1;
            for (int j = 0; j < table_len - 1; ++j) idx -= (s <= s_table[j]) ? 1 : 0;
        }
        const float q = round_even(y[i]);
        symbols[i] = static_cast<int32_t>(q);
        indexes[i] = idx;
        if (yhat) yhat[i] = q;
    }
}

__global__ void eb_quantize_index_kernel(const float *__restrict__ z, const float *__restrict__ medians, int64_t total,
                                         int channels, int hw, int32_t *__restrict__ symbols,
                                         int32_t *__restrict__ indexes, float *__restrict__ zhat)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>((i / hw) % channels);
        const float m = medians[c];
        const float q = round_even(z[i] - m);
        symbols[i] = static_cast<int32_t>(q);
        indexes[i] = c;
        if (zhat) zhat[i] = q + m;
    }
}

__global__ void eb_dequantize_kernel(const int32_t *__restrict__ symbols, const float *__restrict__ medians,
                                     int64_t total, int channels, int hw, float *__restrict__ zhat)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>((i / hw) % channels);
        zhat[i] = static_cast<float>(symbols[i]) + medians[c];
    }
}

__global__ void i32_to_f32_kernel(const int32_t *__restrict__ in, int64_t n, float *__restrict__ out)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        out[i] = static_cast<float>(in[i]);
}

// argmin_j |s - table[j]|, first minimum (torch.argmin on CPU returns the first occurrence).
__device__ __forceinline__ int nearest_scale(float s, const float *tab, int n)
{
    int best = 0;
    float bd = fabsf(s - tab[0]);
    for (int j = 1; j < n; ++j) {
        const float d = fabsf(s - tab[j]);
        if (d < bd) { bd = d; best = j; }
    }
    return best;
}

// MODE 0: encode (symbols + indexes + write-back), 1: indexes only, 2: scatter decoded symbols.
template <int MODE>
__global__ void pgm_gauss_group_kernel(const float *__restrict__ y, const float *__restrict__ params, int batch,
                                       int channels, int hw, const int32_t *__restrict__ elems, int64_t n_elems,
                                       const float *__restrict__ table, int table_len, int32_t *symbols,
                                       int32_t *indexes, int64_t per_image, int64_t out_base, float *ybuf)
{
    extern __shared__ float s_table[];
    if (MODE != 2) {
        for (int i = threadIdx.x; i < table_len; i += blockDim.x) s_table[i] = table[i];
        __syncthreads();
    }
    const int64_t total = static_cast<int64_t>(batch) * n_elems;
    const int64_t chw = static_cast<int64_t>(channels) * hw;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; t < total;
         t += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t b = t / n_elems, k = t - b * n_elems;
        const int32_t e = elems[k];  // c*hw + p
        const int c = e / hw, p = e - c * hw;
        // split_interleave: channel 2c = mean, 2c+1 = scale (pgm_coder.py:743-752)
        const float *pb = params + (b * 2 * chw) + static_cast<int64_t>(2 * c) * hw + p;
        const float mu = pb[0];
        const int64_t o = b * per_image + out_base + k;
        if (MODE == 0 || MODE == 1) indexes[o] = nearest_scale(pb[hw], s_table, table_len);
        if (MODE == 0) {
            const float q = round_even(y[b * chw + e] - mu);
            symbols[o] = static_cast<int32_t>(q);
            ybuf[b * chw + e] = q + mu;
        }
        if (MODE == 2) ybuf[b * chw + e] = static_cast<float>(symbols[o]) + mu;
    }
}

constexpr int kMseBlock = 1024;
__global__ __launch_bounds__(kMseBlock) void mse_per_image_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t elems,
                                                                  float *__restrict__ mse)
{
    // one workgroup per image; per thread four independent 16-byte streams (a batch-1 image is 1.2 M elements on ONE compute
    // unit: scalar loads took 1.9 ms per Kodak-shaped image), pairwise tree in LDS; the order is fixed by the shape alone
    __shared__ float red[kMseBlock];
    const int img = blockIdx.x;
    const float *pa = a + static_cast<int64_t>(img) * elems, *pb = b + static_cast<int64_t>(img) * elems;
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    int64_t done = 0;
    if (((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15u) == 0) {
        const f4 *qa = reinterpret_cast<const f4 *>(pa), *qb = reinterpret_cast<const f4 *>(pb);
        const int64_t n4 = elems >> 2;
        f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
        int64_t i = threadIdx.x;
        for (; i + 3 * kMseBlock < n4; i += 4 * kMseBlock) {
            const f4 d0 = qa[i] - qb[i], d1 = qa[i + kMseBlock] - qb[i + kMseBlock], d2 = qa[i + 2 * kMseBlock] - qb[i + 2 * kMseBlock],
                     d3 = qa[i + 3 * kMseBlock] - qb[i + 3 * kMseBlock];
            s0 += d0 * d0; s1 += d1 * d1; s2 += d2 * d2; s3 += d3 * d3;
        }
        for (; i < n4; i += kMseBlock) { const f4 d = qa[i] - qb[i]; s0 += d * d; }
        const f4 t = (s0 + s1) + (s2 + s3);
        acc = (t[0] + t[1]) + (t[2] + t[3]);
        done = n4 << 2;
    }
    for (int64_t i = done + threadIdx.x; i < elems; i += kMseBlock) {
        const float d = pa[i] - pb[i];
        acc += d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kMseBlock / 2; s > 0; s >>= 1) {
        if (static_cast<int>(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) mse[img] = red[0] / static_cast<float>(elems);
}

// ---- rate estimate ("prior_entropy", nats per image): -sum log(max(P(q), bound)) -------------------------------
// MODE 0: CompressAI GaussianConditional._likelihood (upstream; call site compressai_coder.py:352-375):
//         zero mean, v = |q|, P = Phi((.5 - v)/s) - Phi((-.5 - v)/s), Phi(t) = .5 erfc(-t/sqrt2), s = max(scale, bound_s)
// MODE 1: PGM coder (pgm_coder.py:374-389,757-778): Normal(mu, max(scale, bound_s)).cdf(q + .5) - cdf(q - .5) with
//         torch's cdf = .5 (1 + erf((x - mu) / (s sqrt2))), params "split_interleave" in a [B][2C][HW] tensor.
template <int MODE>
__global__ void gauss_nll_per_image_kernel(const float *__restrict__ q, const float *__restrict__ scales_or_params,
                                           int64_t elems, int hw, float bound_s, float bound_p, float *__restrict__ nll)
{
    __shared__ float red[kBlock];
    const int img = blockIdx.x;
    const float *pq = q + static_cast<int64_t>(img) * elems;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < elems; i += blockDim.x) {
        float p;
        if (MODE == 0) {
            const float s = fmaxf(scales_or_params[static_cast<int64_t>(img) * elems + i], bound_s);
            const float v = fabsf(pq[i]);
            const float c = -0.70710678118654752440f;
            p = 0.5f * erfcf(c * ((0.5f - v) / s)) - 0.5f * erfcf(c * ((-0.5f - v) / s));
        } else {
            const int64_t c = i / hw, pos = i - c * hw;
            const float *pp = scales_or_params + static_cast<int64_t>(img) * 2 * elems + (2 * c) * hw + pos;
            const float mu = pp[0], s = fmaxf(pp[hw], bound_s);
            const float r = 0.70710678118654752440f / s;  // 1 / (s sqrt2) as torch: (x - mu) * s.reciprocal() / sqrt2
            if (MODE == 2) {   // likelihood of the ROUNDED RESIDUAL under the zero-mean density (pgm_coder.py:376-387)
                const float v = rintf(pq[i] - mu);
                p = 0.5f * (1.f + erff((v + 0.5f) * r)) - 0.5f * (1.f + erff((v - 0.5f) * r));
            } else {
                p = 0.5f * (1.f + erff((pq[i] + 0.5f - mu) * r)) - 0.5f * (1.f + erff((pq[i] - 0.5f - mu) * r));
            }
        }
        acc += -logf(fmaxf(p, bound_p));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) nll[img] = red[0];
}

// EntropyBottleneck likelihood (upstream _logits_cumulative; call site compressai_coder.py:203-228): per channel a
// 1-3-3-3-3-1 network with pre-activated parameters coef[c][58] = softplus(M0[3]) b0[3] tanh(f0)[3] | softplus(M1[9]) b1[3]
// tanh(f1)[3] | M2.. | M3.. | softplus(M4[3]) b4[1].
__device__ __forceinline__ float eb_logits(const float *k, float v)
{
    float h[3], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { h[i] = k[i] * v + k[3 + i]; h[i] += k[6 + i] * tanhf(h[i]); }
    k += 9;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { t[i] = k[3 * i] * h[0] + k[3 * i + 1] * h[1] + k[3 * i + 2] * h[2] + k[9 + i]; t[i] += k[12 + i] * tanhf(t[i]); }
#pragma unroll
        for (int i = 0; i < 3; ++i) h[i] = t[i];
        k += 15;
    }
    return k[0] * h[0] + k[1] * h[1] + k[2] * h[2] + k[3];
}

__global__ void eb_nll_per_image_kernel(const float *__restrict__ zq, const float *__restrict__ coef, int channels, int hw,
                                        float bound_p, float *__restrict__ nll)
{
    __shared__ float red[kBlock];
    const int img = blockIdx.x;
    const int64_t elems = static_cast<int64_t>(channels) * hw;
    const float *pz = zq + static_cast<int64_t>(img) * elems;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < elems; i += blockDim.x) {
        const float *k = coef + (i / hw) * 58;
        const float lower = eb_logits(k, pz[i] - 0.5f), upper = eb_logits(k, pz[i] + 0.5f);
        const float sg = (lower + upper) > 0.f ? -1.f : ((lower + upper) < 0.f ? 1.f : 0.f);
        const float p = fabsf(1.f / (1.f + expf(-sg * upper)) - 1.f / (1.f + expf(-sg * lower)));
        acc += -logf(fmaxf(p, bound_p));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if (static_cast<int>(threadIdx.x) < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) nll[img] = red[0];
}

}  // namespace

extern "C" int basic_gc_quantize_index_dev(const float *d_y, const float *d_scales, int64_t n, const float *d_table,
                                           int table_len, float scale_bound, int32_t *d_symbols, int32_t *d_indexes,
                                           float *d_yhat, void *hip_stream)
{
    BASIC_REQUIRE(d_y && d_scales && d_table && d_symbols && d_indexes && n >= 0 && table_len >= 1 && table_len <= 4096,
                  "gc_quantize_index: bad argument");
    if (n == 0) return BASIC_OK;
    hipLaunchKernelGGL(gc_quantize_index_kernel, dim3(grid_for(n)), dim3(kBlock), table_len * sizeof(float),
                       as_stream(hip_stream), d_y, d_scales, n, d_table, table_len, scale_bound, d_symbols, d_indexes, d_yhat);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_eb_quantize_index_dev(const float *d_z, const float *d_medians, int batch, int channels, int hw,
                                           int32_t *d_symbols, int32_t *d_indexes, float *d_zhat, void *hip_stream)
{
    BASIC_REQUIRE(d_z && d_medians && d_symbols && d_indexes && batch >= 1 && channels >= 1 && hw >= 1,
                  "eb_quantize_index: bad argument");
    const int64_t total = static_cast<int64_t>(batch) * channels * hw;
    hipLaunchKernelGGL(eb_quantize_index_kernel, dim3(grid_for(total)), dim3(kBlock), 0, as_stream(hip_stream), d_z,
                       d_medians, total, channels, hw, d_symbols, d_indexes, d_zhat);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_eb_dequantize_dev(const int32_t *d_symbols, const float *d_medians, int batch, int channels,
                                       int hw, float *d_zhat, void *hip_stream)
{
    BASIC_REQUIRE(d_symbols && d_medians && d_zhat && batch >= 1 && channels >= 1 && hw >= 1, "eb_dequantize: bad argument");
    const int64_t total = static_cast<int64_t>(batch) * channels * hw;
    hipLaunchKernelGGL(eb_dequantize_kernel, dim3(grid_for(total)), dim3(kBlock), 0, as_stream(hip_stream), d_symbols,
                       d_medians, total, channels, hw, d_zhat);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_i32_to_f32_dev(const int32_t *d_in, int64_t n, float *d_out, void *hip_stream)
{
    BASIC_REQUIRE(d_in && d_out && n >= 0, "i32_to_f32: bad argument");
    if (n == 0) return BASIC_OK;
    hipLaunchKernelGGL(i32_to_f32_kernel, dim3(grid_for(n)), dim3(kBlock), 0, as_stream(hip_stream), d_in, n, d_out);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_pgm_gauss_encode_group_dev(const float *d_y, const float *d_params, int batch, int channels,
                                                int hw, const int32_t *d_elems, int64_t n_elems, const float *d_table,
                                                int table_len, int32_t *d_symbols, int32_t *d_indexes,
                                                int64_t per_image, int64_t out_base, float *d_ybuf, void *hip_stream)
{
    BASIC_REQUIRE(d_y && d_params && d_elems && d_table && d_symbols && d_indexes && d_ybuf && batch >= 1 &&
                      channels >= 1 && hw >= 1 && n_elems >= 0 && table_len >= 1 && table_len <= 4096,
                  "pgm_gauss_encode_group: bad argument");
    if (n_elems == 0) return BASIC_OK;
    hipLaunchKernelGGL(pgm_gauss_group_kernel<0>, dim3(grid_for(batch * n_elems)), dim3(kBlock),
                       table_len * sizeof(float), as_stream(hip_stream), d_y, d_params, batch, channels, hw, d_elems,
                       n_elems, d_table, table_len, d_symbols, d_indexes, per_image, out_base, d_ybuf);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_pgm_gauss_index_group_dev(const float *d_params, int batch, int channels, int hw,
                                               const int32_t *d_elems, int64_t n_elems, const float *d_table,
                                               int table_len, int32_t *d_indexes, int64_t per_image, int64_t out_base,
                                               void *hip_stream)
{
    BASIC_REQUIRE(d_params && d_elems && d_table && d_indexes && batch >= 1 && channels >= 1 && hw >= 1 &&
                      n_elems >= 0 && table_len >= 1 && table_len <= 4096,
                  "pgm_gauss_index_group: bad argument");
    if (n_elems == 0) return BASIC_OK;
    hipLaunchKernelGGL(pgm_gauss_group_kernel<1>, dim3(grid_for(batch * n_elems)), dim3(kBlock),
                       table_len * sizeof(float), as_stream(hip_stream), nullptr, d_params, batch, channels, hw,
                       d_elems, n_elems, d_table, table_len, nullptr, d_indexes, per_image, out_base, nullptr);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_pgm_gauss_scatter_group_dev(const int32_t *d_symbols, const float *d_params, int batch,
                                                 int channels, int hw, const int32_t *d_elems, int64_t n_elems,
                                                 int64_t per_image, int64_t in_base, float *d_ybuf, void *hip_stream)
{
    BASIC_REQUIRE(d_symbols && d_params && d_elems && d_ybuf && batch >= 1 && channels >= 1 && hw >= 1 && n_elems >= 0,
                  "pgm_gauss_scatter_group: bad argument");
    if (n_elems == 0) return BASIC_OK;
    hipLaunchKernelGGL(pgm_gauss_group_kernel<2>, dim3(grid_for(batch * n_elems)), dim3(kBlock), 0,
                       as_stream(hip_stream), nullptr, d_params, batch, channels, hw, d_elems, n_elems, nullptr, 0,
                       const_cast<int32_t *>(d_symbols), nullptr, per_image, in_base, d_ybuf);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_mse_per_image_dev(const float *d_a, const float *d_b, int batch, int64_t elems_per_image,
                                       float *d_mse, void *hip_stream)
{
    BASIC_REQUIRE(d_a && d_b && d_mse && batch >= 1 && elems_per_image >= 1, "mse_per_image: bad argument");
    hipLaunchKernelGGL(mse_per_image_kernel, dim3(batch), dim3(kMseBlock), 0, as_stream(hip_stream), d_a, d_b,
                       elems_per_image, d_mse);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_gauss_nll_per_image_dev(const float *d_q, const float *d_scales_or_params, int batch, int channels, int hw,
                                             int interleaved_mean_scale, float scale_bound, float likelihood_bound, float *d_nll,
                                             void *hip_stream)
{
    BASIC_REQUIRE(d_q && d_scales_or_params && d_nll && batch >= 1 && channels >= 1 && hw >= 1, "gauss_nll_per_image: bad argument");
    const int64_t elems = static_cast<int64_t>(channels) * hw;
    if (interleaved_mean_scale == 2)
        hipLaunchKernelGGL(gauss_nll_per_image_kernel<2>, dim3(batch), dim3(kBlock), 0, as_stream(hip_stream), d_q, d_scales_or_params,
                           elems, hw, scale_bound, likelihood_bound, d_nll);
    else if (interleaved_mean_scale)
        hipLaunchKernelGGL(gauss_nll_per_image_kernel<1>, dim3(batch), dim3(kBlock), 0, as_stream(hip_stream), d_q, d_scales_or_params,
                           elems, hw, scale_bound, likelihood_bound, d_nll);
    else
        hipLaunchKernelGGL(gauss_nll_per_image_kernel<0>, dim3(batch), dim3(kBlock), 0, as_stream(hip_stream), d_q, d_scales_or_params,
                           elems, hw, scale_bound, likelihood_bound, d_nll);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}

extern "C" int basic_eb_nll_per_image_dev(const float *d_zq, const float *d_coef, int batch, int channels, int hw,
                                          float likelihood_bound, float *d_nll, void *hip_stream)
{
    BASIC_REQUIRE(d_zq && d_coef && d_nll && batch >= 1 && channels >= 1 && hw >= 1, "eb_nll_per_image: bad argument");
    hipLaunchKernelGGL(eb_nll_per_image_kernel, dim3(batch), dim3(kBlock), 0, as_stream(hip_stream), d_zq, d_coef, channels, hw,
                       likelihood_bound, d_nll);
    BASIC_HIP_TRY(hipGetLastError());
    return BASIC_OK;
}
