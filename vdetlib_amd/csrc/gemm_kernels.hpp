// gemm_kernels.hpp -- the path's one dense contraction, svm_scores (vdet/image_det.py:109-114):
//     scores[n, m] = features[n, k] . W[k, m] + B[m]        ([n, 1024] x [1024, 200] in the reference's models)
// as a hand-written MFMA kernel for gfx950.  The reference computes in numpy's result dtype:
// float64 for its .mat SVM models (W is f64), float32 when both operands are f32 -- so both forms exist:
//   v_mfma_f64_16x16x4_f64   one f64 per lane for A and B, 4 f64 accumulators per lane
//                            A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15];  D: col = l & 15, row = (l >> 4) + 4 * reg
//   v_mfma_f32_16x16x4_f32   same A / B maps;                                       D: col = l & 15, row = 4 * (l >> 4) + reg
// (maps: /opt/skills/guides/cdna_hip_programming.md section 3; the f64 D map differs from the f32 one.)
// One wave owns a 16 x 64 strip of the output (4 accumulator tiles share every A fragment); K is walked 4 at a
// time; out-of-range rows / columns / k are zero-filled on load and masked on store, so any n, k, m works.
// Accumulation order per output element: k ascending, one fused multiply-add per product (an f64 / f32 fma chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdet {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct MfmaTile;
template <> struct MfmaTile<double> {
    typedef f64x4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MfmaTile<float> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// grid = (ceil(m / 64), ceil(n / 64)); block = 256 = 4 waves, wave w owns rows [16 * (4 * by + w), +16)
template <typename T>
__global__ __launch_bounds__(256) void svm_scores_kernel(const T *__restrict__ A, const T *__restrict__ W, const T *__restrict__ bias,
                                                         int64_t n, int64_t k, int64_t m, T *__restrict__ out)
{
    typedef MfmaTile<T> MT;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.y * 4 + w) * 16;
    const int64_t col0 = (int64_t)blockIdx.x * 64;
    if (row0 >= n) return;
    const int li = lane & 15, lk = lane >> 4;
    typename MT::acc_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = (T)0;
    const int64_t ar = row0 + li;
    const T *arow = A + (ar < n ? ar : 0) * k;
    for (int64_t k0 = 0; k0 < k; k0 += 4) {
        const int64_t kk = k0 + lk;
        const bool kin = kk < k;
        const T a = (ar < n && kin) ? arow[kk] : (T)0;
        T b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t c = col0 + 16 * t + li;
            b[t] = (kin && c < m) ? W[kk * m + c] : (T)0;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MT::mma(a, b[t], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t c = col0 + 16 * t + li;
        if (c >= m) continue;
        const T bv = bias ? bias[c] : (T)0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t rr = row0 + MT::row(lane, r);
            if (rr < n) out[rr * m + c] = acc[t][r] + bv;
        }
    }
}

}  // namespace vdet
