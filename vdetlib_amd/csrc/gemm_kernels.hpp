// gemm_kernels.hpp -- the path's one dense contraction, svm_scores (vdet/image_det.py:109-114):
//     scores[n, m] = features[n, k] . W[k, m] + B[m]        ([n, 1024] x [1024, 200] in the reference's models)
// as a hand-written MFMA kernel for gfx950.  The reference computes in numpy's result dtype:
// float64 for its .mat SVM models (W is f64), float32 when both operands are f32 -- so both forms exist:
//   v_mfma_f64_16x16x4_f64   one f64 per lane for A and B, 4 f64 accumulators per lane
//                            A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15];  D: col = l & 15, row = (l >> 4) + 4 * reg
//   v_mfma_f32_16x16x4_f32   same A / B maps;                                       D: col = l & 15, row = 4 * (l >> 4) + reg
// (maps: /opt/skills/guides/cdna_hip_programming.md section 3; the f64 D map differs from the f32 one.)
// A 256-thread block owns a 64 x 64 tile of the output, wave w its rows [16 w, 16 w + 16) (4 accumulator tiles share every
// A fragment).  K is walked in chunks of 16: the 64 x 16 slice of the features and the 16 x 64 slice of W are staged in
// LDS with coalesced, 4-element-per-thread global loads (round 3; round 2 fed every MFMA from scalar global loads), the
// next chunk's loads are in flight while the current one is multiplied (register prefetch, two LDS buffers, one barrier
// per chunk), and the A tile is padded to 17 columns so that the 16 rows of a fragment fall into different banks.
// Out-of-range rows / columns / k are zero-filled on load and masked on store, so any n, k, m works.
// Accumulation order per output element: k ascending, 4 products per MFMA (the instruction's own order), as before.
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

// grid = (ceil(m / 64), ceil(n / 64)); block = 256 = 4 waves, wave w owns rows [64 * by + 16 * w, +16)
template <typename T>
__global__ __launch_bounds__(256) void svm_scores_kernel(const T *__restrict__ A, const T *__restrict__ W, const T *__restrict__ bias,
                                                         int64_t n, int64_t k, int64_t m, T *__restrict__ out)
{
    typedef MfmaTile<T> MT;
    constexpr int BK = 16, AP = BK + 1;
    __shared__ T As[2][64 * AP];            // [row][k], padded
    __shared__ T Ws[2][BK * 64];            // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.y * 64, col0 = (int64_t)blockIdx.x * 64;
    const int li = lane & 15, lk = lane >> 4;
    // my four elements of each staged slice: A row tid / 4, k (tid % 4) * 4 .. + 3;  W k tid / 16, columns (tid % 16) * 4 .. + 3
    const int ar = tid >> 2, ak = (tid & 3) * 4, wk = tid >> 4, wc = (tid & 15) * 4;
    const int64_t garow = row0 + ar;
    T pa[4], pw[4];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t kk = k0 + ak + i;
            pa[i] = (garow < n && kk < k) ? A[garow * k + kk] : (T)0;
        }
        const int64_t kw = k0 + wk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t c = col0 + wc + i;
            pw[i] = (kw < k && c < m) ? W[kw * m + c] : (T)0;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { As[buf][ar * AP + ak + i] = pa[i]; Ws[buf][wk * 64 + wc + i] = pw[i]; }
    };
    typename MT::acc_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = (T)0;
    fetch(0);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = 0; k0 < k; k0 += BK) {
        const bool more = k0 + BK < k;
        if (more) fetch(k0 + BK);               // in flight while this chunk is multiplied
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            const T a = As[buf][(16 * w + li) * AP + 4 * q + lk];
            T b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = Ws[buf][(4 * q + lk) * 64 + 16 * t + li];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = MT::mma(a, b[t], acc[t]);
        }
        if (more) stage(buf ^ 1);               // (the other buffer was last read before the previous barrier)
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int64_t c = col0 + 16 * t + li;
        if (c >= m) continue;
        const T bv = bias ? bias[c] : (T)0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t rr = row0 + 16 * w + MT::row(lane, r);
            if (rr < n) out[rr * m + c] = acc[t][r] + bv;
        }
    }
}

}  // namespace vdet
