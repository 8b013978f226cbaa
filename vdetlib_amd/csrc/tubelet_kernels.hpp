// tubelet_kernels.hpp -- gfx950 kernels for the tubelet re-scoring cores of vdet/tubelet_cls.py and
// the per-class threshold / top-k of vdet/video_det.py:89-99.  All float64 (the reference computes
// these on python floats / numpy float64), operation order = the reference's, -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "temporal_kernels.hpp"   // iou_f64_pair

namespace vdet {

// np.argmax order: NaN beats everything, first occurrence wins
__device__ __forceinline__ bool argmax_better(double s, int64_t i, double bs, int64_t bi)
{
    if (bi < 0) return true;
    const bool sn = s != s, bn = bs != bs;
    if (sn || bn) return sn && (!bn || i < bi);
    return s > bs || (s == bs && i < bi);
}

// ------------------------------------------------------------------------------------------------
// Spatial max-pooling core (vdet/tubelet_cls.py:514-532 / :327-347): for one tubelet box, among the
// detections of its frame with iou > thres (strict, float64; iou([cur_bbox], det_boxes),
// utils/common.py:451-468) the first arg-max of the class score.  One block per tubelet box.
// out_idx = index within the frame's detections, -1 when nothing overlaps.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spatial_maxpool_kernel(const double *__restrict__ tub_boxes,
                                                              const int32_t *__restrict__ tub_group,
                                                              const double *__restrict__ det_boxes,
                                                              const double *__restrict__ det_scores,
                                                              const int64_t *__restrict__ group_off, double thres,
                                                              int64_t *__restrict__ out_idx,
                                                              double *__restrict__ out_score)
{
    __shared__ double ss[256];
    __shared__ long long si[256];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int g = tub_group[t];
    const int64_t o = group_off[g], n = group_off[g + 1] - o;
    const double *p = tub_boxes + 4 * (int64_t)t;
    double bs = 0.0;
    int64_t bi = -1;
    for (int64_t j = tid; j < n; j += 256) {
        const double ov = iou_f64_pair(p, det_boxes + 4 * (o + j));
        if (ov > thres) {
            const double s = det_scores[o + j];
            if (argmax_better(s, j, bs, bi)) { bs = s; bi = j; }
        }
    }
    ss[tid] = bs;
    si[tid] = bi;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) {
            const double s2 = ss[tid + d];
            const int64_t i2 = si[tid + d];
            if (i2 >= 0 && argmax_better(s2, i2, ss[tid], si[tid])) { ss[tid] = s2; si[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) { out_idx[t] = si[0]; out_score[t] = si[0] >= 0 ? ss[0] : -1e5; }
}

// ------------------------------------------------------------------------------------------------
// do_score_completion (vdet/tubelet_cls.py:284-303), one thread per tubelet, in place:
// runs of det_score <= -10: leading -> first valid value, trailing -> last valid, interior ->
// l + (r - l) * (k - i + 1) / (j - i + 1).  A tubelet that is missing entirely makes the reference
// index boxes[len(boxes)] -> IndexError: flagged in *err.
// ------------------------------------------------------------------------------------------------
__global__ void series_completion_kernel(double *__restrict__ v, const int64_t *__restrict__ off, int64_t T,
                                         int *__restrict__ err)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    double *s = v + off[t];
    const int64_t n = off[t + 1] - off[t];
    for (int64_t i = 0; i < n; ++i) {
        if (s[i] > -10) continue;
        int64_t j = i;
        while (j < n && s[j] <= -10) ++j;
        if (i == 0) {
            if (j == n) { atomicOr(err, 1); return; }
            for (int64_t k = i; k < j; ++k) s[k] = s[j];
        } else if (j == n) {
            for (int64_t k = i; k < j; ++k) s[k] = s[i - 1];
        } else {
            const double l = s[i - 1], r = s[j];
            for (int64_t k = i; k < j; ++k) s[k] = l + (r - l) * (double)(k - i + 1) / (double)(j - i + 1);
        }
    }
}

// score_proto_temporal_maxpool core (vdet/tubelet_cls.py:399-412) on ragged float64 series.
__global__ void series_maxpool_kernel(const double *__restrict__ in, double *__restrict__ out,
                                      const int64_t *__restrict__ off, const int32_t *__restrict__ elem_series,
                                      int64_t total, int window, double pad)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int t = elem_series[e];
    const int64_t o = off[t], n = off[t + 1] - o, i = e - o;
    const int h = window / 2;
    double m = 0.0;
    bool nan = false;
    for (int d = -h; d <= h; ++d) {
        const int64_t g = i + d;
        const int64_t gc = g < 0 ? 0 : (g >= n ? n - 1 : g);
        double x = in[o + gc];
        x = (g == gc) ? x : pad;
        nan |= (x != x);
        m = (d == -h) ? x : (x > m ? x : m);
    }
    out[e] = nan ? __longlong_as_double(0x7FF8000000000000ll) : m;
}

// ------------------------------------------------------------------------------------------------
// score_proto_interpolation core (vdet/tubelet_cls.py:453-487): scipy interp1d(kind='linear') --
// numpy.interp semantics (exact y at a knot, else slope*(x - x_lo) + y_lo) -- plus extrap1d's
// one-step linear extrapolation (:416-428).  One thread per (query, field).
//   knots: x[koff[t] .. koff[t+1]) ascending, y[K][...] field-major per tubelet: y[(koff[t]*K) + f*L + k]
//   queries: q[qoff[t] .. qoff[t+1]); out[(qoff[t]*K) + f*Lq + n]
// ------------------------------------------------------------------------------------------------
__global__ void series_interp_kernel(const double *__restrict__ x, const double *__restrict__ y,
                                     const int64_t *__restrict__ koff, const double *__restrict__ q,
                                     const int64_t *__restrict__ qoff, const int32_t *__restrict__ query_series,
                                     int64_t total_q, int K, double *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total_q * K) return;
    const int64_t qi = e / K;
    const int f = (int)(e - qi * K);
    const int t = query_series[qi];
    const int64_t ko = koff[t], L = koff[t + 1] - ko;
    const int64_t qo = qoff[t], Lq = qoff[t + 1] - qo;
    const double *xs = x + ko;
    const double *ys = y + ko * K + (int64_t)f * L;
    const double xv = q[qi];
    double r;
    if (xv < xs[0]) {
        r = ys[0] + (xv - xs[0]) * (ys[1] - ys[0]) / (xs[1] - xs[0]);
    } else if (xv > xs[L - 1]) {
        r = ys[L - 1] + (xv - xs[L - 1]) * (ys[L - 1] - ys[L - 2]) / (xs[L - 1] - xs[L - 2]);
    } else {
        int64_t lo = 0, hi = L;                 // largest j with xs[j] <= xv
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (xs[mid] <= xv) lo = mid; else hi = mid;
        }
        const int64_t j = lo;
        if (j == L - 1 || xs[j] == xv) r = ys[j];
        else {
            const double slope = (ys[j + 1] - ys[j]) / (xs[j + 1] - xs[j]);
            r = slope * (xv - xs[j]) + ys[j];
        }
    }
    out[qo * K + (int64_t)f * Lq + (qi - qo)] = r;
}

// ------------------------------------------------------------------------------------------------
// Per-class threshold + top-k of one frame (vdet/video_det.py:89-99): for class column j,
// inds = where(scores[:, j] > thresh); more than k -> the k best by argsort(-scores) (stable:
// ties by ascending index), in that order; else all of inds in ascending order.
// One block per class; candidates are compacted in order into LDS, ranks by counting.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void threshold_topk_kernel(const T *__restrict__ scores, int64_t B, int64_t ld,
                                                             int col0, double thresh, int k,
                                                             int32_t *__restrict__ out_idx,
                                                             int32_t *__restrict__ out_cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // all scratch in the dynamic region (a static __shared__ in front would misalign it)
    const size_t csB = (sizeof(T) * B + 15) & ~(size_t)15, ciB = ((size_t)4 * B + 15) & ~(size_t)15;
    T *cs = reinterpret_cast<T *>(smem);                                   // [B] candidate scores
    int32_t *ci = reinterpret_cast<int32_t *>(smem + csB);                 // [B] candidate indices
    uint32_t *sscan = reinterpret_cast<uint32_t *>(smem + csB + ciB);      // [256]
    uint32_t &srun = sscan[256];
    const int cls = blockIdx.x, col = col0 + cls, tid = threadIdx.x;
    if (tid == 0) srun = 0;
    __syncthreads();
    // ordered compaction, 256 rows at a time
    for (int64_t b0 = 0; b0 < B; b0 += 256) {
        const int64_t b = b0 + tid;
        T s = 0;
        bool c = false;
        // np.where(scores[:, j] > thresh) with a python-float thresh (vdet/video_det.py:90): numpy compares in the
        // ARRAY's dtype, i.e. float32 scores against float32(thresh) (round to nearest), float64 against thresh
        if (b < B) { s = scores[b * ld + col]; c = s > (T)thresh; }
        sscan[tid] = c ? 1u : 0u;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint32_t t = (tid >= d) ? sscan[tid - d] : 0u;
            __syncthreads();
            sscan[tid] += t;
            __syncthreads();
        }
        const uint32_t base = srun;
        if (c) { const uint32_t pos = base + sscan[tid] - 1; cs[pos] = s; ci[pos] = (int32_t)b; }
        __syncthreads();
        if (tid == 255) srun = base + sscan[255];
        __syncthreads();
    }
    const int n = (int)srun;
    int32_t *out = out_idx + (int64_t)cls * k;
    if (n <= k) {
        for (int i = tid; i < n; i += 256) out[i] = ci[i];
        if (tid == 0) out_cnt[cls] = n;
        return;
    }
    for (int i = tid; i < n; i += 256) {
        const T si = cs[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const T sj = cs[j];
            // argsort(-s, stable): larger score first, ties by ascending position; NaN last
            const bool before = (sj > si) || (sj == si && j < i) || (si != si && sj == sj) || (si != si && sj != sj && j < i);
            rank += before ? 1 : 0;
        }
        if (rank < k) out[rank] = ci[i];
    }
    if (tid == 0) out_cnt[cls] = k;
}

// ------------------------------------------------------------------------------------------------
// Temporal convolution layer of the tubelet TCN (stand-in for the external Caffe net behind
// score_conv_cls, vdet/tubelet_cls.py:15-51; parity UNPINNED: no prototxt/weights in the reference).
//   out[co, l] = act( b[co] + sum_ci sum_k w[co, ci, k] * in[ci, l + k - K/2] )   ("same" zero padding)
// accumulated in exactly that order (ci outer, k inner) in f32 without contraction.
// act: 0 none, 1 ReLU.  One thread per output element.
// ------------------------------------------------------------------------------------------------
__global__ void conv1d_kernel(const float *__restrict__ in, int Cin, int L, const float *__restrict__ w,
                              const float *__restrict__ b, int Cout, int K, int act, float *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cout * L) return;
    const int co = e / L, l = e - co * L;
    const int h = K / 2;
    float acc = b[co];
    for (int ci = 0; ci < Cin; ++ci)
        for (int k = 0; k < K; ++k) {
            const int g = l + k - h;
            const float x = (g >= 0 && g < L) ? in[ci * L + min(max(g, 0), L - 1)] : 0.0f;
            const float p = w[(co * Cin + ci) * K + k] * x;
            acc = acc + p;
        }
    if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
    out[e] = acc;
}

// softmax over the channel axis of [Cout, L] (Caffe SoftmaxLayer: subtract the max, exp, normalise)
__global__ void softmax_channels_kernel(const float *__restrict__ in, int Cout, int L, float *__restrict__ out)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float m = in[l];
    for (int c = 1; c < Cout; ++c) m = fmaxf(m, in[c * L + l]);
    float sum = 0.0f;
    for (int c = 0; c < Cout; ++c) sum = sum + expf(in[c * L + l] - m);
    for (int c = 0; c < Cout; ++c) out[c * L + l] = expf(in[c * L + l] - m) / sum;
}

}  // namespace vdet
