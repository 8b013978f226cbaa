// temporal_kernels.hpp -- gfx950 kernels for the temporal passes over [F, S] series
// (S = B*C for a score volume [F,B,C], S = #tubelets for tubelet score tracks) and the float64
// IoU matrix.  Reference: vdet/tubelet_cls.py:386-414 (score_proto_temporal_maxpool),
// vdet/tubelet_cls.py:15-51 (score_conv_cls; external net -> build-defined op),
// utils/common.py:451-468 (iou).
//
// HBM-bound streaming kernels: one thread owns 4 adjacent series (16-B loads/stores, fully
// coalesced along S) and walks the frames with the window held in registers, so every input
// element is read from HBM exactly once and every output written once (8 B per element).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"     // score_key

namespace vdet {

struct Taps { float w[32]; };

// Batched videos (round 3): the frames of several videos concatenated along F.  seg[f] = {first frame, one past the
// last frame} of the video frame f belongs to; a temporal window never reaches across: frames outside the video
// count as padding, exactly like frames outside [0, F) of a single video.  seg == null: one video.
__device__ __forceinline__ bool seg_in(const int2 *__restrict__ seg, int64_t f, int64_t g, int64_t F)
{
    if (!seg) return g >= 0 && g < F;
    const int2 r = seg[f];
    return g >= (int64_t)r.x && g < (int64_t)r.y;
}

__device__ __forceinline__ float4 splat4(float v) { return make_float4(v, v, v, v); }

// np.max semantics: NaN propagates (v_max_f32 alone would drop it)
struct MaxAcc {
    float4 m; uint32_t nan;
    __device__ __forceinline__ void init(float4 v)
    {
        m = v;
        nan = (v.x != v.x) | ((v.y != v.y) << 1) | ((v.z != v.z) << 2) | ((v.w != v.w) << 3);
    }
    __device__ __forceinline__ void add(float4 v)
    {
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        nan |= (v.x != v.x) | ((v.y != v.y) << 1) | ((v.z != v.z) << 2) | ((v.w != v.w) << 3);
    }
    __device__ __forceinline__ float4 get() const
    {
        const float q = __uint_as_float(0x7FC00000u);
        return make_float4((nan & 1) ? q : m.x, (nan & 2) ? q : m.y, (nan & 4) ? q : m.z, (nan & 8) ? q : m.w);
    }
};

// MODE 0: max-pool, MODE 1: convolution.  W = window (odd).  Each block covers 256*4 series and
// the frame range [blockIdx.y*fchunk, ...) with a W/2 halo re-read at chunk borders.
template <int W, int MODE>
__global__ __launch_bounds__(256) void temporal_vec4_kernel(const float4 *__restrict__ in, float4 *__restrict__ out,
                                                            int64_t F, int64_t S4, int64_t fchunk, float pad,
                                                            float bias, Taps taps, const int2 *__restrict__ seg)
{
    constexpr int H = W / 2;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= S4) return;
    const int64_t f0 = (int64_t)blockIdx.y * fchunk;
    const int64_t f1 = min(F, f0 + fchunk);
    if (f0 >= f1) return;
    const float4 padv = splat4(pad);
    float4 win[W];   // win[k] = in[f - H + k]
    // out-of-range frames: always load a VALID (clamped) address, then select the pad value --
    // "cond ? in[i] : padv" makes hipcc select between a global and a scratch POINTER (flat load).
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
        const int64_t g = f0 - H + k;
        const int64_t gc = min(max(g, (int64_t)0), F - 1);
        const float4 v = in[gc * S4 + s];
        win[k + 1] = (g == gc) ? v : padv;
    }
    for (int64_t f = f0; f < f1; ++f) {
#pragma unroll
        for (int k = 0; k < W - 1; ++k) win[k] = win[k + 1];
        const int64_t g = f + H;
        const int64_t gc = min(g, F - 1);
        const float4 v = in[gc * S4 + s];
        win[W - 1] = (g == gc) ? v : padv;
        float4 wv[W];
#pragma unroll
        for (int k = 0; k < W; ++k) wv[k] = win[k];
        if (seg) {      // batched videos: neighbours in another video are padding (the raw values stay in the window)
#pragma unroll
            for (int k = 0; k < W; ++k) if (!seg_in(seg, f, f - H + k, F)) wv[k] = padv;
        }
        float4 r;
        if (MODE == 0) {
            MaxAcc a;
            a.init(wv[0]);
#pragma unroll
            for (int k = 1; k < W; ++k) a.add(wv[k]);
            r = a.get();
        } else {
            r = splat4(bias);
#pragma unroll
            for (int k = 0; k < W; ++k) {
                const float t = taps.w[k];
                r.x = r.x + t * wv[k].x; r.y = r.y + t * wv[k].y;
                r.z = r.z + t * wv[k].z; r.w = r.w + t * wv[k].w;
            }
        }
        out[f * S4 + s] = r;
    }
}

// Both temporal operators of one volume in ONE pass (same odd window W): the volume is read once
// instead of twice -- the pass is HBM-bound, so 12 instead of 16 bytes per element.  Identical
// arithmetic to MODE 0 / MODE 1 above (max-pool pad value / convolution pad value are separate).
template <int W>
__global__ __launch_bounds__(256) void temporal_both_vec4_kernel(const float4 *__restrict__ in, float4 *__restrict__ out_max,
                                                                 float4 *__restrict__ out_conv, int64_t F, int64_t S4,
                                                                 int64_t fchunk, float pad_max, float pad_conv, float bias,
                                                                 Taps taps, const int2 *__restrict__ seg)
{
    constexpr int H = W / 2;
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= S4) return;
    const int64_t f0 = (int64_t)blockIdx.y * fchunk;
    const int64_t f1 = min(F, f0 + fchunk);
    if (f0 >= f1) return;
    float4 win[W];      // raw values (clamped loads)
    bool ok[W];         // frame in range
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
        const int64_t g = f0 - H + k;
        const int64_t gc = min(max(g, (int64_t)0), F - 1);
        win[k + 1] = in[gc * S4 + s];
        ok[k + 1] = (g == gc);
    }
    const float4 pm = splat4(pad_max), pc = splat4(pad_conv);
    for (int64_t f = f0; f < f1; ++f) {
#pragma unroll
        for (int k = 0; k < W - 1; ++k) { win[k] = win[k + 1]; ok[k] = ok[k + 1]; }
        const int64_t g = f + H;
        const int64_t gc = min(g, F - 1);
        win[W - 1] = in[gc * S4 + s];
        ok[W - 1] = (g == gc);
        bool okf[W];
#pragma unroll
        for (int k = 0; k < W; ++k) okf[k] = seg ? seg_in(seg, f, f - H + k, F) : ok[k];
        MaxAcc a;
        a.init(okf[0] ? win[0] : pm);
#pragma unroll
        for (int k = 1; k < W; ++k) a.add(okf[k] ? win[k] : pm);
        out_max[f * S4 + s] = a.get();
        float4 r = splat4(bias);
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const float t = taps.w[k];
            const float4 v = okf[k] ? win[k] : pc;
            r.x = r.x + t * v.x; r.y = r.y + t * v.y;
            r.z = r.z + t * v.z; r.w = r.w + t * v.w;
        }
        out_conv[f * S4 + s] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// The ONE pass over a score volume [F,B,C] (class innermost, as zs[B,C] of utils/protocol.py:538):
// per element it is read once and produces
//   out_max  [F,B,C]  temporal max-pool  (score_proto_temporal_maxpool, vdet/tubelet_cls.py:386-414)
//   out_conv [F,B,C]  temporal convolution (stand-in for the external TCN of :15-51), optional
//   keys     [F,C,B]  class-major sortable keys of every (frame, class) NMS problem (what
//                     transpose_keys_kernel produced from a second read of the volume)
// = 4 B in, 12 B out per element instead of 8 B in, 12 B out (and no strided 256-B reads: a block
// owns TB whole rows of C scores, which are contiguous in memory).
// Block = NT threads, tile = TB boxes x C classes (C % 4 == 0), item i of thread t is float4
// number t + NT*i of the tile; a thread walks the frames of its chunk with the window of every
// item in registers (same arithmetic as temporal_both_vec4_kernel).  The keys of the centre frame
// go through a double-buffered LDS tile [C][TB] (one barrier per frame; columns XOR-swizzled against bank
// conflicts) and leave as 16-byte stores, TB*4 contiguous bytes per class row.
// ------------------------------------------------------------------------------------------------
// the outputs are written once and read by other kernels much later: streaming (non-temporal) stores
// (measured: pass 1.9-2.0 -> 1.7-1.9 ms one video at a time against plain stores)
typedef float vp_f4 __attribute__((ext_vector_type(4)));
typedef uint32_t vp_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vp_store(float4 *p, const float4 v)
{
    vp_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<vp_f4 *>(p));
}
__device__ __forceinline__ void vp_store_u4(uint4 *p, const uint4 v)
{
    vp_u4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<vp_u4 *>(p));
}

template <int W, int ITEMS, bool CONV, int NT>
__global__ __launch_bounds__(NT) void volume_pass_kernel(const float4 *__restrict__ in, float4 *__restrict__ out_max,
                                                         float4 *__restrict__ out_conv, uint32_t *__restrict__ keys,
                                                         int F, int B, int C4, int TB, int tb_shift, int fchunk,
                                                         float pad_max, float pad_conv, float bias, Taps taps,
                                                         int use_thr, float thr, const int2 *__restrict__ seg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char vp_smem[];
    uint32_t *tile = reinterpret_cast<uint32_t *>(vp_smem);      // [2][C][TB] keys, columns XOR-swizzled per class group
    constexpr int H = W / 2;
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * TB;
    const int rows = min(TB, B - b0);
    const int n4 = rows * C4;                                    // valid float4 of this tile
    const int f0 = blockIdx.y * fchunk;
    const int f1 = min(F, f0 + fchunk);
    if (f0 >= f1) return;
    const int64_t S4 = (int64_t)B * C4;
    const int tile_sz = C4 * 4 * TB;
    const int qmask = (TB >> 2) - 1;                             // 16-byte chunks per key row - 1
    // everything per item is a 32-bit offset from a per-frame (wave-uniform) base: the frame's slab is < 2^31 B
    const float4 *in_t = in + (int64_t)b0 * C4;
    float4 *om_t = out_max + (int64_t)b0 * C4;
    float4 *oc_t = CONV ? out_conv + (int64_t)b0 * C4 : nullptr;

    int goff[ITEMS];          // float4 offset inside the tile (clamped: every load address is valid)
    int loff[ITEMS];          // LDS slot of the item's keys: [c4][b]
    bool on[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int idx = tid + NT * i;
        on[i] = idx < n4;
        const int idc = min(idx, n4 - 1);
        const int b = idc / C4, c4 = idc - b * C4;
        goff[i] = idc;
        // key (class 4*c4 + j, box b) sits at dword [(4*c4 + j) * TB + (b ^ s)], s = (c4 mod TB/4) * 4: the lanes of a wave
        // (consecutive c4, same b) then spread over the banks (4-way instead of 32-way), and the 4 boxes of a 16-byte chunk
        // stay together for the ds_read_b128 of the copy-out
        loff[i] = 4 * c4 * TB + (b ^ ((c4 & qmask) << 2));
    }
    float4 win[ITEMS][W];
    float4 nxt[ITEMS];        // frame f + H + 1, in flight while frame f is processed
    bool ok[W];
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
        const int g = f0 - H + k;
        const int gc = min(max(g, 0), F - 1);
        const float4 *fin = in_t + (int64_t)gc * S4;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) win[i][k + 1] = fin[goff[i]];
        ok[k + 1] = (g == gc);
    }
    {
        const float4 *fin = in_t + (int64_t)min(f0 + H, F - 1) * S4;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) nxt[i] = fin[goff[i]];
    }
    const float4 pm = splat4(pad_max), pc = splat4(pad_conv);
    for (int f = f0; f < f1; ++f) {
#pragma unroll
        for (int k = 0; k < W - 1; ++k) {
            ok[k] = ok[k + 1];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) win[i][k] = win[i][k + 1];
        }
        ok[W - 1] = (f + H <= F - 1);
        if (seg) {      // batched videos: the window stops at the video's first / last frame (wave-uniform: scalar loads)
#pragma unroll
            for (int k = 0; k < W; ++k) ok[k] = seg_in(seg, f, f - H + k, F);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) win[i][W - 1] = nxt[i];
        {   // next iteration's newest frame: issued now, first used after this iteration's stores and barrier
            const float4 *fin = in_t + (int64_t)min(f + 1 + H, F - 1) * S4;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) nxt[i] = fin[goff[i]];
        }
        uint32_t *tb = tile + (f & 1) * tile_sz;
        float4 *om = om_t + (int64_t)f * S4;
        float4 *oc = CONV ? oc_t + (int64_t)f * S4 : nullptr;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            if (!on[i]) continue;
            MaxAcc a;
            a.init(ok[0] ? win[i][0] : pm);
#pragma unroll
            for (int k = 1; k < W; ++k) a.add(ok[k] ? win[i][k] : pm);
            vp_store(om + goff[i], a.get());
            if (CONV) {
                float4 r = splat4(bias);
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    const float t = taps.w[k];
                    const float4 v = ok[k] ? win[i][k] : pc;
                    r.x = r.x + t * v.x; r.y = r.y + t * v.y;
                    r.z = r.z + t * v.z; r.w = r.w + t * v.w;
                }
                vp_store(oc + goff[i], r);
            }
            const float4 s = win[i][H];
            uint4 k4 = make_uint4(score_key(s.x), score_key(s.y), score_key(s.z), score_key(s.w));
            if (use_thr) {
                if (!(s.x > thr)) k4.x = 0u;
                if (!(s.y > thr)) k4.y = 0u;
                if (!(s.z > thr)) k4.z = 0u;
                if (!(s.w > thr)) k4.w = 0u;
            }
            tb[loff[i]] = k4.x; tb[loff[i] + TB] = k4.y; tb[loff[i] + 2 * TB] = k4.z; tb[loff[i] + 3 * TB] = k4.w;
        }
        __syncthreads();       // (the other buffer was last read one iteration ago, before this barrier's predecessor)
        uint32_t *kf = keys + (int64_t)f * C4 * 4 * B + b0;
        const bool vec_ok = (B & 3) == 0;                        // every key row starts 16-byte aligned
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int idx = tid + NT * i;                        // now (class c, 16-byte chunk q of its row), q fastest:
            const int cc = idx >> (tb_shift - 2), q = idx & qmask;   // a quad of lanes stores 64 contiguous bytes
            if (cc < C4 * 4 && 4 * q < rows) {
                const uint4 k4 = *reinterpret_cast<const uint4 *>(tb + cc * TB + ((4 * q) ^ (((cc >> 2) & qmask) << 2)));
                const int o = cc * B + 4 * q;
                if (vec_ok && 4 * q + 3 < rows) {
                    vp_store_u4(reinterpret_cast<uint4 *>(kf + o), k4);
                } else {
                    kf[o] = k4.x;
                    if (4 * q + 1 < rows) kf[o + 1] = k4.y;
                    if (4 * q + 2 < rows) kf[o + 2] = k4.z;
                    if (4 * q + 3 < rows) kf[o + 3] = k4.w;
                }
            }
        }
    }
}

// Generic fallback: any odd window, any S (no alignment requirement); one thread per element.
template <int MODE>
__global__ __launch_bounds__(256) void temporal_scalar_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                              int64_t F, int64_t S, int W, float pad, float bias,
                                                              Taps taps, const int2 *__restrict__ seg)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= F * S) return;
    const int64_t f = i / S, s = i - f * S;
    const int H = W / 2;
    if (MODE == 0) {
        float m = 0.0f;
        bool nan = false;
        for (int k = 0; k < W; ++k) {
            const int64_t g = f + k - H;
            const int64_t gc = min(max(g, (int64_t)0), F - 1);
            float v = in[gc * S + s];
            v = (g == gc && seg_in(seg, f, g, F)) ? v : pad;
            nan |= (v != v);
            m = (k == 0) ? v : fmaxf(m, v);
        }
        out[i] = nan ? __uint_as_float(0x7FC00000u) : m;
    } else {
        float acc = bias;
        for (int k = 0; k < W; ++k) {
            const int64_t g = f + k - H;
            const int64_t gc = min(max(g, (int64_t)0), F - 1);
            float v = in[gc * S + s];
            v = (g == gc && seg_in(seg, f, g, F)) ? v : pad;
            acc = acc + taps.w[k] * v;
        }
        out[i] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// utils/common.py:451-468  iou(boxes1, boxes2) -> [n1, n2] float64
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double np_maximum(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }
__device__ __forceinline__ double np_minimum(double a, double b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

__device__ __forceinline__ double iou_f64_pair(const double *p, const double *q)
{
    const double ix1 = np_maximum(p[0], q[0]);
    const double ix2 = np_minimum(p[2], q[2]);
    const double iy1 = np_maximum(p[1], q[1]);
    const double iy2 = np_minimum(p[3], q[3]);
    const double iw = np_maximum(0.0, (ix2 - ix1) + 1.0);
    const double ih = np_maximum(0.0, (iy2 - iy1) + 1.0);
    const double a1 = ((p[2] - p[0]) + 1.0) * ((p[3] - p[1]) + 1.0);
    const double a2 = ((q[2] - q[0]) + 1.0) * ((q[3] - q[1]) + 1.0);
    const double inter = iw * ih;
    return inter / ((a1 + a2) - inter);
}

__global__ __launch_bounds__(256) void iou_f64_kernel(const double *__restrict__ b1, int64_t n1,
                                                      const double *__restrict__ b2, int64_t n2,
                                                      double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n2) return;
    for (int64_t i = blockIdx.y; i < n1; i += gridDim.y)
        out[i * n2 + j] = iou_f64_pair(b1 + 4 * i, b2 + 4 * j);
}

}  // namespace vdet
