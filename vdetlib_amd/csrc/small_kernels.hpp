// small_kernels.hpp -- (round 4) frames of at most 384 proposals: the ILSVRC-VID shape (<= 300 per frame, BASELINE configs[0]
// and [4]) and every other small problem of utils/nms.pyx:17-68 / vdet/video_det.py:89-99.
//
// The walk of nms_kernels.hpp is built for lists of ~10 000 candidates: one WAVE per list, eight survivors per pass of vector
// code.  On a batch of 64 VID-shaped videos (972 000 lists of <= 300 candidates) it costs ~2 400 instructions per list and was
// 4.3 of 20.5 ms (profiles/r04_vid_batch_kernel_stats.csv).  A small list does not need a wave:
//
//   small_walk_kernel   ONE LANE per list.  A wave takes the lists of one frame (C > 32) or of several (64 / C frames); the
//                       frames' suppression ROWS (bit v of row u <=> u suppresses v) are built in LDS from the adjacency
//                       lists once and serve all their classes.  Every lane runs the reference's loop (utils/nms.pyx:33-66) on
//                       its own list: next candidate, test its bit in the lane's dead mask (LDS), and -- if it is clear --
//                       keep it and OR its row into the mask (three 16-byte LDS reads, twelve ds_or).  ~35 instructions per
//                       step of 64 lists instead of ~8 per candidate of one.
//   walk_rest_kernel    what small_walk_kernel leaves: the lists of irregular frames (NaN / degenerate boxes: zero-union
//                       tags, asymmetric rows), through the general walk, one wave per list.
//
// Results are identical to the large-list walk's (tests run both: VDET_SMALL_LISTS=0).
// Measured and dropped in the same round: ONE WAVE per list sorting it by counting (every lane counts, for each of its <= 6
// keys, the keys before it: N^2 / 64 compare + add-with-carry pairs per lane) -- 11.5 ms against the LSD kernel's 3.7 on that
// batch (a wave64 instruction takes four cycles: 3 500 of them per list are 6 ms of pure issue), and the first small walk,
// one block per frame with a scalar loop over the alive candidates of one list per wave (7.6 ms against 4.3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"

namespace vdet {

constexpr int kSmallMax = 384;          // boxes per frame: rows of at most 12 words

__device__ __forceinline__ bool small_walk_takes(const WalkParams &prm, int g)
{
    return prm.group_flags && (prm.group_flags[g] & kFlagRegular) && prm.group_z[g] == 0u;
}

// NQ = 16-byte pieces of a row (N <= 128 * NQ).  One wave per block; it serves `fpw` frames (fpw * C <= 64, or fpw == 1 and
// the lanes take the classes in rounds of 64).  Dynamic LDS: [fpw][nmax][4 NQ] row words, then 64 masks of 4 NQ + 1 words.
template <int NQ>
__global__ __launch_bounds__(64) void small_walk_kernel(const WalkParams prm, int G, int fpw, int nmax)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slds[];
    constexpr int W32 = 4 * NQ, MS = W32 + 1;
    typedef __attribute__((address_space(3))) uint32_t lds_word;
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * fpw;
    uint32_t *rows = slds;
    uint32_t *masks = slds + (size_t)fpw * nmax * W32;
    // ---- the frames' rows, from the adjacency lists (16-byte pieces: lists are aligned and padded to 8 entries with copies)
    for (int j = 0; j < fpw; ++j) {
        const int g = g0 + j;
        if (g >= G || !small_walk_takes(prm, g)) continue;              // (wave-uniform)
        const GroupDesc gd = prm.groups[g];
        const int N = gd.nbox, rb = gd.box_off;
        uint32_t *rj = rows + (size_t)j * nmax * W32;
        for (int i = lane; i < N * W32; i += 64) rj[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int v = lane; v < N; v += 64) {
            const uint2 meta = prm.row_meta[rb + v];
            const AdjVec *pa = reinterpret_cast<const AdjVec *>(prm.adj + meta.x);
            const int np = ((int)meta.y + 7) >> 3;
            for (int i = 0; i < np; ++i) {
                const AdjVec a = pa[i];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t e0 = a.v[t] & 0x7FFFu, e1 = (a.v[t] >> 16) & 0x7FFFu;
                    rj[v * W32 + (e0 >> 5)] |= 1u << (e0 & 31u);
                    rj[v * W32 + (e1 >> 5)] |= 1u << (e1 & 31u);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- one list per lane
    const int C = prm.C;
    const int rounds = fpw > 1 ? 1 : (C + 63) >> 6;
    uint32_t *msk = masks + lane * MS;
    for (int r = 0; r < rounds; ++r) {
        const int j = fpw > 1 ? lane / C : 0;
        const int cls = fpw > 1 ? lane - j * C : lane + 64 * r;
        const int g = g0 + j;
        const bool mine = j < fpw && cls < C && g < G && small_walk_takes(prm, g);
        const int p = mine ? g * C + cls : 0;
        const int ncand = mine ? prm.ncand[p] : 0;
        const uint16_t *order = prm.order + (int64_t)p * prm.B;
        int32_t *out = prm.keep_idx + (int64_t)p * prm.cap;
        const uint32_t *rj = rows + (size_t)j * nmax * W32;
        const int last = max(ncand - 1, 0);
#pragma unroll
        for (int i = 0; i < W32; ++i) msk[i] = 0u;
        int nk = 0;
        int maxn = ncand;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) maxn = max(maxn, __shfl_xor(maxn, d, 64));
        maxn = __builtin_amdgcn_readfirstlane(maxn);
        // candidates four at a time, two register sets taken in turn (the next four are requested before these are walked)
        int ca[4], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ca[i] = (int)order[min(i, last)];
#define VDET_SMALL_STEP(CAND, Q)                                                                          \
        {                                                                                                     \
            const int c_ = (CAND);                                                                            \
            const uint32_t wd_ = msk[c_ >> 5];                                                                \
            if ((Q) < ncand && !((wd_ >> (c_ & 31)) & 1u)) {                                                  \
                if ((int64_t)nk < prm.cap) out[nk] = c_;                                                      \
                ++nk;                                                                                         \
                const uint4 *rw_ = reinterpret_cast<const uint4 *>(rj + c_ * W32);                            \
                _Pragma("unroll") for (int i_ = 0; i_ < NQ; ++i_) {                                           \
                    const uint4 x_ = rw_[i_];                                                                 \
                    __hip_atomic_fetch_or((lds_word *)(msk + 4 * i_ + 0), x_.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
                    __hip_atomic_fetch_or((lds_word *)(msk + 4 * i_ + 1), x_.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
                    __hip_atomic_fetch_or((lds_word *)(msk + 4 * i_ + 2), x_.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
                    __hip_atomic_fetch_or((lds_word *)(msk + 4 * i_ + 3), x_.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
                }                                                                                             \
            }                                                                                                 \
        }
        for (int q = 0; q < maxn; q += 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cb[i] = (int)order[min(q + 4 + i, last)];
#pragma unroll
            for (int i = 0; i < 4; ++i) VDET_SMALL_STEP(ca[i], q + i)
#pragma unroll
            for (int i = 0; i < 4; ++i) ca[i] = (int)order[min(q + 8 + i, last)];
#pragma unroll
            for (int i = 0; i < 4; ++i) VDET_SMALL_STEP(cb[i], q + 4 + i)
        }
#undef VDET_SMALL_STEP
        if (mine) {
            prm.keep_cnt[p] = nk;
            if ((int64_t)nk > prm.cap) atomicOr(prm.status, kStCap);
        }
    }
}

// the lists of the frames small_walk_kernel did not take, through the general walk: blocks stride over the frames, the four
// waves of a block over a frame's classes
__global__ __launch_bounds__(256) void walk_rest_kernel(const WalkParams prm, int G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        if (small_walk_takes(prm, g)) continue;
        for (int cls = w; cls < prm.C; cls += 4) walk_one(prm, g * prm.C + cls, smem, lane, w);
    }
}

}  // namespace vdet
