// detnms_kernels.hpp -- the Fast R-CNN per-class flow on the device (round 4):
//   vdet/video_det.py:89-99   for every frame and class j: rows = {boxes[i, 4j:4j+4], scores[i, j]} of the boxes with
//                             score > thresh; more than max_per_image -> the best max_per_image by argsort(-score)
//   vdet/image_det.py:117-123 apply_image_nms of those rows (utils/nms.pyx:17-68)
// Every class suppresses ITS OWN regressed boxes, so there is no suppression graph to share between the classes of a
// frame (the volume kernels' [F,B,4] geometry): each (frame, class) is a small dense problem of <= 128 boxes.
// One WAVE per problem: the selected boxes (selection = the LSD sort's threshold + top-k, sort_kernel) are staged in
// LDS in score order, lane r evaluates rows r and r + 64 of the upper-triangular suppression matrix with the exact
// pair predicate (pair_pred: the reference's f32 operation order, IEEE division, zero-union flag) into 128-bit row
// masks held in registers, and the greedy pass is a scalar loop over the <= 128 candidates that ORs the surviving rows'
// masks (lane broadcasts) into the dead mask -- no LDS traffic, no barrier.  ZeroDivisionError (Cython cdivision =
// False) is latched iff a zero-union pair (i kept, j later and not yet suppressed) is evaluated, like the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"

namespace vdet {

constexpr int kDetMax = 128;      // selected boxes per (frame, class): two per lane

struct DetNmsParams {
    const float4 *boxes;          // [F,B,K] per-class boxes (the reference's [B, 4K] row per frame)
    const float *scores;          // [F,B,K]
    int F, B, K, class0;          // classes < class0 are skipped (background), their counts are 0
    const uint16_t *order;        // [F*K, B] candidates in descending score order (sort_kernel with threshold + topk)
    const int32_t *ncand;         // [F*K] <= topk
    const int32_t *nover;         // [F*K] candidates before the cut
    int topk;
    float t32;
    float *dets;                  // [F,K,topk,5] rows (x1,y1,x2,y2,score) in the REFERENCE's row order, or null
    int32_t *sel_idx;             // [F,K,topk] box index of every row, or null
    int32_t *det_cnt;             // [F,K]
    int32_t *keep;                // [F,K,topk] kept ROW positions, descending score (what apply_image_nms returns)
    int32_t *keep_cnt;            // [F,K]
    int *status;
};

__global__ __launch_bounds__(256) void det_nms_kernel(const DetNmsParams prm)
{
    __shared__ float4 sbox[4][kDetMax];
    __shared__ int sidx[4][kDetMax];
    __shared__ uint32_t skey[4][kDetMax];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.x * 4 + w;
    if (p >= prm.F * prm.K) return;
    const int f = p / prm.K, j = p - f * prm.K;
    if (j < prm.class0) {
        if (lane == 0) { prm.det_cnt[p] = 0; prm.keep_cnt[p] = 0; }
        return;
    }
    const int M = prm.ncand[p];                     // (<= topk <= kDetMax: host)
    const bool by_index = prm.nover[p] <= prm.topk; // vdet/video_det.py:93: rows stay in box order unless the cut applies
    const uint16_t *ord = prm.order + (int64_t)p * prm.B;
    float4 bx[2];
    float sc[2];
    int id[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        id[h] = q < M ? (int)ord[q] : 0x7FFFFFFF;
        bx[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        sc[h] = 0.f;
        if (q < M) {
            const int64_t e = ((int64_t)f * prm.B + id[h]) * prm.K + j;
            bx[h] = prm.boxes[e];
            sc[h] = prm.scores[e];
            sbox[w][q] = bx[h];
            sidx[w][q] = id[h];
            skey[w][q] = score_key(sc[h]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // row position of candidate q in the reference's array: its rank by box index, or -- after the cut -- its position in
    // argsort(-score)[:k] (vdet/video_det.py:93-97; stable: equal scores by ASCENDING box index).  The list holds a run of
    // equal scores [a, b) by DESCENDING index (the order the NMS of those rows visits them: ties by descending row), so
    // inside a run the row positions are mirrored: row = a + (b - 1 - q)
    int row[2] = {lane, lane + 64};
    if (by_index) {
        row[0] = row[1] = 0;
        for (int q = 0; q < M; ++q) {
            const int v = sidx[w][q];
            row[0] += v < id[0] ? 1 : 0;
            row[1] += v < id[1] ? 1 : 0;
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            if (q < M) {
                const uint32_t k = skey[w][q];
                int a = q, b = q + 1;
                while (a > 0 && skey[w][a - 1] == k) --a;
                while (b < M && skey[w][b] == k) ++b;
                row[h] = a + (b - 1 - q);
            }
        }
    }
    const int64_t obase = (int64_t)p * prm.topk;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (lane + 64 * h < M) {
            if (prm.dets) {
                float *d = prm.dets + (obase + row[h]) * 5;
                d[0] = bx[h].x; d[1] = bx[h].y; d[2] = bx[h].z; d[3] = bx[h].w; d[4] = sc[h];
            }
            if (prm.sel_idx) prm.sel_idx[obase + row[h]] = id[h];
        }
    }
    // suppression rows: bit c of (lo, hi) of candidate r <=> r (kept, box "i") suppresses the later candidate c (box "j")
    unsigned long long slo[2] = {0ull, 0ull}, shi[2] = {0ull, 0ull}, zlo[2] = {0ull, 0ull}, zhi[2] = {0ull, 0ull};
    const float ar[2] = {box_area(bx[0]), box_area(bx[1])};
    for (int c = 1; c < M; ++c) {
        const float4 bc = sbox[w][c];
        const float ac = box_area(bc);
        const unsigned long long bit = 1ull << (c & 63);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = lane + 64 * h;
            if (r < c && r < M) {
                const uint32_t pr = pair_pred(bx[h], ar[h], bc, ac, prm.t32);
                if (pr & 1u) { if (c < 64) slo[h] |= bit; else shi[h] |= bit; }
                if (pr & 2u) { if (c < 64) zlo[h] |= bit; else zhi[h] |= bit; }
            }
        }
    }
    // the greedy pass (utils/nms.pyx:33-66), wave-uniform: dead = suppressed by a kept box so far
    unsigned long long dlo = 0ull, dhi = 0ull, klo = 0ull, khi = 0ull;
    int bad = 0;
    for (int i = 0; i < M; ++i) {
        const bool dead = i < 64 ? ((dlo >> i) & 1ull) : ((dhi >> (i - 64)) & 1ull);
        if (dead) continue;
        if (i < 64) klo |= 1ull << i; else khi |= 1ull << (i - 64);
        const int l = i & 63;
        unsigned long long rlo, rhi, ylo, yhi;
        if (i < 64) {
            rlo = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(slo[0] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)slo[0], l);
            rhi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(shi[0] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)shi[0], l);
            ylo = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(zlo[0] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)zlo[0], l);
            yhi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(zhi[0] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)zhi[0], l);
        } else {
            rlo = 0ull; ylo = 0ull;
            rhi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(shi[1] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)shi[1], l);
            yhi = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(zhi[1] >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)zhi[1], l);
        }
        if ((ylo & ~dlo) | (yhi & ~dhi)) bad = 1;      // an evaluated pair with zero union
        dlo |= rlo; dhi |= rhi;
    }
    // kept candidates in score order -> their row positions
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int q = lane + 64 * h;
        const bool kept = h == 0 ? ((klo >> lane) & 1ull) : ((khi >> lane) & 1ull);
        if (q < M && kept) {
            const int slot = h == 0 ? __popcll(klo & ((1ull << lane) - 1ull))
                                    : __popcll(klo) + __popcll(khi & ((1ull << lane) - 1ull));
            prm.keep[obase + slot] = row[h];
        }
    }
    if (lane == 0) {
        prm.det_cnt[p] = M;
        prm.keep_cnt[p] = __popcll(klo) + __popcll(khi);
        if (bad) atomicOr(prm.status, kStDivZero);
    }
}

}  // namespace vdet
