// track_kernels.hpp -- device-resident greedy tubelet generation for a whole score volume
// (the array form of greedily_track_from_raw_dets, vdet/track.py:189-252, for every class at once).
//
// State per (frame, class): the list of still-kept detections in descending score order (initially
// the full argsort produced by sort_kernel).  The reference only ever REMOVES detections, so the next
// anchor is the best live list head over the frames, and each track_det_nms call
// (utils/nms.pyx:128-189) rewrites one list: round 1 drops entries overlapping the track box, round 2
// is a greedy walk over what is left.
//
//   track_pick_kernel      one block per class: next anchor = best live head over the frames, honouring
//                          the reference's monotone cursor (cur_top_det_id never goes back) and its
//                          stop rule (score < opts.thres).  On regular frames it also IS track_det_nms,
//                          lazily: see LazyLists (kept prefix evaluated on demand, persistent heads).
//   track_link_kernel      one block per (class, direction): the built-in IoU-linking tracker plug-in
//                          (stand-in for the external MATLAB trackers): from the anchor, frame by
//                          frame, the proposal with the highest IoU with the current box (>= link_thres).
//   track_suppress_kernel  eager track_det_nms, one wave per (frame, class) crossed by the new track:
//                          irregular frames only (and everywhere under VDET_NO_LAZY=1).
//   rescore_*_kernel       spatial max-pool / completion / temporal max-pool of the finished tubelets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"
#include "tubelet_kernels.hpp"

namespace vdet {

struct TrackState {          // per class
    int32_t active;          // 0 once the class stopped (low confidence / nothing left / max_tracks)
    int32_t ntracks;
    uint32_t last_key;       // global-order position of the last anchor: (key desc, flat index asc)
    int32_t last_flat;       // -1: none yet
    int32_t anchor_frame;    // 0-based frame of the current anchor
    int32_t anchor_box;
    float anchor_score;
    int32_t resolved;        // the current anchor's tubelet was copied from the materialised warm chains: no link launch needed
};

// entry e = (key, flat) is at or before the cursor (ka, fa) in the global order
__device__ __forceinline__ bool at_or_before(uint32_t k, int flat, uint32_t ka, int fa)
{
    return fa >= 0 && (k > ka || (k == ka && flat <= fa));
}

// ------------------------------------------------------------------------------------------------
// anchor: for every frame the first list entry after the cursor; best over frames by
// (score desc, flat index asc) == the stable descending sort of vdet/track.py:200.
// keys: [F*C, B] sortable keys (transpose_keys_kernel);  lists: [F*C, B] u16;  cnt: [F*C].
// ------------------------------------------------------------------------------------------------
// Lazy lists (regular frames: finite boxes with positive areas, so no union can be zero and skipping
// the evaluation of a pair can not hide a ZeroDivisionError).  The only thing ever read from a
// (frame, class) list is its best live entry, at most once per track -- so track_det_nms
// (utils/nms.pyx:128-189) is never materialised for these lists:
//   * the first track that crosses the list (t1) defines round 1: entries overlapping its box are
//     not part of the list any more;
//   * round 2 (vid_nms of the rest) is a greedy walk in list order, evaluated only as far as needed:
//     list[0 .. nkp) holds the kept prefix found so far (in place: nkp <= pos), list[pos .. n) is
//     the unexamined tail; an entry is kept iff it does not overlap t1's box and no earlier kept
//     entry suppresses it (pair_pred on the boxes == the suppression-graph edge);
//   * later tracks only remove kept entries (the survivors of a vid_nms are an independent set, a
//     second vid_nms keeps them all), and a removed entry still suppresses what it suppressed:
//     the persistent head skips kept entries that overlap any track box of this class on this frame.
struct LazyLists {
    const float4 *boxes;            // [F*B]
    const float *tracks;            // [C, max_tracks, F, 5]
    const uint32_t *group_flags;    // per frame, or null (= never lazy)
    int32_t *t1;                    // [F*C] 0: no track crossed the list yet, else first track + 1
    int32_t *head, *nkp, *pos;      // [F*C]
    float t32;
};

__device__ __forceinline__ bool lazy_dead(const LazyLists &lz, int f, int e, int F, int B, int c, int max_tracks, int nt)
{
    const float4 bd = lz.boxes[(int64_t)f * B + e];
    const float darea = box_area(bd);
    for (int t = 0; t < nt; ++t) {
        const float *row = lz.tracks + (((int64_t)c * max_tracks + t) * F + f) * 5;
        const float tx1 = row[0];
        if (tx1 != tx1) continue;
        const float4 tb = make_float4(tx1, row[1], row[2], row[3]);
        if (pair_pred(bd, darea, tb, box_area(tb), lz.t32) & 1u) return true;      // the DET is the "i" box
    }
    return false;
}

// examine list[pos]: true if it joins the kept prefix
__device__ __forceinline__ bool lazy_examine(const LazyLists &lz, uint16_t *l, int f, int B, float4 t1b, float t1area,
                                             int &np, int &ps)
{
    const int e = l[ps++];
    const float4 bd = lz.boxes[(int64_t)f * B + e];
    const float darea = box_area(bd);
    if (pair_pred(bd, darea, t1b, t1area, lz.t32) & 1u) return false;              // round 1
    for (int i = 0; i < np; ++i) {
        const float4 bk = lz.boxes[(int64_t)f * B + l[i]];
        if (pair_pred(bk, box_area(bk), bd, darea, lz.t32) & 1u) return false;     // round 2: kept box "i", candidate "j"
    }
    l[np++] = (uint16_t)e;
    return true;
}

struct ResolveArgs {           // the materialised warm chains (null chains: none)
    const int32_t *warm;      // [C, m] flat anchor or -1
    int m;
    const float *chains;      // [C, m, F, 5]
    const int32_t *chain_nodes;
    float *tracks;
    int32_t *nodes;
};

// (block-uniform control flow: every early return below is taken by the whole block)
__device__ __forceinline__ void track_pick_body(const int c, const uint32_t *__restrict__ keys, uint16_t *lists,
                                                const int32_t *__restrict__ cnt, int F, int B, int C,
                                                const float *__restrict__ scores, double thres, int max_tracks,
                                                TrackState *__restrict__ st, float *__restrict__ anchors,
                                                const LazyLists &lz, const ResolveArgs &rv)
{
    __shared__ uint32_t sk[256];
    __shared__ int sf[256];
    __shared__ int sslot;
    const int tid = threadIdx.x;
    TrackState s = st[c];
    if (!s.active) return;
    uint32_t bk = 0;
    int bflat = -1;
    for (int f = tid; f < F; f += 256) {
        const int p = f * C + c;
        const int n = cnt[p];
        uint16_t *l = lists + (int64_t)p * B;
        const uint32_t *kk = keys + (int64_t)p * B;
        int t1 = 0;
        if (lz.group_flags && (lz.group_flags[f] & kFlagRegular)) {
            t1 = lz.t1[p];
            if (t1 == 0 && s.ntracks > 0) {    // did the newest track cross this list?  (older ones were checked before)
                const float r0 = lz.tracks[(((int64_t)c * max_tracks + (s.ntracks - 1)) * F + f) * 5];
                if (r0 == r0) { t1 = s.ntracks; lz.t1[p] = t1; }
            }
        }
        if (t1 == 0) {
            // full list (or an eagerly compacted one): skip entries at/before the cursor (only the
            // previous anchor itself can be there)
            int q = 0;
            for (;;) {
                if (q >= n) break;
                if (!at_or_before(kk[l[q]], f * B + l[q], s.last_key, s.last_flat)) break;
                ++q;
            }
            if (q >= n) continue;
            // ties inside the frame are listed by DESCENDING index; the global rule wants the lowest
            const uint32_t k0 = kk[l[q]];
            int best = l[q];
            for (int r = q + 1; r < n; ++r) {
                if (kk[l[r]] != k0) break;
                if (!at_or_before(k0, f * B + l[r], s.last_key, s.last_flat)) best = l[r];
            }
            const int flat = f * B + best;
            if (bflat < 0 || k0 > bk || (k0 == bk && flat < bflat)) { bk = k0; bflat = flat; }
            continue;
        }
        const float *row1 = lz.tracks + (((int64_t)c * max_tracks + (t1 - 1)) * F + f) * 5;
        const float4 t1b = make_float4(row1[0], row1[1], row1[2], row1[3]);
        const float t1area = box_area(t1b);
        int h = lz.head[p], np = lz.nkp[p], ps = lz.pos[p];
        int found = -1;
        for (;;) {
            if (h >= np) {                      // extend the kept prefix by one entry
                bool got = false;
                while (ps < n && !got) {
                    got = lazy_examine(lz, l, f, B, t1b, t1area, np, ps);
                }
                if (!got) break;                // list exhausted
            }
            const int e = l[h];
            if (at_or_before(kk[e], f * B + e, s.last_key, s.last_flat) ||
                lazy_dead(lz, f, e, F, B, c, max_tracks, s.ntracks)) { ++h; continue; }   // gone for good
            found = e;
            break;
        }
        if (found >= 0) {
            // ties: later kept entries with the same key (descending index), the lowest live one wins
            const uint32_t k0 = kk[found];
            int best = found;
            int r = h + 1;
            for (;;) {
                if (r >= np) {
                    if (ps >= n) break;
                    if (kk[l[ps]] != k0) break;
                    lazy_examine(lz, l, f, B, t1b, t1area, np, ps);
                    continue;
                }
                const int e = l[r];
                if (kk[e] != k0) break;
                if (!at_or_before(k0, f * B + e, s.last_key, s.last_flat) &&
                    !lazy_dead(lz, f, e, F, B, c, max_tracks, s.ntracks)) best = e;
                ++r;
            }
            const int flat = f * B + best;
            if (bflat < 0 || k0 > bk || (k0 == bk && flat < bflat)) { bk = k0; bflat = flat; }
        }
        lz.head[p] = h; lz.nkp[p] = np; lz.pos[p] = ps;
    }
    sk[tid] = bk;
    sf[tid] = bflat;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) {
            const uint32_t k2 = sk[tid + d];
            const int f2 = sf[tid + d];
            if (f2 >= 0 && (sf[tid] < 0 || k2 > sk[tid] || (k2 == sk[tid] && f2 < sf[tid]))) { sk[tid] = k2; sf[tid] = f2; }
        }
        __syncthreads();
    }
    if (tid == 0) {     // (no early return: every thread meets the barriers below)
        if (sf[0] < 0 || s.ntracks >= max_tracks) {                                 // "while np.any(keep) and len(tracks) < max_tracks"
            st[c].active = 0;
        } else {
            const int f = sf[0] / B, b = sf[0] - f * B;
            const float sc = scores[((int64_t)f * B + b) * C + c];
            st[c].last_key = sk[0];
            st[c].last_flat = sf[0];
            if ((double)sc < thres) {                                               // vdet/track.py:218 (f32 score vs python float)
                st[c].active = 0;
            } else {
                st[c].anchor_frame = f;
                st[c].anchor_box = b;
                st[c].anchor_score = sc;
                st[c].resolved = 0;
                float *a = anchors + ((int64_t)c * max_tracks + s.ntracks) * 3;
                a[0] = (float)(f + 1);      // 1-based frame id
                a[1] = (float)b;
                a[2] = sc;
            }
        }
    }
    // The anchor is (almost always) one of the class's warm anchors, whose tubelet the materialise launch
    // (track_link_memo_kernel MODE 2) has already written: copy it into the class's next track slot.  A tubelet depends on
    // its anchor only -- next(f, j, dir) knows neither class nor track -- so the copy IS what the link would write.  An
    // anchor that was not predicted leaves resolved = 0 and the link kernel runs for this class.
    if (!rv.chains) return;
    if (tid == 0) sslot = -1;
    __threadfence_block();
    __syncthreads();
    const TrackState s2 = st[c];             // (written by thread 0 above: visible after the fence + barrier)
    if (!s2.active) return;
    if (tid < rv.m && rv.warm[c * rv.m + tid] == s2.anchor_frame * B + s2.anchor_box) sslot = tid;
    __syncthreads();
    const int slot = sslot;
    if (slot < 0) return;
    const float *src = rv.chains + ((int64_t)c * rv.m + slot) * F * 5;
    float *dst = rv.tracks + ((int64_t)c * max_tracks + s2.ntracks) * F * 5;
    for (int i = tid; i < F * 5; i += 256) dst[i] = src[i];
    if (rv.nodes) {
        const int32_t *ns = rv.chain_nodes + ((int64_t)c * rv.m + slot) * F;
        int32_t *nd = rv.nodes + ((int64_t)c * max_tracks + s2.ntracks) * F;
        for (int i = tid; i < F; i += 256) nd[i] = ns[i];
    }
    if (tid == 0) st[c].resolved = 1;
}

// IoU of the current track box (as the "i" box) with a proposal (as "j"), utils/nms.pyx arithmetic
__device__ __forceinline__ float link_iou(float4 cur, float carea, float4 b)
{
    const float xx1 = ref_max(cur.x, b.x), yy1 = ref_max(cur.y, b.y);
    const float xx2 = ref_min(cur.z, b.z), yy2 = ref_min(cur.w, b.w);
    const float w = ref_max(0.0f, (xx2 - xx1) + 1.0f), h = ref_max(0.0f, (yy2 - yy1) + 1.0f);
    const float inter = w * h;
    return inter / ((carea + box_area(b)) - inter);
}

__device__ __forceinline__ float4 trunc4(float4 b)   // int(cor) of utils/protocol.py:406
{
    return make_float4(truncf(b.x), truncf(b.y), truncf(b.z), truncf(b.w));
}

// ------------------------------------------------------------------------------------------------
// built-in tracker: tracks [C, max_tracks, F, 5] rows (x1,y1,x2,y2,score), NaN where the track has
// no box.  The anchor row is the int-truncated anchor box with score 1; a neighbour frame gets the
// (int-truncated) proposal with the highest f32 IoU with the current box, first index on ties,
// while that IoU >= link_t32; at most `reach` frames to each side.
// ------------------------------------------------------------------------------------------------
// grid = (C, 2): blockIdx.y = 0 links forward (and writes the anchor row), 1 backward; block = LT.
// One barrier per frame: wave-level argmax by shuffles, LT/64 partial results in a parity-double-
// buffered LDS slot, every thread finishes the reduction redundantly.  The step is a serial chain
// executed by every wave, so a SMALL block wins: with 1024 threads the replicated serial part cost
// ~7 us per frame (measured), the x-window leaves only ~2 000 candidates per frame anyway.
// LT = threads per chain (template parameter)

// One batch of the x-window scan of track_link_kernel: WB boxes per thread starting at rank rb0.
// All loads are issued before the first use (a load inside the ballot-branching loop body is waited
// for immediately: one full memory latency per box, ~7 us per frame measured), and the boxes'
// original indices (tie rule) come with them: fetching the index inside the passing branch was a
// second, serialised memory round trip per passing iteration (16.4 -> 13.6 ms per video).
template <int WB, int LT>
__device__ __forceinline__ void link_scan(const float4 *__restrict__ xb, const uint16_t *__restrict__ xo, int rb0, int r1,
                                          int B, int tid, float4 cur, float carea, float link_t32, float t32e, float &bv,
                                          int &bi, float4 &bb)
{
    float4 xs[WB];
    uint16_t xi[WB];
#pragma unroll
    for (int i = 0; i < WB; ++i) xs[i] = xb[min(rb0 + i * LT + tid, B - 1)];
#pragma unroll
    for (int i = 0; i < WB; ++i) xi[i] = xo[min(rb0 + i * LT + tid, B - 1)];
    // one ballot-guarded candidate at a time: each starts as soon as ITS load has landed (computing all
    // WB predicates branch-free first was measured slower: 12.3 vs 11.5 ms per video)
#pragma unroll
    for (int i = 0; i < WB; ++i) {
        const int r = rb0 + i * LT + tid;
        const bool inb = r < r1;
        bool border;
        const bool pass = pred_regular(cur, carea, xs[i], box_area(xs[i]), link_t32, t32e, border);
        if (__ballot((pass || border) && inb)) {
            const float v = link_iou(cur, carea, xs[i]);
            const int b = (int)xi[i];
            if (inb && v >= link_t32 && (v > bv || (v == bv && b < bi))) { bv = v; bi = b; bb = xs[i]; }
        }
    }
}
// The same batch over a kFlagU16 frame's compact index (FrameIndex::xbox16): WP PAIRS of neighbouring ranks per thread,
// a pair's boxes in one 16-byte and its indices in one 4-byte load.  pb0 = first pair, ranks >= r1 are masked; ranks below
// the window's start (the first pair's even half) are simply evaluated: the window only has to be a superset.
__device__ __forceinline__ float4 unpack_box16(uint32_t lo, uint32_t hi)
{
    return make_float4((float)(lo & 0xFFFFu), (float)(lo >> 16), (float)(hi & 0xFFFFu), (float)(hi >> 16));
}

template <int WP, int LT>
__device__ __forceinline__ void link_scan16(const uint4 *__restrict__ xb2, const uint32_t *__restrict__ xo2, int pb0, int r1,
                                            int B, int tid, float4 cur, float carea, float link_t32, float t32e, float &bv,
                                            int &bi, float4 &bb)
{
    uint4 xs[WP];
    uint32_t xi[WP];
    const int last = (B - 1) >> 1;
#pragma unroll
    for (int i = 0; i < WP; ++i) xs[i] = xb2[min(pb0 + i * LT + tid, last)];
#pragma unroll
    for (int i = 0; i < WP; ++i) xi[i] = xo2[min(pb0 + i * LT + tid, last)];
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = 2 * (pb0 + i * LT + tid);
        const bool in0 = r < r1, in1 = r + 1 < r1;
        const float4 b0 = unpack_box16(xs[i].x, xs[i].y), b1 = unpack_box16(xs[i].z, xs[i].w);
        bool bd0, bd1;
        const bool p0 = pred_regular(cur, carea, b0, box_area(b0), link_t32, t32e, bd0);
        const bool p1 = pred_regular(cur, carea, b1, box_area(b1), link_t32, t32e, bd1);
        if (__ballot(((p0 | bd0) & in0) | ((p1 | bd1) & in1))) {
            const float v0 = link_iou(cur, carea, b0), v1 = link_iou(cur, carea, b1);
            const int i0 = (int)(xi[i] & 0xFFFFu), i1 = (int)(xi[i] >> 16);
            if (in0 && v0 >= link_t32 && (v0 > bv || (v0 == bv && i0 < bi))) { bv = v0; bi = i0; bb = b0; }
            if (in1 && v1 >= link_t32 && (v1 > bv || (v1 == bv && i1 < bi))) { bv = v1; bi = i1; bb = b1; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same tracker with a LINK MEMO.  One link step is a pure function of the node it starts from:
//     next(f, j, dir) = argmax_k IoU(trunc(box[f][j]), box[f + dir][k])   (>= link_thres, first index on ties)
// -- it depends neither on the class nor on the track, and every chain (2 000 per config-2 video: 200
// classes x 10 tracks, ~300 steps each) walks the same functional graph.  Chains that meet stay
// together, so most steps of most chains were already computed by another chain: memo[dir][f][j]
// holds (valid | index + 1 | IoU bits) of every step taken so far in this video, a step first looks
// there (ONE 8-byte load instead of a scan of the frame's ~2 400-box IoU window) and only scans on a
// miss.  Values are deterministic, so concurrent chains racing on an entry write the same 8 bytes
// (relaxed agent-scope 64-bit atomics: never torn; a stale "unknown" only costs a redundant scan).
// The memo lives for one vdet_track_volume call (cleared at its start).
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long kMemoValid = 1ull << 63;

__device__ __forceinline__ unsigned long long memo_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// WARM = true: the chain starts at warm[blockIdx.x] (a flat detection index, < 0: nothing to do) and only
// fills the memo (no track rows): see track_warm_anchors_kernel.
//
// Structure: KNOWN steps are walked by wave 0 alone, one 8-byte load per step and no barrier (the row of the
// previous step is written while the next memo word is in flight); an UNKNOWN step is scanned by the whole
// block like track_link_kernel and its result published.
// MODE 0: the tracking loop's link of one class (anchor from its TrackState, rows into its next track slot)
// MODE 1: warm-up (anchor = warm[blockIdx.x], nothing written but the memo, a chain stops at its first known step)
// MODE 2: materialise (anchor = warm[blockIdx.x], rows into chain slot blockIdx.x of `tracks` / `nodes`): run once after
//         the warm-up, when every step of these chains is known -- the tracking loop then COPIES a predicted anchor's
//         tubelet (the tail of track_pick_kernel) instead of walking its ~300 dependent steps again, track after track
// chain = the class (MODE 0) or the warm-chain slot (MODE 1 / 2); dir = +1 / -1
// (helpers of the window scans below)
#define LINK_DPP_STEP(CTRL, ROWMASK) { \
        const float v2 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-1.0f), __float_as_int(bv), CTRL, ROWMASK, 0xf, false)); \
        const int i2 = __builtin_amdgcn_update_dpp(-1, bi, CTRL, ROWMASK, 0xf, false); \
        if (i2 >= 0 && (bi < 0 || v2 > bv || (v2 == bv && i2 < bi))) { bv = v2; bi = i2; } }
#define LSCAN(W) link_scan<W, LT>(xb, xo, rb0, r1, B, tid, cur, carea, link_t32, t32e, bv, bi, bb);
template <int LT, int MODE, int MAXB>
__device__ __forceinline__ void track_link_memo_body(const int chain, const int dir, const float4 *__restrict__ boxes, int F, int B, int max_tracks,
                                                     float link_t32, int reach, const TrackState *__restrict__ st,
                                                     float *__restrict__ tracks,
                                                     const uint32_t *__restrict__ group_flags,
                                                     const FrameIndex &ix, double link_thres,
                                                     unsigned long long *memo, unsigned int *__restrict__ stats,
                                                     const int32_t *__restrict__ warm, int32_t *__restrict__ nodes)
{
    __shared__ float sv[2][LT / 64];
    __shared__ int si[2][LT / 64];
    __shared__ float4 sb[2][LT / 64];
    __shared__ uint32_t scum[2][261];          // bucket table + (xmin, scale, wmax) + group flags of frame f in slot f & 1
    __shared__ int sstate[4];                  // node, fprev, step, done -- where wave 0's run of known steps ended
    __shared__ int speek[2];                   // (per step parity) the scanned step turned out to be in the memo
    static_assert(LT % 64 == 0 && LT >= 64 && LT <= 1024, "whole waves");
    constexpr int NPF = (261 + LT - 1) / LT;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int anchor_frame, anchor_box;
    float *trk = nullptr;
    constexpr bool WARM = MODE == 1;
    if (MODE != 0) {
        const int flat = warm[chain];
        if (flat < 0) return;
        anchor_frame = flat / B;
        anchor_box = flat - anchor_frame * B;
    } else {
        const int c = chain;
        const TrackState s = st[c];
        if (!s.active || s.resolved) return;
        anchor_frame = s.anchor_frame;
        anchor_box = s.anchor_box;
    }
    if (MODE != 1) {
        if (MODE == 0) {
            const TrackState s = st[chain];
            trk = tracks + ((int64_t)chain * max_tracks + s.ntracks) * F * 5;
            if (nodes) nodes += ((int64_t)chain * max_tracks + s.ntracks) * F;      // which proposal each row of the track is
        } else {
            trk = tracks + (int64_t)chain * F * 5;
            if (nodes) nodes += (int64_t)chain * F;
        }
        const float qnan = __uint_as_float(0x7FC00000u);
        if (dir > 0) { for (int i = anchor_frame * 5 + tid; i < F * 5; i += LT) trk[i] = qnan; }
        else { for (int i = tid; i < anchor_frame * 5; i += LT) trk[i] = qnan; }
        __syncthreads();
        if (dir > 0 && tid == 0) {
            const float4 anchor = trunc4(boxes[(int64_t)anchor_frame * B + anchor_box]);
            float *r = trk + (int64_t)anchor_frame * 5;
            r[0] = anchor.x; r[1] = anchor.y; r[2] = anchor.z; r[3] = anchor.w; r[4] = 1.0f;
            if (nodes) nodes[anchor_frame] = anchor_box;
        }
    }
    unsigned long long *mm = memo + (int64_t)(dir > 0 ? 0 : 1) * F * B;
    const bool use_ix = ix.xbox != nullptr;
    const float omt = (float)(1.0 - link_thres) * 1.002f + 1.0e-6f;
    const float inv_t = (float)(1.002 / fmax(link_thres, 1.0e-6));
    const float t32e = link_t32 * 4.76837158203125e-7f;
    int node = anchor_box, fprev = anchor_frame, step = 1;
    int tbl[2] = {-1, -1};           // which frame's table sits in scum[0] / scum[1]
    unsigned int nhit = 0, nmiss = 0;
    // A step costs dependent memory round trips, not work.  After a step that had to be scanned, the next one is scanned
    // WITHOUT asking the memo first (in the warm-up nearly every step is unknown), starting from the winner's box that the
    // scan already holds; the memo word of the step is fetched alongside the window and only decides, afterwards, whether
    // the chain goes back to following known steps.  Scanning a known step is redundant, never wrong: the scan and the memo
    // hold the same next(f, j, dir).  One round trip per step instead of three (memo, current box, window).
    bool scanning = false;
    float4 cur_next = make_float4(0.f, 0.f, 0.f, 0.f);
    for (;;) {
        // ---- wave 0: the run of known steps from (fprev, node)
        if (!scanning) {
        if (w == 0) {
            int done = 0;
            int pend_f = -1;                     // row not yet written: frame, IoU bits, box (load in flight)
            uint32_t pend_s = 0u;
            float4 pend_b = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned long long m = 0ull;
            bool have = false;
            for (;;) {
                const int f = anchor_frame + dir * step;
                if (step > reach || f < 0 || f >= F) { done = 1; break; }
                if (!have) m = memo_load(&mm[(int64_t)fprev * B + node]);
                m = __builtin_amdgcn_readfirstlane((uint32_t)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32);
                have = false;
                if (!(m & kMemoValid)) break;                              // unknown: the block scans this step
                // warm-up with unlimited reach: a known step has an owner -- the chain that scanned it went on from
                // there and runs (or hands over, like this one) to the end of the video -- so there is nothing left to do
                if (WARM && reach >= F) { ++nhit; done = 1; break; }
                const int bidx = (int)((m >> 32) & 0x7FFFFFFFull) - 1;
                if (bidx < 0) { done = 1; break; }                         // known: the chain ends here
                float4 nb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!WARM) nb = boxes[(int64_t)f * B + bidx];              // (issued before the next memo word ...
                const uint32_t sbits = (uint32_t)m;
                node = bidx; fprev = f; ++step; ++nhit;
                const int f2 = anchor_frame + dir * step;
                if (step <= reach && f2 >= 0 && f2 < F) { m = memo_load(&mm[(int64_t)fprev * B + node]); have = true; }
                if (!WARM) {                                               //  ... and written one step later)
                    if (pend_f >= 0 && lane == 0) {
                        const float4 t = trunc4(pend_b);
                        float *r = trk + (int64_t)pend_f * 5;
                        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; r[4] = __uint_as_float(pend_s);
                    }
                    if (nodes && lane == 0) nodes[f] = bidx;
                    pend_f = f; pend_s = sbits; pend_b = nb;
                }
            }
            if (!WARM && pend_f >= 0 && lane == 0) {
                const float4 t = trunc4(pend_b);
                float *r = trk + (int64_t)pend_f * 5;
                r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; r[4] = __uint_as_float(pend_s);
            }
            if (lane == 0) { sstate[0] = node; sstate[1] = fprev; sstate[2] = step; sstate[3] = done; }
        }
        __syncthreads();
        node = sstate[0]; fprev = sstate[1]; step = sstate[2];
        if (sstate[3]) break;
        }
        // ---- unknown step: scan frame f with the whole block (as track_link_kernel)
        ++nmiss;
        const int f = anchor_frame + dir * step;
        const int par = step & 1;
        unsigned long long peek = 0ull;          // (thread 0) was this step in the memo after all?
        if (scanning && tid == 0) peek = memo_load(&mm[(int64_t)fprev * B + node]);
        float4 cur = scanning ? cur_next : trunc4(boxes[(int64_t)fprev * B + node]);
        const float carea = box_area(cur);
        const float4 *fb = boxes + (int64_t)f * B;
        float bv = -1.0f;
        int bi = -1;
        float4 bb = cur;
        // the frame's flags travel with its bucket table (prefetched one step ahead): a global load here would be one more
        // dependent round trip in front of the window loads
        const int slot = f & 1;
        uint32_t fflags = 0u;
        if (use_ix && group_flags) {
            if (tbl[slot] != f) {    // (after a run of known steps) this frame's table was not prefetched
                for (int i = tid; i < 261; i += LT)
                    scum[slot][i] = i < 257 ? ix.cum[(int64_t)f * 257 + i] : (i < 260 ? __float_as_uint(ix.info[f * 4 + (i - 257)]) : group_flags[f]);
                __syncthreads();
                tbl[slot] = f;
            }
            fflags = scum[slot][260];
        } else if (group_flags) {
            fflags = group_flags[f];
        }
        const bool fast = (fflags & kFlagRegular) && link_t32 > 1e-30f && carea > 0.0f && carea < __uint_as_float(0x7F800000u);
        if (fast && use_ix) {
            const int f2 = min(max(f + dir, 0), F - 1);
            uint32_t pf[NPF];        // the next frame's table: loads issued here, stored to the other slot after the scan
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                const int i = tid + j * LT;
                pf[j] = i < 257 ? ix.cum[(int64_t)f2 * 257 + min(i, 256)]
                                : (i < 260 ? __float_as_uint(ix.info[f2 * 4 + (i - 257)]) : group_flags[f2]);
            }
            int r0, r1;
            {
                const float xmin = __uint_as_float(scum[slot][257]), scale = __uint_as_float(scum[slot][258]);
                const float wmax = __uint_as_float(scum[slot][259]);
                const float wc = (cur.z - cur.x) + 1.0f;
                const float lo = cur.x - omt * fminf(wmax, wc * inv_t) - 2.0f;
                const float hi = cur.x + omt * wc + 2.0f;
                r0 = (int)scum[slot][xbucket(fmaxf(lo, -3.0e38f), xmin, scale)];
                r1 = (int)scum[slot][xbucket(fminf(hi, 3.0e38f), xmin, scale) + 1];
            }
            if (ix.xbox16 && (fflags & kFlagU16)) {
                // integer pixel coordinates: the compact index, two candidates per load
                const int64_t p16 = pair_pos(ix.bias16 + (int64_t)f * (B + 1));
                const uint4 *xb2 = reinterpret_cast<const uint4 *>(ix.xbox16 + p16);
                const uint32_t *xo2 = reinterpret_cast<const uint32_t *>(ix.xord16 + p16);
                int pb0 = r0 >> 1;
                const int pe = (r1 + 1) >> 1;
#define LSCAN16(W) link_scan16<W, LT>(xb2, xo2, pb0, r1, B, tid, cur, carea, link_t32, t32e, bv, bi, bb);
                while (pe - pb0 > MAXB * LT) { LSCAN16(MAXB) pb0 += MAXB * LT; }
                const int np = (pe - pb0 + LT - 1) / LT;
                if (MAXB >= 16 && np > 12) { LSCAN16(16) } else if (MAXB >= 16 && np > 8) { LSCAN16(12) } else if (np > 4) { LSCAN16(8) }
                else if (np > 2) { LSCAN16(4) } else if (np > 0) { LSCAN16(2) }
#undef LSCAN16
            } else {
            const float4 *xb = ix.xbox + (int64_t)f * B;
            const uint16_t *xo = ix.xord + (int64_t)f * B;
            // batches of at most MAXB boxes per thread: 16 = one memory round trip for a typical window at the price of
            // ~180 registers (2 waves / SIMD); 8 = two round trips at twice the occupancy -- what the chip-filling warm-up wants
            int rb0 = r0;
            while (r1 - rb0 > MAXB * LT) { LSCAN(MAXB) rb0 += MAXB * LT; }
            const int nb = (r1 - rb0 + LT - 1) / LT;
            if (MAXB >= 16 && nb > 12) { LSCAN(16) } else if (MAXB >= 16 && nb > 8) { LSCAN(12) } else if (nb > 4) { LSCAN(8) } else if (nb > 0) { LSCAN(4) }
            }
            if (f2 != f) {
#pragma unroll
                for (int j = 0; j < NPF; ++j)
                    if (tid + j * LT < 261) scum[slot ^ 1][tid + j * LT] = pf[j];
                tbl[slot ^ 1] = f2;
            }
        } else {
            // irregular frame / no index / degenerate current box: the plain arg-max (NaN never wins, lowest index on ties)
            for (int b = tid; b < B; b += LT) {
                const float4 x = fb[b];
                const float v = link_iou(cur, carea, x);
                if (v > bv) { bv = v; bi = b; bb = x; }
            }
        }
        const int my_bi = bi;
        LINK_DPP_STEP(0x111, 0xf) LINK_DPP_STEP(0x112, 0xf) LINK_DPP_STEP(0x114, 0xf) LINK_DPP_STEP(0x118, 0xf)
        LINK_DPP_STEP(0x142, 0xa) LINK_DPP_STEP(0x143, 0xc)
        bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), 63));
        bi = __builtin_amdgcn_readlane(bi, 63);
        if (lane == 0) { sv[par][w] = bv; si[par][w] = bi; }
        if (bi >= 0 && my_bi == bi) sb[par][w] = bb;
        if (tid == 0) speek[par] = (peek & kMemoValid) ? 1 : 0;
        __syncthreads();
        float best = sv[par][0];
        int bidx = si[par][0];
        int bw = 0;
#pragma unroll
        for (int k = 1; k < LT / 64; ++k) {
            const float v2 = sv[par][k];
            const int i2 = si[par][k];
            if (i2 >= 0 && (bidx < 0 || v2 > best || (v2 == best && i2 < bidx))) { best = v2; bidx = i2; bw = k; }
        }
        const bool linked = bidx >= 0 && best >= link_t32;
        if (tid == 0)
            __hip_atomic_store(&mm[(int64_t)fprev * B + node],
                               kMemoValid | ((unsigned long long)(linked ? bidx + 1 : 0) << 32) | (linked ? __float_as_uint(best) : 0u),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!linked) break;
        if (!WARM && tid == 0) {
            const float4 t = trunc4(sb[par][bw]);
            float *r = trk + (int64_t)f * 5;
            r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w; r[4] = best;
            if (nodes) nodes[f] = bidx;
        }
        node = bidx; fprev = f; ++step;
        {
            const int f2 = anchor_frame + dir * step;
            if (step > reach || f2 < 0 || f2 >= F) break;
        }
        if (speek[par]) {           // the step just scanned was known: (warm-up, unlimited reach) its owner goes on from here
            ++nhit;                 // (counted as found AND as scanned)
            if (WARM && reach >= F) break;
            scanning = false;
        } else {
            scanning = true;
            cur_next = trunc4(sb[par][bw]);
        }
    }
    if (stats && tid == 0) { atomicAdd(&stats[WARM ? 2 : 0], nhit); atomicAdd(&stats[WARM ? 3 : 1], nmiss); }
}

template <int LT, int MODE, int MAXB>
__global__ __launch_bounds__(LT, (MODE == 1 && MAXB == 8) ? 5 : 1) void track_link_memo_kernel(const float4 *__restrict__ boxes, int F, int B, int max_tracks,
                                                             float link_t32, int reach, const TrackState *__restrict__ st,
                                                             float *__restrict__ tracks,
                                                             const uint32_t *__restrict__ group_flags,
                                                             const FrameIndex ix, double link_thres,
                                                             unsigned long long *memo, unsigned int *__restrict__ stats,
                                                             const int32_t *__restrict__ warm, int32_t *__restrict__ nodes,
                                                             const int32_t *__restrict__ order = nullptr)
{
    int chain = blockIdx.x, dir = blockIdx.y == 0 ? 1 : -1;
    if (order) {         // (warm-up) longest chains first: warm_order_kernel
        const int e = order[blockIdx.y * gridDim.x + blockIdx.x];
        chain = e >> 1;
        dir = (e & 1) ? -1 : 1;
    }
    track_link_memo_body<LT, MODE, MAXB>(chain, dir, boxes, F, B, max_tracks, link_t32, reach, st, tracks,
                                         group_flags, ix, link_thres, memo, stats, warm, nodes);
}

// The warm-up's (chain, direction) pairs by DESCENDING length (frames from the anchor to the video's end in that direction,
// capped by reach): a chain is a serial sequence of steps, the longest one (~F steps) bounds the kernel from below, and in
// launch order it may start when most of the chip's block slots have already turned over several times.  One block,
// counting sort on 256 length classes; order[i] = chain << 1 | (backward ? 1 : 0).
__global__ __launch_bounds__(1024) void warm_order_kernel(const int32_t *__restrict__ warm, int n, int F, int B, int reach,
                                                          int32_t *__restrict__ order)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t base[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    auto bin_of = [&](int e) {
        const int flat = warm[e >> 1];
        if (flat < 0) return 255;
        const int af = flat / B;
        const int len = min((e & 1) ? af : F - 1 - af, reach);
        return 254 - (int)(((int64_t)len * 254) / max(F, 1));
    };
    for (int e = tid; e < 2 * n; e += 1024) atomicAdd(&hist[bin_of(e)], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 256; ++k) { base[k] = run; run += hist[k]; }
    }
    __syncthreads();
    for (int e = tid; e < 2 * n; e += 1024) order[atomicAdd(&base[bin_of(e)], 1u)] = e;
}

// ------------------------------------------------------------------------------------------------
// The WHOLE link table at once (round 4), for frames of up to a few hundred proposals (VID shape): one thread per node
// (frame, proposal) and direction evaluates next(f, j, dir) -- the same window scan, screen and arg-max rule as an
// unknown step of track_link_memo_body, serially over the node's window (tens of candidates) -- and writes the memo
// word.  ~2 F B windows per video: microseconds of chip-filling work, after which EVERY chain of the video is pointer
// chasing (0.38 us per step) whatever its anchor is -- no anchor prediction, no warm-up launch, no serial window scans
// inside the tracking loop (on coherent videos the warm-up predicted one chain per class and the loop scanned the rest).
// The values are the ones a scan would publish, so the memo-driven kernels are unchanged.
// ------------------------------------------------------------------------------------------------
constexpr int kLinkFillMax = 1024;      // frames of up to this many proposals get their whole link table up front

// The same table, one BLOCK per (frame f, direction) with frame f + dir's x-sorted index -- boxes, areas, box numbers, bucket
// table -- staged in LDS once (<= 24 KB at B = 1 024): a node's window scan then reads LDS instead of gathering 16-byte
// records from global memory four at a time (link_fill_node waits for every such turn: the kernel was latency-bound, 3.4 ms
// for the 64-video batch).  One thread per node, same window, screen, arg-max and tie rule, same memo words.
// Dynamic LDS: [B] float4 boxes, [B] float areas, [260] bucket table + (xmin, scale, wmax), [B] u16 box numbers.
inline size_t link_fill_lds_bytes(int B) { return (size_t)B * 22 + 260 * 4 + 16; }

__device__ __forceinline__ void link_fill_frame(const int f, const int dir, const float4 *__restrict__ boxes, int F, int B, float link_t32,
                                                const uint32_t *__restrict__ group_flags, const FrameIndex &ix, double link_thres,
                                                unsigned long long *memo, unsigned char *smem)
{
    const int f2 = f + dir;
    if (f2 < 0 || f2 >= F) return;                    // (a chain stops at the video's border before it asks)
    float4 *sxb = reinterpret_cast<float4 *>(smem);
    float *sar = reinterpret_cast<float *>(sxb + B);
    uint32_t *scum = reinterpret_cast<uint32_t *>(sar + B);
    uint16_t *sxo = reinterpret_cast<uint16_t *>(scum + 260);
    const int tid = threadIdx.x, nt = blockDim.x;
    const uint32_t fflags = group_flags ? group_flags[f2] : 0u;
    const bool fastf = ix.xbox && (fflags & kFlagRegular) && link_t32 > 1e-30f;       // (block-uniform)
    if (fastf) {
        for (int i = tid; i < B; i += nt) {
            const float4 x = ix.xbox[(int64_t)f2 * B + i];
            sxb[i] = x; sar[i] = box_area(x); sxo[i] = ix.xord[(int64_t)f2 * B + i];
        }
        for (int i = tid; i < 260; i += nt)
            scum[i] = i < 257 ? ix.cum[(int64_t)f2 * 257 + i] : __float_as_uint(ix.info[f2 * 4 + (i - 257)]);
    }
    __syncthreads();
    const float omt = (float)(1.0 - link_thres) * 1.002f + 1.0e-6f;
    const float inv_t = (float)(1.002 / fmax(link_thres, 1.0e-6));
    const float t32e = link_t32 * 4.76837158203125e-7f;
    unsigned long long *mm = memo + (int64_t)(dir > 0 ? 0 : 1) * F * B;
    for (int j = tid; j < B; j += nt) {
        const float4 cur = trunc4(boxes[(int64_t)f * B + j]);
        const float carea = box_area(cur);
        float bv = -1.0f;
        int bi = -1;
        if (fastf && carea > 0.0f && carea < __uint_as_float(0x7F800000u)) {
            const float xmin = __uint_as_float(scum[257]), scale = __uint_as_float(scum[258]), wmax = __uint_as_float(scum[259]);
            const float wc = (cur.z - cur.x) + 1.0f;
            const float lo = cur.x - omt * fminf(wmax, wc * inv_t) - 2.0f;
            const float hi = cur.x + omt * wc + 2.0f;
            const int r0 = (int)scum[xbucket(fmaxf(lo, -3.0e38f), xmin, scale)];
            const int r1 = (int)scum[xbucket(fminf(hi, 3.0e38f), xmin, scale) + 1];
            for (int r = r0; r < r1; ++r) {
                const float4 x = sxb[r];
                bool border;
                const bool pass = pred_regular(cur, carea, x, sar[r], link_t32, t32e, border);
                if (pass || border) {
                    const float v = link_iou(cur, carea, x);
                    const int b = (int)sxo[r];
                    if (v >= link_t32 && (v > bv || (v == bv && b < bi))) { bv = v; bi = b; }
                }
            }
        } else {
            // irregular frame / no index / degenerate current box: the plain arg-max (NaN never wins, lowest index on ties)
            const float4 *fb = boxes + (int64_t)f2 * B;
            for (int b = 0; b < B; ++b) {
                const float v = link_iou(cur, carea, fb[b]);
                if (v > bv) { bv = v; bi = b; }
            }
        }
        const bool linked = bi >= 0 && bv >= link_t32;
        __hip_atomic_store(&mm[(int64_t)f * B + j],
                           kMemoValid | ((unsigned long long)(linked ? bi + 1 : 0) << 32) | (linked ? __float_as_uint(bv) : 0u),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// grid (F, 2), block = 64 * ceil(B / 64) threads (<= 1 024), dynamic LDS link_fill_lds_bytes(B)
__global__ __launch_bounds__(1024) void link_fill_frame_kernel(const float4 *__restrict__ boxes, int F, int B, float link_t32,
                                                               const uint32_t *__restrict__ group_flags, const FrameIndex ix, double link_thres,
                                                               unsigned long long *memo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fill_smem[];
    link_fill_frame(blockIdx.x, blockIdx.y == 0 ? 1 : -1, boxes, F, B, link_t32, group_flags, ix, link_thres, memo, fill_smem);
}

// ------------------------------------------------------------------------------------------------
// Likely anchors of a class, for warming the link memo: the best `m` detections of the class in the
// global order of vdet/track.py:200 (score descending, flat index ascending) that score >= thres --
// taken from the first two entries of every (frame, class) sorted list.  The tracking loop's real
// anchors are (almost always) among them: a detection only stops being an anchor candidate when an earlier
// track or a better detection of its own frame overlaps it.  Whatever is predicted wrongly costs time,
// never correctness: the memo only ever holds next(f, j, dir), which no prediction can change.
// One block per class; warm[c * m + k] = flat index or -1.
// ------------------------------------------------------------------------------------------------
// Coherent videos (round 4).  When proposals persist from frame to frame (real box protos do) the best detections of a
// class are the SAME object in every frame: the m best candidates above sit on one or two chains, the other tubelets'
// anchors are not predicted and the tracking loop scans their ~300 steps each itself, serially (measured on a coherent
// config-2 video: 650 k of 660 k link steps scanned inside the loop, 4.4 ms instead of 0.56).  So, when a class's
// raw candidates overlap each other ACROSS frames (>= kWarmCoherent of them repeat an earlier candidate's object), the
// slots [m_raw, m) are filled by what the loop will do: candidates in the same global order, but one that overlaps an
// already predicted anchor's box (as if it stood in the same frame: the object barely moves) is skipped -- the real
// track through its frame would have suppressed it (utils/nms.pyx:163-183).  On independent frames the test fails and
// the extra slots stay empty: nothing changes there.
struct WarmExtra {
    const float4 *boxes;      // [F*B] (null: raw candidates only)
    float t32;                // the NMS threshold of the lists
    int m_raw;                // slots filled by raw rank
    int max_distinct;         // stop predicting once this many DISTINCT objects have an anchor (= max_tracks: the loop takes no more)
};
constexpr int kWarmCoherent = 4;
constexpr int kWarmMax = 32;

__device__ __forceinline__ void track_warm_anchors_body(const int c, const uint32_t *__restrict__ keys, uint16_t *lists,
                                                        const int32_t *__restrict__ cnt, int F, int B, int C,
                                                        const float *__restrict__ scores, double thres, int m,
                                                        int32_t *__restrict__ warm, const WarmExtra &wx)
{
    __shared__ uint32_t sk[256];
    __shared__ int sf[256];
    __shared__ float4 pb[kWarmMax];      // boxes of the predicted anchors (distinct objects first)
    __shared__ int spf[kWarmMax];        // ... their flat indices
    __shared__ int snpb, sdup;
    const int tid = threadIdx.x;
    const int m_raw = (wx.boxes && wx.m_raw < m) ? wx.m_raw : m;
    uint32_t lk = 0xFFFFFFFFu;       // cursor: the last candidate taken (key, flat); everything at or before it is out
    int lf = -1;
    int nraw = 0;
    for (int k = 0; k < m_raw; ++k) {
        uint32_t bk = 0;
        int bflat = -1;
        for (int f = tid; f < F; f += 256) {
            const int p = f * C + c;
            const int n = min(cnt[p], 2);
            for (int q = 0; q < n; ++q) {
                const int e = lists[(int64_t)p * B + q];
                const uint32_t kk = keys[(int64_t)p * B + e];
                const int flat = f * B + e;
                if (lf >= 0 && (kk > lk || (kk == lk && flat <= lf))) continue;       // already taken
                if (bflat < 0 || kk > bk || (kk == bk && flat < bflat)) { bk = kk; bflat = flat; }
            }
        }
        sk[tid] = bk; sf[tid] = bflat;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (tid < d) {
                const uint32_t k2 = sk[tid + d];
                const int f2 = sf[tid + d];
                if (f2 >= 0 && (sf[tid] < 0 || k2 > sk[tid] || (k2 == sk[tid] && f2 < sf[tid]))) { sk[tid] = k2; sf[tid] = f2; }
            }
            __syncthreads();
        }
        lk = sk[0]; lf = sf[0];
        __syncthreads();
        if (lf < 0) {                 // nothing left
            if (tid == 0) for (int r = k; r < m; ++r) warm[c * m + r] = -1;
            return;
        }
        const int f = lf / B, b = lf - f * B;
        const bool below = (double)scores[((int64_t)f * B + b) * C + c] < thres;      // the loop stops at the first such anchor
        if (tid == 0) warm[c * m + k] = below ? -1 : lf;
        if (below) {
            if (tid == 0) for (int r = k + 1; r < m; ++r) warm[c * m + r] = -1;
            return;
        }
        if (tid == 0 && k < kWarmMax) spf[k] = lf;
        nraw = k + 1;
    }
    if (m_raw >= m) return;
    // ---- do the raw candidates repeat each other's objects?  distinct ones -> pb[0 .. npb)
    __syncthreads();
    if (tid == 0) {
        int npb = 0, dup = 0;
        for (int k = 0; k < nraw && k < kWarmMax; ++k) {
            const float4 bx = wx.boxes[spf[k]];
            const float ar = box_area(bx);
            bool rep = false;
            for (int j = 0; j < npb && !rep; ++j) rep = (pair_pred(pb[j], box_area(pb[j]), bx, ar, wx.t32) & 1u) != 0u;
            if (rep) ++dup; else pb[npb++] = bx;
        }
        snpb = npb; sdup = dup;
    }
    __syncthreads();
    if (sdup < kWarmCoherent) {       // independent frames: the raw ranks are the anchors (measured: 99.9 % of them)
        if (tid == 0) for (int r = m_raw; r < m; ++r) warm[c * m + r] = -1;
        return;
    }
    // ---- coherent: the remaining slots by the loop's own rule (global order, minus what a predicted anchor's object covers)
    int pos[2] = {0, 0};              // per frame of this thread (F <= 512 here: the host limits the extra slots to such videos)
    for (int k = m_raw; k < m; ++k) {
        const int npb = snpb;
        if (npb >= wx.max_distinct) {     // every tubelet the loop can take has its anchor: more chains would be scanned for nothing
            if (tid == 0) for (int r = k; r < m; ++r) warm[c * m + r] = -1;
            return;
        }
        uint32_t bk = 0;
        int bflat = -1;
        int h = 0;
        for (int f = tid; f < F && h < 2; f += 256, ++h) {
            const int p = f * C + c;
            const int n = min(cnt[p], 64);                       // (prediction only: the head of the list is plenty)
            uint16_t *l = lists + (int64_t)p * B;
            while (pos[h] < n) {
                const int e = l[pos[h]];
                const float4 bx = wx.boxes[(int64_t)f * B + e];
                const float ar = box_area(bx);
                bool covered = false;
                for (int j = 0; j < npb && !covered; ++j) covered = (pair_pred(pb[j], box_area(pb[j]), bx, ar, wx.t32) & 1u) != 0u;
                if (!covered) break;
                ++pos[h];
            }
            if (pos[h] < n) {
                const int e = l[pos[h]];
                const uint32_t kk = keys[(int64_t)p * B + e];
                const int flat = f * B + e;
                if (bflat < 0 || kk > bk || (kk == bk && flat < bflat)) { bk = kk; bflat = flat; }
            }
        }
        sk[tid] = bk; sf[tid] = bflat;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) {
            if (tid < d) {
                const uint32_t k2 = sk[tid + d];
                const int f2 = sf[tid + d];
                if (f2 >= 0 && (sf[tid] < 0 || k2 > sk[tid] || (k2 == sk[tid] && f2 < sf[tid]))) { sk[tid] = k2; sf[tid] = f2; }
            }
            __syncthreads();
        }
        const int pick = sf[0];
        __syncthreads();
        bool stop = pick < 0;
        if (!stop) {
            const int f = pick / B, b = pick - f * B;
            stop = (double)scores[((int64_t)f * B + b) * C + c] < thres;
        }
        if (stop) {
            if (tid == 0) for (int r = k; r < m; ++r) warm[c * m + r] = -1;
            return;
        }
        if (tid == 0) {
            warm[c * m + k] = pick;
            if (snpb < kWarmMax) { pb[snpb] = wx.boxes[pick]; snpb = snpb + 1; }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void track_warm_anchors_kernel(const uint32_t *__restrict__ keys, uint16_t *lists,
                                                                 const int32_t *__restrict__ cnt, int F, int B, int C,
                                                                 const float *__restrict__ scores, double thres, int m,
                                                                 int32_t *__restrict__ warm, const WarmExtra wx)
{
    track_warm_anchors_body(blockIdx.x, keys, lists, cnt, F, B, C, scores, thres, m, warm, wx);
}

// ------------------------------------------------------------------------------------------------
// track_det_nms of the new track against every frame it crosses: one wave per (frame, class).
// ------------------------------------------------------------------------------------------------
struct SuppressParams {
    const float4 *boxes;
    int F, B, C, max_tracks;
    const GroupDesc *groups;       // one group per frame
    const uint2 *row_meta;         // per box: x = offset of its adjacency list, y = its length
    const uint16_t *adj;
    const uint32_t *group_z;
    const uint32_t *group_flags;   // kFlagRegular per frame, or null
    FrameIndex ix;                 // x-sorted proposal index (xbox == null: none)
    double thres;                  // the NMS threshold as given (window computation)
    uint16_t *lists;               // [F*C, B] in/out
    int32_t *cnt;                  // [F*C]   in/out
    uint8_t *visited;              // [F*C]   1 once a list went through round 2 (it is an independent set)
    const TrackState *st;
    const float *tracks;
    float t32;
    int *status;
    int mask_words;
    int lazy;                      // 1: the lists of regular frames are maintained by the pick
    const int *n_irregular;        // device counter of irregular frames (frame_flags_kernel)
    int rb_bias;                   // batched videos: groups[] holds absolute row offsets, boxes / row_meta are the video's own
};

// one (frame, class) list p = f * C + c, one wave
__device__ __forceinline__ void track_suppress_list(const SuppressParams &prm, lds_mask_t mask, int lane, int p)
{
    const int f = p / prm.C, c = p - f * prm.C;
    const TrackState s = prm.st[c];
    if (!s.active) return;
    const float *row = prm.tracks + (((int64_t)c * prm.max_tracks + s.ntracks) * prm.F + f) * 5;
    const float tx1 = row[0];
    if (tx1 != tx1) return;                           // the track has no box on this frame
    const float4 tb = make_float4(tx1, row[1], row[2], row[3]);
    const float tarea = box_area(tb);
    const int B = prm.B, rb = prm.groups[f].box_off - prm.rb_bias;
    uint16_t *list = prm.lists + (int64_t)p * B;
    const int n = prm.cnt[p];
    if (n == 0) return;
    const bool has_z = prm.group_z[f] != 0;

    const bool seen = prm.visited[p] != 0;
    // lists of regular frames are maintained lazily by track_pick_kernel (see LazyLists)
    if (prm.lazy && !has_z && prm.group_flags && (prm.group_flags[f] & kFlagRegular)) return;
    int nk = 0, bad = 0;
    if (seen) {
        // The list already went through round 2 once: it is an independent set of the frame's
        // suppression graph, so vid_nms of the round-1 survivors keeps them all -- only round 1
        // (utils/nms.pyx:163-183) can remove entries.  (A zero-union pair inside the list would
        // have raised on the first visit.)
        for (int q0 = 0; q0 < n; q0 += 64) {
            const int q = q0 + lane;
            const bool valid = q < n;
            const int cidx = (int)list[min(q, n - 1)];
            const float4 bd = prm.boxes[rb + cidx];
            const uint32_t pp = pair_pred(bd, box_area(bd), tb, tarea, prm.t32);
            if (valid && (pp & 2u)) bad = 1;
            const bool stay = valid && !(pp & 1u);
            const unsigned long long sm = __ballot(stay);
            if (stay) list[nk + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u))] = (uint16_t)cidx;
            nk += __popcll(sm);
        }
        if (lane == 0) prm.cnt[p] = nk;
        if (__ballot(bad != 0) && lane == 0) atomicOr(prm.status, kStDivZero);
        return;
    }
    for (int i = lane; i < ((B + 31) >> 5); i += 64) mask[i] = 0u;
    if (has_z) {   // detections that left the list earlier are "not in d": dead for the zero-union rule
        for (int i = lane; i < ((B + 31) >> 5); i += 64) mask[i] = 0xFFFFFFFFu;
        for (int q = lane; q < n; q += 64) {
            const int v = list[q];
            lds_and(mask, v >> 5, ~(1u << (v & 31)));
        }
    } else if (n > 2048 && prm.group_flags && (prm.group_flags[f] & kFlagRegular)) {
        // long list (first visit of a full frame): round 1 as ONE coalesced sweep over the frame's
        // boxes by index (sets the dead bit), instead of a 16-B gather per list entry.  Boxes that
        // are not in the list get a bit too -- harmless, they are never visited.
        const float tw = (tb.z - tb.x) + 1.0f;
        if (prm.ix.xbox && prm.thres > 1e-6 && tw > 0.0f && tw < 3.0e38f) {
            // only the x-window of the frame that can reach IoU >= thres with the track box
            int r0, r1;
            xwindow(prm.ix, f, tb.x, tw, prm.thres, r0, r1);
            for (int r = r0 + lane; r < r1; r += 64) {
                const float4 bd = prm.ix.xbox[(int64_t)f * B + r];
                const uint32_t pp = pair_pred(bd, box_area(bd), tb, tarea, prm.t32);
                if (pp & 1u) {
                    const int b = prm.ix.xord[(int64_t)f * B + r];
                    lds_or(mask, b >> 5, 1u << (b & 31));
                }
            }
        } else
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = min(b0 + lane, B - 1);
            const float4 bd = prm.boxes[rb + b];
            const uint32_t pp = pair_pred(bd, box_area(bd), tb, tarea, prm.t32);
            const unsigned long long hit = __ballot((b0 + lane < B) && (pp & 1u));
            if (lane == 0) { mask[b0 >> 5] = (uint32_t)hit; mask[(b0 >> 5) + 1] = (uint32_t)(hit >> 32); }
        }
    }
    // (regular frame: every det area is > 0, so no union with the track box can be zero)
    const bool swept = !has_z && n > 2048 && prm.group_flags && (prm.group_flags[f] & kFlagRegular);
    const int last = n - 1;
    int c_cur = (int)list[min(lane, last)];
    int stage = 0;
    for (int q0 = 0; q0 < n; q0 += 64) {
        const int q = q0 + lane;
        const bool valid = q < n;
        const int cidx = c_cur;
        const int c_nn = (int)list[min(q0 + 64 + lane, last)];       // next chunk's ids, one iteration ahead
        bool r1 = false;
        if (!swept) {
            // round 1 (utils/nms.pyx:163-183): the DET is the "i" box, the track the "j" box
            const float4 bd = prm.boxes[rb + cidx];
            const uint32_t pp = pair_pred(bd, box_area(bd), tb, tarea, prm.t32);
            if (valid && (pp & 2u)) bad = 1;
            r1 = valid && (pp & 1u);
            if (r1) lds_or(mask, cidx >> 5, 1u << (cidx & 31));
        }
        const bool alive = valid && !r1 && !((mask[cidx >> 5] >> (cidx & 31)) & 1u);
        unsigned long long am = __ballot(alive);
        uint2 meta = make_uint2(0u, 0u);
        if (alive) meta = prm.row_meta[rb + cidx];                   // alive lanes only (see walk_kernel)
        const uint32_t off = meta.x;
        const int deg = (int)meta.y;
        while (am) {
            int ls[kWalkGrp];
            int ng = 0;
#pragma unroll
            for (int k = 0; k < kWalkGrp; ++k) {
                ls[k] = k ? ls[k - 1] : 0;
                if (am) { ls[k] = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)am) - 1); am &= am - 1; ng = k + 1; }
            }
            uint16_t pre0[kWalkGrp], pre1[kWalkGrp];
#pragma unroll
            for (int k = 0; k < kWalkGrp; ++k) {
                const uint32_t o = __builtin_amdgcn_readlane(off, ls[k]);
                const int d = __builtin_amdgcn_readlane(deg, ls[k]);
                const int dm = max(d, 1) - 1;
                pre0[k] = prm.adj[o + min(lane, dm)];
                pre1[k] = prm.adj[o + min(lane + 64, dm)];
            }
#pragma unroll
            for (int k = 0; k < kWalkGrp; ++k) {
                if (k >= ng) break;
                const int cu = __builtin_amdgcn_readlane(cidx, ls[k]);
                if ((mask[cu >> 5] >> (cu & 31)) & 1u) continue;
                if ((nk & 63) == lane) stage = cu;
                ++nk;
                if ((nk & 63) == 0) list[nk - 64 + lane] = (uint16_t)stage;   // coalesced; positions < q0 + 64, all read already
                if (lane == 0) lds_or(mask, cu >> 5, 1u << (cu & 31));
                const int d = __builtin_amdgcn_readlane(deg, ls[k]);
                if (has_z) {
                    walk_apply_slice<true>(mask, pre0[k], lane < d, bad);
                    if (d > 64) walk_apply_slice<true>(mask, pre1[k], lane + 64 < d, bad);
                } else {
                    walk_apply_slice<false>(mask, pre0[k], lane < d, bad);
                    if (d > 64) walk_apply_slice<false>(mask, pre1[k], lane + 64 < d, bad);
                }
                if (d > 128) {   // rare: long lists
                    const uint32_t o = __builtin_amdgcn_readlane(off, ls[k]);
                    for (int e0 = 128; e0 < d; e0 += 64)
                        walk_apply_slice<true>(mask, prm.adj[o + min(e0 + lane, d - 1)], e0 + lane < d, bad);
                }
            }
        }
        c_cur = c_nn;
    }
    if (lane < (nk & 63)) list[(nk & ~63) + lane] = (uint16_t)stage;         // tail
    if (lane == 0) prm.visited[p] = 1;
    if (lane == 0) prm.cnt[p] = nk;
    if (__ballot(bad != 0) && lane == 0) atomicOr(prm.status, kStDivZero);
}

// grid-stride over the lists (4 waves per block, one list each): a video whose frames are all regular
// has nothing to do here when the pick maintains the lists lazily -- every block leaves after one load
// close the iteration: count the finished track
__global__ void track_init_kernel(TrackState *__restrict__ st, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    TrackState s;
    s.active = 1; s.ntracks = 0; s.last_key = 0; s.last_flat = -1; s.anchor_frame = 0; s.anchor_box = 0; s.anchor_score = 0.f; s.resolved = 0;
    st[c] = s;
}

// ------------------------------------------------------------------------------------------------
// The whole tracking loop of vdet/track.py:207-252 in ONE launch (round 3).  Nothing in an iteration of that loop
// crosses classes -- the anchor, the tubelet, the lists it suppresses and the stop rule all belong to one class; the
// only shared object is the link memo, whose entries are deterministic -- so one persistent block per class runs
// pick -> (copy | link forward, link backward) -> suppress (irregular frames only) -> commit for all max_tracks
// iterations without returning to the host queue: 1 launch instead of 4 per track, and no block waits for the
// slowest class of its iteration.  Same device functions as the per-iteration kernels, in the same order per class:
// bit-identical results.
// block = 256; dynamic LDS = 4 dead masks of sp.mask_words words (track_suppress_list).
// ------------------------------------------------------------------------------------------------
struct LoopArgs {
    const uint32_t *keys;
    uint16_t *lists;
    const int32_t *cnt;
    const float *scores;
    double thres, link_thres;
    float *anchors;
    float link_t32;
    int reach;
    unsigned long long *memo;
    unsigned int *stats;
    int32_t *nodes;
    int32_t *ntracks_out;
    int need_suppress;
};

__device__ __forceinline__ void track_loop_body(const int c, const LoopArgs &a, const LazyLists &lz, const ResolveArgs &rv, const SuppressParams &sp,
                                                unsigned char *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int F = sp.F, B = sp.B, C = sp.C, T = sp.max_tracks;
    TrackState *st = const_cast<TrackState *>(sp.st);
    for (int t = 0; t < T; ++t) {
        track_pick_body(c, a.keys, a.lists, a.cnt, F, B, C, a.scores, a.thres, T, st, a.anchors, lz, rv);
        __threadfence_block();
        __syncthreads();
        const TrackState s = st[c];
        if (!s.active) break;
        if (!s.resolved) {     // the anchor was not among the materialised warm chains: walk its two half tubelets here
            track_link_memo_body<256, 0, 8>(c, 1, sp.boxes, F, B, T, a.link_t32, a.reach, st, const_cast<float *>(sp.tracks),
                                            sp.group_flags, sp.ix, a.link_thres, a.memo, a.stats, nullptr, a.nodes);
            __syncthreads();
            track_link_memo_body<256, 0, 8>(c, -1, sp.boxes, F, B, T, a.link_t32, a.reach, st, const_cast<float *>(sp.tracks),
                                            sp.group_flags, sp.ix, a.link_thres, a.memo, a.stats, nullptr, a.nodes);
        }
        __threadfence_block();
        __syncthreads();
        if (a.need_suppress && !(sp.lazy && sp.group_flags && *sp.n_irregular == 0)) {
            // eager track_det_nms of this class's lists on the frames the pick does not maintain (irregular frames)
            lds_mask_t mask = lds_mask_ptr(smem, w * sp.mask_words);
            for (int f = w; f < F; f += 4) {
                track_suppress_list(sp, mask, lane, f * C + c);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            __threadfence_block();
            __syncthreads();
        }
        if (tid == 0) st[c].ntracks = s.ntracks + 1;        // track_commit_kernel
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) a.ntracks_out[c] = st[c].ntracks;
}

__global__ __launch_bounds__(256) void track_loop_kernel(const LoopArgs a, const LazyLists lz, const ResolveArgs rv, const SuppressParams sp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    track_loop_body(blockIdx.x, a, lz, rv, sp, smem);
}

// ------------------------------------------------------------------------------------------------
// Tubelet re-scoring of the device tracks (raw_dets_spatial_max_pooling, vdet/tubelet_cls.py:493-535,
// then do_score_completion :284-303 and score_proto_temporal_maxpool :386-414), all float64.
//   tracks [C,T,F,5] f32 (NaN = no box), ntracks [C]
//   out_score [C,T,F] f64 (NaN = no box), out_box [C,T,F,4] f32 (the "regressed" box)
// ------------------------------------------------------------------------------------------------
// one WAVE per (class, track, frame) (600 000 of them at c2: with one 256-thread block each the
// kernel was bound by the block launch rate and an 8-barrier LDS reduction), 4 per block.
// wv = (f * C + c) * T + t: neighbours in the dispatch order read the same frame's x-window (L2).
__device__ __forceinline__ void rescore_one_scan(int64_t wv, int tid, const float *__restrict__ tracks,
                                                 const int32_t *__restrict__ ntracks, const float4 *__restrict__ boxes,
                                                 const float *__restrict__ scores, int F, int B, int C, int T, double thres,
                                                 double *__restrict__ out_score, float *__restrict__ out_box,
                                                 const FrameIndex &ix, const uint32_t *__restrict__ group_flags)
{
    const int t = (int)(wv % T);
    const int fc = (int)(wv / T);
    const int f = fc / C, c = fc - f * C;
    const int64_t e = ((int64_t)c * T + t) * F + f;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    const float *row = tracks + e * 5;
    if (t >= ntracks[c] || row[0] != row[0]) {
        if (tid == 0) out_score[e] = qnan;
        if (tid < 4) out_box[e * 4 + tid] = __uint_as_float(0x7FC00000u);
        return;
    }
    const double p[4] = {(double)row[0], (double)row[1], (double)row[2], (double)row[3]};
    double bs = 0.0;
    int64_t bi = -1;
    const float wc = (row[2] - row[0]) + 1.0f;
    if (ix.xbox && group_flags && (group_flags[f] & kFlagRegular) && thres > 1e-6 && wc > 0.0f && wc < 3.0e38f) {
        int r0, r1;
        xwindow(ix, f, row[0], wc, thres, r0, r1);
        // f32 screen before the float64 IoU (vdet/tubelet_cls.py:514-532 computes it in f64): the f32
        // quotient is within a few 1e-7 (relative) of the exact one on regular boxes, so everything the
        // f64 test accepts satisfies inter > (thres - 1e-3) * union in f32; the rest (almost all of
        // the x-window) never reaches the half-rate f64 unit
        const float pa = ((row[2] - row[0]) + 1.0f) * ((row[3] - row[1]) + 1.0f);
        const float thr_lo = (float)thres - 1.0e-3f;
        for (int r = r0 + tid; r < r1; r += 64) {
            const float4 bb = ix.xbox[(int64_t)f * B + r];
            const float sw = (fminf(row[2], bb.z) - fmaxf(row[0], bb.x)) + 1.0f;
            const float sh = (fminf(row[3], bb.w) - fmaxf(row[1], bb.y)) + 1.0f;
            if (!(sw > 0.0f && sh > 0.0f)) continue;
            const float sinter = sw * sh;
            const float suni = (pa + box_area(bb)) - sinter;
            if (!(sinter > thr_lo * suni)) continue;
            const double q[4] = {(double)bb.x, (double)bb.y, (double)bb.z, (double)bb.w};
            if (iou_f64_pair(p, q) > thres) {
                const int64_t j = ix.xord[(int64_t)f * B + r];
                const double s = (double)scores[((int64_t)f * B + j) * C + c];
                if (argmax_better(s, j, bs, bi)) { bs = s; bi = j; }
            }
        }
    } else
    for (int j = tid; j < B; j += 64) {
        const float4 bb = boxes[(int64_t)f * B + j];
        const double q[4] = {(double)bb.x, (double)bb.y, (double)bb.z, (double)bb.w};
        if (iou_f64_pair(p, q) > thres) {
            const double s = (double)scores[((int64_t)f * B + j) * C + c];
            if (argmax_better(s, j, bs, bi)) { bs = s; bi = j; }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {        // wave argmax (first index on ties, NaN rules of argmax_better)
        const double s2 = __shfl_xor(bs, d, 64);
        const long long i2 = __shfl_xor((long long)bi, d, 64);
        if (i2 >= 0 && argmax_better(s2, (int64_t)i2, bs, bi)) { bs = s2; bi = (int64_t)i2; }
    }
    if (tid == 0) {
        if (bi >= 0) {
            const float4 bb = boxes[(int64_t)f * B + bi];
            out_score[e] = bs;
            out_box[e * 4 + 0] = bb.x; out_box[e * 4 + 1] = bb.y; out_box[e * 4 + 2] = bb.z; out_box[e * 4 + 3] = bb.w;
        } else {   // no overlapping detection: sentinel score, box unchanged (:526-530)
            out_score[e] = -1e5;
            out_box[e * 4 + 0] = row[0]; out_box[e * 4 + 1] = row[1]; out_box[e * 4 + 2] = row[2]; out_box[e * 4 + 3] = row[3];
        }
    }
}

// TODO_LIST = false: every (class, track, frame); true: only the entries rescore_adj_kernel could not serve (grid-stride)
template <bool TODO_LIST>
__global__ __launch_bounds__(256) void rescore_spatial_kernel(const float *__restrict__ tracks,
                                                              const int32_t *__restrict__ ntracks,
                                                              const float4 *__restrict__ boxes,
                                                              const float *__restrict__ scores, int F, int B, int C, int T,
                                                              double thres, double *__restrict__ out_score,
                                                              float *__restrict__ out_box, const FrameIndex ix,
                                                              const uint32_t *__restrict__ group_flags,
                                                              const int32_t *__restrict__ todo, const unsigned int *__restrict__ todo_cnt)
{
    const int tid = threadIdx.x & 63;
    const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (!TODO_LIST) {
        if (w0 < (int64_t)F * C * T) rescore_one_scan(w0, tid, tracks, ntracks, boxes, scores, F, B, C, T, thres, out_score, out_box, ix, group_flags);
    } else {
        const int64_t n = (int64_t)*todo_cnt;
        for (int64_t i = w0; i < n; i += (int64_t)gridDim.x * 4)
            rescore_one_scan(todo[i], tid, tracks, ntracks, boxes, scores, F, B, C, T, thres, out_score, out_box, ix, group_flags);
    }
}

// ------------------------------------------------------------------------------------------------
// The same spatial max-pool with the candidates taken from the suppression graph.  The tubelet box T is
// trunc(box of proposal j) and the link kernels recorded j (nodes): every detection with IoU(T, k) > thres then
// is j itself or one of j's NEIGHBOURS in the frame's graph -- 1 - IoU is a metric (Jaccard distance), so
//     IoU(b_j, b_k) >= IoU(T, b_k) + IoU(b_j, T) - 1 > nms_thres + margin
// whenever IoU(b_j, T) > min_self_iou = 1 - (thres - nms_thres) + margin (IoU(b_j, T) = 1 for integer boxes).
// ~93 candidates from the adjacency list instead of the ~1 100 boxes of the x-window, the same f64 predicate and
// arg-max decide.  16 lanes per tubelet box, 16 boxes per block (one wave per box spent its time in 5 dependent
// memory round trips: 0.72 ms); entries this path can not serve (no recorded node, box mismatch, irregular frame,
// IoU(b_j, T) too small) are appended to `todo` for rescore_spatial_kernel.
// ------------------------------------------------------------------------------------------------
// one tubelet box (class, track, frame) = wv, 16 lanes (l = lane in the group).  Returns false -- for all 16 lanes -- when the box
// can not be served from the graph (the caller hands it to the window scan)
__device__ __forceinline__ bool rescore_adj_one(const int64_t wv, const int l, const float *__restrict__ tracks, const int32_t *__restrict__ ntracks,
                                                const float4 *__restrict__ boxes, const float *__restrict__ scores,
                                                int F, int B, int C, int T, double thres, double *__restrict__ out_score,
                                                float *__restrict__ out_box, const uint32_t *__restrict__ group_flags,
                                                const int32_t *__restrict__ nodes, const uint2 *__restrict__ row_meta,
                                                const uint16_t *__restrict__ adj, double min_self_iou)
{
    const int t = (int)(wv % T);
    const int fc = (int)(wv / T);
    const int f = fc / C, c = fc - f * C;
    const int64_t e = ((int64_t)c * T + t) * F + f;
    // (round 4) the box is a chain of dependent reads (row -> node -> box + list head -> list -> boxes -> scores): what does
    // not depend on each other is requested together and without a branch of its own -- a load inside a condition is
    // waited for on the spot (the compiler also sinks an early load to its first conditional use)
    const float *row = tracks + e * 5;
    const float r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    const int nt = ntracks[c];
    const uint32_t gf = group_flags[f];
    const int jn = nodes[e];
    asm volatile("" :: "v"(r3), "v"(jn), "v"(nt), "v"(gf));     // (all seven in flight before the first is looked at)
    if (t >= nt || r0 != r0) {
        if (l == 0) out_score[e] = __longlong_as_double(0x7FF8000000000000ll);
        if (l < 4) out_box[e * 4 + l] = __uint_as_float(0x7FC00000u);
        return true;
    }
    const double p[4] = {(double)r0, (double)r1, (double)r2, (double)r3};
    const bool jok = (gf & kFlagRegular) && jn >= 0 && jn < B;
    const int j = jok ? jn : 0;
    const float4 bj = boxes[(int64_t)f * B + j];
    const uint2 meta = row_meta[(int64_t)f * B + j];
    asm volatile("" :: "v"(bj.w), "v"(meta.y));
    bool ok = false;
    if (jok) {
        const float4 tj = trunc4(bj);
        if (tj.x == r0 && tj.y == r1 && tj.z == r2 && tj.w == r3) {
            const double q[4] = {(double)bj.x, (double)bj.y, (double)bj.z, (double)bj.w};
            ok = iou_f64_pair(p, q) > min_self_iou;
        }
    }
    if (!ok) return false;     // (uniform over the 16 lanes of the box)
    const int n = (int)meta.y + 1;                // the neighbours + j itself
    double bs = 0.0;
    int64_t bi = -1;
    const float pa = ((r2 - r0) + 1.0f) * ((r3 - r1) + 1.0f);
    const float thr_lo = (float)thres - 1.0e-3f;  // f32 screen before the f64 IoU, as in the window scan
    constexpr int kU = 4;                         // candidates per lane and memory round trip
    for (int i0 = l; i0 < n; i0 += 16 * kU) {
        int kk[kU];
        float4 bbs[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = min(i0 + 16 * u, n - 1);
            const int a = (int)(adj[meta.x + (uint32_t)max(i - 1, 0)] & 0x7FFF);
            kk[u] = i == 0 ? j : a;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) bbs[u] = boxes[(int64_t)f * B + kk[u]];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (i0 + 16 * u >= n) continue;
            const float4 bb = bbs[u];
            const int64_t k = kk[u];
            const float sw = (fminf(r2, bb.z) - fmaxf(r0, bb.x)) + 1.0f;
            const float sh = (fminf(r3, bb.w) - fmaxf(r1, bb.y)) + 1.0f;
            if (!(sw > 0.0f && sh > 0.0f)) continue;
            const float sinter = sw * sh;
            const float suni = (pa + box_area(bb)) - sinter;
            if (!(sinter > thr_lo * suni)) continue;
            const double qq[4] = {(double)bb.x, (double)bb.y, (double)bb.z, (double)bb.w};
            if (iou_f64_pair(p, qq) > thres) {
                const double s = (double)scores[((int64_t)f * B + k) * C + c];
                if (argmax_better(s, k, bs, bi)) { bs = s; bi = k; }
            }
        }
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {             // arg-max over the box's 16 lanes
        const double s2 = __shfl_xor(bs, d, 16);
        const long long i2 = __shfl_xor((long long)bi, d, 16);
        if (i2 >= 0 && argmax_better(s2, (int64_t)i2, bs, bi)) { bs = s2; bi = (int64_t)i2; }
    }
    if (l == 0) {
        if (bi >= 0) {
            const float4 bb = boxes[(int64_t)f * B + bi];
            out_score[e] = bs;
            out_box[e * 4 + 0] = bb.x; out_box[e * 4 + 1] = bb.y; out_box[e * 4 + 2] = bb.z; out_box[e * 4 + 3] = bb.w;
        } else {   // no overlapping detection: sentinel score, box unchanged (:526-530)
            out_score[e] = -1e5;
            out_box[e * 4 + 0] = r0; out_box[e * 4 + 1] = r1; out_box[e * 4 + 2] = r2; out_box[e * 4 + 3] = r3;
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void rescore_adj_kernel(const float *__restrict__ tracks, const int32_t *__restrict__ ntracks,
                                                          const float4 *__restrict__ boxes, const float *__restrict__ scores,
                                                          int F, int B, int C, int T, double thres, double *__restrict__ out_score,
                                                          float *__restrict__ out_box, const uint32_t *__restrict__ group_flags,
                                                          const int32_t *__restrict__ nodes, const uint2 *__restrict__ row_meta,
                                                          const uint16_t *__restrict__ adj, double min_self_iou,
                                                          int32_t *__restrict__ todo, unsigned int *__restrict__ todo_cnt)
{
    const int l = threadIdx.x & 15;
    // block b runs on XCD b % 8: a contiguous eighth of the tubelet boxes (wv is frame-major: whole frames) per XCD, so that a
    // frame's boxes, lists and score rows are fetched into ONE L2 instead of eight (the grid is a multiple of 8 blocks)
    const int64_t per = gridDim.x >> 3;
    const int64_t wv = ((int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * 16 + (threadIdx.x >> 4);
    if (wv >= (int64_t)F * C * T) return;
    if (!rescore_adj_one(wv, l, tracks, ntracks, boxes, scores, F, B, C, T, thres, out_score, out_box, group_flags, nodes, row_meta, adj,
                         min_self_iou) && l == 0)
        todo[atomicAdd(todo_cnt, 1u)] = (int32_t)wv;
}

// one thread per (class, track): completion over the track's boxes (its non-NaN frames, in order),
// then the centred temporal max-pool of window w (w == 1: none) into out2.
__device__ __forceinline__ void rescore_series_serial_body(const int ct, double *__restrict__ sc, double *__restrict__ out2,
                                                           const int32_t *__restrict__ ntracks, int F, int C, int T, int window,
                                                           int *__restrict__ err)
{
    if (ct >= C * T) return;
    const int c = ct / T, t = ct - c * T;
    double *s = sc + (int64_t)ct * F;
    double *o = out2 + (int64_t)ct * F;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    for (int f = 0; f < F; ++f) o[f] = qnan;
    if (t >= ntracks[c]) return;
    // the tubelet = frames with a box; the linker produces one contiguous run
    int a = 0;
    while (a < F && s[a] != s[a]) ++a;
    int b = F;
    while (b > a && s[b - 1] != s[b - 1]) --b;
    const int n = b - a;
    if (n <= 0) return;
    double *v = s + a;
    for (int i = 0; i < n; ++i) {           // do_score_completion
        if (v[i] > -10) continue;
        int j = i;
        while (j < n && v[j] <= -10) ++j;
        if (i == 0) {
            if (j == n) { atomicOr(err, 1); return; }
            for (int k = i; k < j; ++k) v[k] = v[j];
        } else if (j == n) {
            for (int k = i; k < j; ++k) v[k] = v[i - 1];
        } else {
            const double l = v[i - 1], r = v[j];
            for (int k = i; k < j; ++k) v[k] = l + (r - l) * (double)(k - i + 1) / (double)(j - i + 1);
        }
    }
    const int h = window / 2;
    for (int i = 0; i < n; ++i) {           // score_proto_temporal_maxpool
        double m = v[i];
        for (int d = -h; d <= h; ++d) {
            const int g = i + d;
            const double x = (g < 0 || g >= n) ? -1e5 : v[g];
            m = (x > m) ? x : m;
        }
        o[a + i] = m;
    }
}

// (videos of more than kSeriesWaveMaxF frames: the wave form's LDS stage does not hold the series)
__global__ void rescore_series_kernel(double *__restrict__ sc, double *__restrict__ out2,
                                      const int32_t *__restrict__ ntracks, int F, int C, int T, int window,
                                      int *__restrict__ err)
{
    rescore_series_serial_body(blockIdx.x * blockDim.x + threadIdx.x, sc, out2, ntracks, F, C, T, window, err);
}

// The same, one WAVE per series (F <= kSeriesWaveMaxF): the series sits in LDS, a lane owns every 64th frame.  A missing
// score (<= -10) is filled from the nearest present scores on either side -- found by walking a flag array that is
// complete before anything is written, so the in-place fill of the serial kernel becomes order-free: a fill only reads
// present entries, and those never change.  Same arithmetic per element, same NaN behaviour (a NaN inside the run is
// neither missing nor filled and bounds the gaps next to it), same error rule (no present score at all).
constexpr int kSeriesWaveMaxF = 1536;

__device__ __forceinline__ void rescore_series_wave_body(const int blk, unsigned char *series_smem, double *__restrict__ sc, double *__restrict__ out2,
                                                         const int32_t *__restrict__ ntracks, int F, int C, int T,
                                                         int window, int *__restrict__ err, int stride_bytes)
{
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = blk * 4 + w;
    if (ct >= C * T) return;
    volatile double *v = reinterpret_cast<volatile double *>(series_smem + (size_t)w * stride_bytes);
    volatile unsigned char *miss = series_smem + (size_t)w * stride_bytes + (size_t)F * 8;
    const int c = ct / T, t = ct - c * T;
    double *s = sc + (int64_t)ct * F;
    double *o = out2 + (int64_t)ct * F;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    int amin = F, bmax = -1;
    for (int f = lane; f < F; f += 64) {
        const double x = s[f];
        v[f] = x;
        miss[f] = (x <= -10.0) ? 1 : 0;
        o[f] = qnan;
        if (x == x) { amin = min(amin, f); bmax = max(bmax, f); }
    }
    if (t >= ntracks[c]) return;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { amin = min(amin, __shfl_xor(amin, d, 64)); bmax = max(bmax, __shfl_xor(bmax, d, 64)); }
    const int a = amin, n = bmax + 1 - amin;       // the tubelet = the run from the first to the last frame with a box
    if (n <= 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool bad = false;
    for (int i = lane; i < n; i += 64) {           // do_score_completion
        if (!miss[a + i]) continue;
        int i0 = i, j = i + 1;
        while (i0 > 0 && miss[a + i0 - 1]) --i0;
        while (j < n && miss[a + j]) ++j;
        double x;
        if (i0 == 0) {
            if (j == n) { bad = true; continue; }
            x = v[a + j];
        } else if (j == n) {
            x = v[a + i0 - 1];
        } else {
            const double l = v[a + i0 - 1], r = v[a + j];
            x = l + (r - l) * (double)(i - i0 + 1) / (double)(j - i0 + 1);
        }
        v[a + i] = x;
        s[a + i] = x;
    }
    if (__ballot(bad)) { if (lane == 0) atomicOr(err, 1); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int h = window / 2;
    for (int i = lane; i < n; i += 64) {           // score_proto_temporal_maxpool
        double m = v[a + i];
        for (int d = -h; d <= h; ++d) {
            const int g = i + d;
            const double x = (g < 0 || g >= n) ? -1e5 : v[a + g];
            m = (x > m) ? x : m;
        }
        o[a + i] = m;
    }
}

__global__ __launch_bounds__(256) void rescore_series_wave_kernel(double *__restrict__ sc, double *__restrict__ out2,
                                                                  const int32_t *__restrict__ ntracks, int F, int C, int T,
                                                                  int window, int *__restrict__ err, int stride_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char series_smem[];
    rescore_series_wave_body(blockIdx.x, series_smem, sc, out2, ntracks, F, C, T, window, err, stride_bytes);
}

}  // namespace vdet
