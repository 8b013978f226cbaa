// nms_kernels.hpp -- gfx950 kernels for greedy NMS (utils/nms.pyx:17-189 of the reference).
//
// MI355X-first formulation (not the reference's sequential double loop):
//
//   The suppression predicate  IoU_f32(i, j) >= thresh  depends only on the two boxes, not on the
//   class whose scores order them.  So per FRAME (geometry group) we build the suppression graph
//   once, and every (frame, class) problem is a lexicographically-first maximal independent set on
//   that shared graph under its own priority order -- which is exactly what the reference's greedy
//   loop computes (box v is kept iff no higher-priority box that suppresses it is kept).
//
//   K0 frame_flags / frame_index   per frame: "regular" flag (finite boxes, positive sizes) and the
//                        x1-sorted index (boxes in x1 order + bucket table) that every window test uses.
//   K1s iou_bits_sym_kernel   regular frames: upper triangle of 256 x 256 tiles in x1-rank space, tiles /
//                        blocks beyond the IoU reach skipped, divide-free exact predicate, row words
//                        accumulated with add-with-carry, the transposed block by a 64 x 64 bit transpose
//                        on the lane-exchange network; also accumulates the rows' degrees.  At the VALU
//                        roofline.
//   K1 iou_bits_kernel   irregular frames (NaN / inf / degenerate boxes): all-pairs predicate with the
//                        reference's asymmetric NaN semantics, one lane per row (the "i" box), column
//                        boxes (the "j" boxes) broadcast from LDS; 64 predicates packed per u64, written
//                        transposed ([word][row]) so the stores and the later loads are coalesced.
//   K2 adj_build_kernel  bit rows -> compact u16 adjacency lists = OUT-lists "whom do I suppress"
//                        (CSR slabs allocated with one atomicAdd per 128-row block; lists 16-byte aligned)
//                        + one 32-byte record {box, list} per box of a regular frame for the packed walk.
//   K3 sort_kernel       one workgroup per (frame, class): stable LSD radix argsort of the scores,
//                        entirely in LDS (keys stay put, a u16 index list is permuted).
//   K4 walk_kernel       one WAVE per (frame, class): visits the candidates in descending order;
//                        a survivor ORs its adjacency list into a "dead" bitmask held in LDS.
//                        32 independent walks per CU hide the L2 latency of the list reads.
//                        Regular frames: eight candidates per pass (walk_list_packed), their mutual
//                        suppression decided geometrically by an 8 x 8 lane grid.
//
//   The O(B^2) float work is shared by all C classes; per class only integer work remains.
//   (A first version resolved each class as a parallel greedy MIS without sorting -- correct, but
//   latency-bound at the real graph degree (~90 at B = 10k): 1021 ms vs the sorted walk.)
//
// Exactness notes (all verified against the oracle / golden vectors in tests/):
//   * the predicate reproduces utils/nms.pyx:57-65 operation by operation in f32 (TU is compiled
//     with -ffp-contract=off; '/' is the correctly rounded IEEE division), with the reference's
//     "a if a >= b else b" max/min (NaN: second operand wins) and explicit i/j roles (the graph is
//     directed: row = i box, entry = j box), so it stays exact for NaN coordinates;
//   * "ovr >= thresh" is an f64 compare of the promoted f32 quotient (thresh is a boxed python
//     float); t32 = min{f in f32 : (double)f >= thresh} makes  ovr >= t32  the same predicate;
//   * a zero union raises ZeroDivisionError in the reference (Cython cdivision=False) only for
//     pairs it actually evaluates; zero-union pairs are kept as TAGGED adjacency entries and the
//     walk applies the evaluated-pair rule (entry not dead when its row box survives).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdet {

struct GroupDesc {
    int32_t box_off;   // first row of this group in the flat (grouped) box array
    int32_t nbox;      // boxes in the group (<= 32767)
    int64_t bits_off;  // u64-word offset of the group's bit matrix (bit_word) inside the batch scratch
};

// Bit-matrix layout (round 3): word c of row v lives at  ((v >> 6) * W + c) * 64 + (v & 63),  W = ceil(nbox / 64)  -- the 64
// rows of a wave are lane-contiguous (512 B per word, as before) and the words of one 64-row group FOLLOW each other, so
// the ~50 words K2 reads per row group are one contiguous stretch instead of 512-byte pieces 8 * nbox bytes apart.
__device__ __host__ __forceinline__ int64_t bit_word(int nbox, int v, int c)
{
    const int W = (nbox + 63) >> 6;
    return ((int64_t)(v >> 6) * W + c) * 64 + (v & 63);
}
__host__ __device__ __forceinline__ int64_t bit_words_of_group(int nbox)
{
    const int64_t W = (nbox + 63) >> 6;
    return W * W * 64;
}

struct TileDesc {      // one 256-row tile of one group
    int32_t group;
    int32_t row_tile;
};

constexpr int kRowsPerTile = 256;
constexpr uint16_t kZTag = 0x8000;
// Workgroup barrier that orders LDS traffic only: global loads stay in flight across it.  A plain __syncthreads() carries a
// workgroup-scope fence over ALL address spaces, which on gfx9 (one counter for loads and stores) is s_waitcnt vmcnt(0): every
// outstanding global load -- e.g. the next list's prefetched keys -- is drained at every barrier.
__device__ __forceinline__ void lds_only_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// status bits latched by kernels into ctx->d_status
constexpr int kStCap = 1;       // survivors > cap
constexpr int kStDivZero = 2;   // evaluated zero-union pair
constexpr int kStPool = 4;      // adjacency pool too small (internal, retried by the host)
constexpr int kStPoolAsync = 8; // ... in an asynchronous build (no retry possible: reported by vdet_sync)
constexpr int kStDirect = 32;   // a row has more neighbours than its direct slot holds (comes with kStPool: the host rebuilds through the bit matrix)
constexpr int kStBadOrder = 16; // a caller-supplied candidate list holds a count or an index out of range (vdet_nms_volume_ordered)
constexpr uint32_t kFlagRegular = 1u;   // group_flags bit: see frame_flags_kernel
constexpr uint32_t kFlagU16 = 2u;       // group_flags bit: every coordinate of the frame is an integer in [0, 65535] (and not -0.0)

// Sortable key of a float32 score: larger key == earlier in "argsort()[::-1]".
// -0.0 == +0.0 (numpy compares them equal); NaN sorts last ascending => first descending.
__device__ __forceinline__ uint32_t score_key(float s)
{
    if (s != s) return 0xFFFFFFFFu;
    if (s == 0.0f) s = 0.0f;  // -0.0 -> +0.0
    uint32_t b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// utils/nms.pyx:11-15
__device__ __forceinline__ float ref_max(float a, float b) { return a >= b ? a : b; }
__device__ __forceinline__ float ref_min(float a, float b) { return a <= b ? a : b; }

// numpy float32:  (x2 - x1 + 1) * (y2 - y1 + 1)       utils/nms.pyx:24
__device__ __forceinline__ float box_area(float4 b)
{
    return ((b.z - b.x) + 1.0f) * ((b.w - b.y) + 1.0f);
}

// One pair, box i = the kept / higher-priority box, box j = the candidate (utils/nms.pyx:57-65).
// Returns bit0 = suppress (ovr >= t32), bit1 = zero union.
__device__ __forceinline__ uint32_t pair_pred(float4 bi, float iarea, float4 bj, float jarea, float t32)
{
    const float xx1 = ref_max(bi.x, bj.x);
    const float yy1 = ref_max(bi.y, bj.y);
    const float xx2 = ref_min(bi.z, bj.z);
    const float yy2 = ref_min(bi.w, bj.w);
    const float w = ref_max(0.0f, (xx2 - xx1) + 1.0f);
    const float h = ref_max(0.0f, (yy2 - yy1) + 1.0f);
    const float inter = w * h;
    const float uni = (iarea + jarea) - inter;
    const float ovr = inter / uni;
    const uint32_t z = (uni == 0.0f) ? 2u : 0u;
    const uint32_t s = (ovr >= t32) ? 1u : 0u;
    return z ? z : s;
}

// ------------------------------------------------------------------------------------------------
// K1: all-pairs predicate bits.  grid = (n_tiles, col_splits); block = 256 (one lane per row v).
// The ROW box is the "i" box (the survivor that suppresses), the column box the "j" box:
// bits[g.bits_off + bit_word(nbox, v, w)] bit k  <=>  box v (as i) suppresses box u = 64*w+k (as j), u != v,
// i.e. row v is v's OUT-list -- what the greedy walk needs when v survives.
// row_z[flat v] += number of zero-union partners of v.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iou_bits_kernel(const float4 *__restrict__ boxes,
                                                       const GroupDesc *__restrict__ groups,
                                                       const TileDesc *__restrict__ tiles, float t32,
                                                       uint64_t *__restrict__ bits,
                                                       uint32_t *__restrict__ row_z,
                                                       uint32_t *__restrict__ group_z,
                                                       const uint32_t *__restrict__ group_flags)
{
    __shared__ float4 sbox[256];
    __shared__ float sarea[256];
    const TileDesc td = tiles[blockIdx.x];
    if (group_flags && (group_flags[td.group] & kFlagRegular)) return;   // done by iou_bits_sym_kernel
    const GroupDesc gd = groups[td.group];
    const int B = gd.nbox;
    const int tid = threadIdx.x;
    const int v = td.row_tile * kRowsPerTile + tid;
    const int W = (B + 63) >> 6;
    // column words handled by this split, in multiples of 4 words (= one 256-box LDS tile)
    int wper = (W + gridDim.y - 1) / gridDim.y;
    wper = (wper + 3) & ~3;
    const int w0 = blockIdx.y * wper;
    const int w1 = min(W, w0 + wper);
    if (w0 >= w1) return;

    const float qnan = __uint_as_float(0x7FC00000u);
    float4 brow = make_float4(qnan, qnan, qnan, qnan);
    if (v < B) brow = boxes[gd.box_off + v];
    const float rarea = box_area(brow);
    uint32_t zcnt = 0;

    for (int wt = w0; wt < w1; wt += 4) {
        __syncthreads();
        {
            const int u = wt * 64 + tid;
            float4 bi = make_float4(qnan, qnan, qnan, qnan);
            if (u < B) bi = boxes[gd.box_off + u];
            sbox[tid] = bi;
            sarea[tid] = box_area(bi);   // NaN for padding columns: predicate and zero test both false
        }
        __syncthreads();
        const int nw = min(4, w1 - wt);
        for (int q = 0; q < nw; ++q) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t p = pair_pred(brow, rarea, sbox[q * 64 + k], sarea[q * 64 + k], t32);
                lo |= (p & 1u) << k;
                zcnt += p >> 1;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t p = pair_pred(brow, rarea, sbox[q * 64 + 32 + k], sarea[q * 64 + 32 + k], t32);
                hi |= (p & 1u) << k;
                zcnt += p >> 1;
            }
            uint64_t m = ((uint64_t)hi << 32) | lo;
            const int cbase = (wt + q) * 64;
            if (v >= cbase && v < cbase + 64) {
                m &= ~(1ull << (v - cbase));                       // no self edge
                const uint32_t ps = pair_pred(brow, rarea, brow, rarea, t32);
                zcnt -= ps >> 1;                                   // ... and no self zero-union
            }
            if (v < B) bits[gd.bits_off + bit_word(B, v, wt + q)] = m;
        }
    }
    if (v < B && zcnt) {
        atomicAdd(&row_z[gd.box_off + v], zcnt);
        group_z[td.group] = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-frame proposal index (class-independent, built once per video): the frame's boxes sorted by
// x1 ascending (xbox / xord = original index) plus a 256-bucket cumulative table over
// [xmin, xmax] and the frame's largest width.  A box b can only reach IoU >= t with a box c if
//     x1c - (1-t) * Wmax  <=  x1b  <=  x1c + (1-t) * w_c         (+1 convention, real arithmetic:
// IoU <= iw / w_c and IoU <= iw / w_b, iw <= x2c - x1b + 1, iw <= x2b - x1c + 1), so linking,
// spatial max-pooling and the round-1 sweep only read that window (widened by 1 px + 0.1 %, far
// more than any rounding of the f32/f64 IoU).  Used for regular frames only.
// ------------------------------------------------------------------------------------------------
struct FrameIndex {
    const float4 *xbox;      // flat, at each group's box_off: its boxes in x1-ascending order
    const uint16_t *xord;    // flat: original (in-group) index of each sorted box
    const uint32_t *cum;     // [G*257] cum[k] = #boxes with bucket < k
    const float *info;       // [G*4] xmin, scale (= 256 / (xmax - xmin)), wmax, unused
    // kFlagU16 frames only (null: not built): the same sorted boxes as four u16 (8 B instead of 16) and their indices,
    // each group at an EVEN position pair_pos() so that two neighbours leave in one 16-byte / one 4-byte load --
    // the LINK window scans move 10 instead of 18 bytes per candidate
    const uint2 *xbox16;
    const uint16_t *xord16;
    int64_t bias16;          // (batched videos: position of the view's first frame, see pair_pos)
};

// position of group g (first box box_off) in xbox16 / xord16: every group adds at most one pad element
__device__ __host__ __forceinline__ int64_t pair_pos(int64_t box_off_plus_g) { return (box_off_plus_g + 1) & ~(int64_t)1; }

__device__ __forceinline__ int xbucket(float x, float xmin, float scale)
{
    const float t = (x - xmin) * scale;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

// keys for the x1 sort: k = ~score_key(x1)  (sort_kernel sorts by descending key => ascending x1)
__global__ void xkey_kernel(const float4 *__restrict__ boxes, uint32_t *__restrict__ keys, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ~score_key(boxes[i].x);
}

// one block per frame (group): gather the sorted copy, frame extrema, bucket table
__global__ __launch_bounds__(256) void frame_index_kernel(const float4 *__restrict__ boxes, const GroupDesc *__restrict__ groups,
                                                          const uint16_t *__restrict__ xord_all, float4 *__restrict__ xbox_all,
                                                          uint32_t *__restrict__ cum, float *__restrict__ info,
                                                          const uint32_t *__restrict__ group_flags, uint2 *__restrict__ xbox16,
                                                          uint16_t *__restrict__ xord16)
{
    __shared__ float smin[256], smax[256], swm[256];
    __shared__ uint32_t hist[257];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int B = groups[f].nbox;
    const int64_t fo = groups[f].box_off;
    const float4 *fb = boxes + fo;
    const uint16_t *xord = xord_all + fo;
    float4 *xbox = xbox_all + fo;
    float mn = 3.0e38f, mx = -3.0e38f, wm = 0.0f;
    const bool u16 = xbox16 && (group_flags[f] & kFlagU16);
    const int64_t p16 = pair_pos(fo + f);
    for (int r = tid; r < B; r += 256) {
        const uint16_t o = xord[r];
        const float4 b = fb[o];
        xbox[r] = b;
        if (u16) {
            xbox16[p16 + r] = make_uint2((uint32_t)b.x | ((uint32_t)b.y << 16), (uint32_t)b.z | ((uint32_t)b.w << 16));
            xord16[p16 + r] = o;
        }
        mn = fminf(mn, b.x); mx = fmaxf(mx, b.x); wm = fmaxf(wm, (b.z - b.x) + 1.0f);
    }
    smin[tid] = mn; smax[tid] = mx; swm[tid] = wm;
    for (int i = tid; i < 257; i += 256) hist[i] = 0;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) { smin[tid] = fminf(smin[tid], smin[tid + d]); smax[tid] = fmaxf(smax[tid], smax[tid + d]); swm[tid] = fmaxf(swm[tid], swm[tid + d]); }
        __syncthreads();
    }
    const float xmin = smin[0], xmax = smax[0];
    const float scale = xmax > xmin ? 256.0f / (xmax - xmin) : 0.0f;
    for (int r = tid; r < B; r += 256) atomicAdd(&hist[xbucket(fb[r].x, xmin, scale) + 1], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k <= 256; ++k) { run += hist[k]; cum[(int64_t)f * 257 + k] = run; }
        info[f * 4 + 0] = xmin; info[f * 4 + 1] = scale; info[f * 4 + 2] = swm[0]; info[f * 4 + 3] = 0.0f;
    }
}

// rank range [r0, r1) of the boxes of frame f whose x1 lies in the IoU >= t window of box c
__device__ __forceinline__ void xwindow(const FrameIndex &ix, int f, float x1c, float wc, double t, int &r0, int &r1)
{
    const float xmin = ix.info[f * 4 + 0], scale = ix.info[f * 4 + 1], wmax = ix.info[f * 4 + 2];
    // a partner j that starts left of c needs x1c - x1_j <= (1 - t) * w_j, and IoU >= t bounds its width:
    // t <= inter / union <= (wc * h_j) / (w_j * h_j)  =>  w_j <= wc / t  (so not only w_j <= wmax)
    const double wj = fmin((double)wmax, (double)wc / t * 1.001);
    const double lo = (double)x1c - (1.0 - t) * wj * 1.001 - 1.0;
    const double hi = (double)x1c + (1.0 - t) * (double)wc * 1.001 + 1.0;
    const int b0 = xbucket((float)fmax(lo, -3.0e38), xmin, scale);
    const int b1 = xbucket((float)fmin(hi, 3.0e38), xmin, scale);
    // (float) rounding of lo/hi moves them by < 1 ulp of a pixel coordinate, covered by the margin;
    // a bucket holds every x1 that maps to it, so [cum[b0], cum[b1 + 1]) is a superset of the window
    r0 = (int)ix.cum[(int64_t)f * 257 + b0];
    r1 = (int)ix.cum[(int64_t)f * 257 + b1 + 1];
}

// ------------------------------------------------------------------------------------------------
// K0: per-frame "regular" flag.  A frame is regular when every box is finite with positive
// width/height (+1 convention) and a finite area: then the predicate is symmetric in (i, j) (no NaN
// for the asymmetric max/min to see), unions are > 0 (no ZeroDivisionError) and the fast kernel
// below is exact.  Irregular frames take the general iou_bits_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_flags_kernel(const float4 *__restrict__ boxes,
                                                          const GroupDesc *__restrict__ groups,
                                                          uint32_t *__restrict__ group_flags,
                                                          int *__restrict__ n_irregular)
{
    const GroupDesc gd = groups[blockIdx.x];
    int bad = 0, wide = 0;
    const float inf = __uint_as_float(0x7F800000u);
    for (int v = threadIdx.x; v < gd.nbox; v += 256) {
        const float4 b = boxes[gd.box_off + v];
        const float w = (b.z - b.x) + 1.0f, h = (b.w - b.y) + 1.0f;
        const float a = w * h;
        const bool ok = fabsf(b.x) < inf && fabsf(b.y) < inf && fabsf(b.z) < inf && fabsf(b.w) < inf &&
                        w > 0.0f && h > 0.0f && a < inf;
        bad |= ok ? 0 : 1;
        // pixel coordinates that four u16 hold exactly (NaN fails the compares, -0.0 the sign test)
        const bool small = b.x == truncf(b.x) && b.y == truncf(b.y) && b.z == truncf(b.z) && b.w == truncf(b.w) &&
                           fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)) <= 65535.0f &&
                           ((__float_as_uint(b.x) | __float_as_uint(b.y) | __float_as_uint(b.z) | __float_as_uint(b.w)) >> 31) == 0u;
        wide |= small ? 0 : 1;
    }
    const int any_bad = __syncthreads_or(bad);
    const int any_wide = __syncthreads_or(wide);
    if (threadIdx.x == 0) {
        group_flags[blockIdx.x] = (any_bad ? 0u : kFlagRegular) | (any_wide ? 0u : kFlagU16);
        if (any_bad) atomicAdd(n_irregular, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// K1s: predicate bits of REGULAR frames, upper triangle only.  grid = tile pairs (rt <= ct) of
// 256 x 256 boxes; block = 256 = 4 waves, wave w owns the 64 rows of word-row r = 4*rt + w.
// For each column word c >= r the wave evaluates the 64 x 64 pairs once and emits BOTH
//   bits[c][64r + lane]   (row-major accumulation, bit k = column 64c + k)   and, for c > r,
//   bits[r][64c + lane]   (the transpose: 64 ballots, gathered with v_writelane)
// -- the predicate is symmetric on regular frames.
// Predicate, exact without a divide: with r = fma(-t32, uni, inter) (one rounding, sign exact),
//   r >= 0            =>  inter/uni >= t32            =>  RN(inter/uni) >= t32   (true)
//   r <  -2^-22*t32*uni => inter/uni < pred(t32)-ish  =>  RN(inter/uni) <  t32   (false)
//   otherwise (|r| tiny, e.g. IoU exactly 3/10): the wave falls back to the IEEE division.
// Requires 0 < t32 < inf (the host routes other thresholds to the general kernel).
// ------------------------------------------------------------------------------------------------
// The exact (IEEE divide) predicate as an out-of-line call: it is needed for ~1e-6 of the pairs, and
// when it is inlined hipcc if-converts the rare branch and computes the divide for EVERY pair
// (55 instead of ~30 VALU instructions per pair, seen in the ISA).
__device__ __attribute__((noinline)) bool pair_pred_exact_slow(float4 bi, float iarea, float4 bj, float jarea, float t32)
{
    return (pair_pred(bi, iarea, bj, jarea, t32) & 1u) != 0;
}

struct TilePair {
    int32_t group;
    int16_t rt, ct;
};

// v_max_f32 / v_min_f32 without the canonicalisation hipcc adds around fmaxf (inputs are finite here)
__device__ __forceinline__ float amax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float amin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// XSORTED: the caller knows bc.x >= br.x (K1s off-diagonal blocks: columns come later in the x1 order), so
// max(br.x, bc.x) is bc.x -- one half-rate v_max_f32 less per pair
template <bool XSORTED = false>
__device__ __forceinline__ bool pred_regular(float4 br, float rarea, float4 bc, float carea, float t32, float t32e,
                                             bool &border)
{
    const float xx1 = XSORTED ? bc.x : amax(br.x, bc.x);
    const float yy1 = amax(br.y, bc.y);
    const float xx2 = amin(br.z, bc.z);
    const float yy2 = amin(br.w, bc.w);
    const float w = amax(0.0f, (xx2 - xx1) + 1.0f);
    // (h is NOT clamped: v_max_f32 issues at half rate on gfx950 -- profiles/r02_valu_bench.csv -- and a negative h
    //  only makes inter <= 0, hence r <= -t32 * uni: "no", and never "borderline", exactly like the clamped form)
    const float h = (yy2 - yy1) + 1.0f;
    const float inter = w * h;
    const float uni = (rarea + carea) - inter;
    const float r = __builtin_fmaf(-t32, uni, inter);
    const float bnd = t32e * uni;                       // 2^-21 * t32 * uni  >> the half-ulp zone
    const bool p = r >= 0.0f;
    border = !p && (r >= -bnd);                         // (r is never NaN on regular frames: !p == r < 0)
    return p;
}

// The same test as two SIGNED margins (K1s, round 3): r = inter - t32 * uni and q = inter - t_lo * uni, each one fma
// (sign exact), t_lo = t32 * (1 - 2^-21) rounded.  sign(r) clear <=> the predicate holds; sign(q) clear while sign(r) is
// set <=> the pair sits in the band just below the threshold where only the IEEE quotient decides (q >= r always, the
// band is >= 0.875 * 2^-21 * t32 * uni wide, the test above needs 2^-22).  The caller shifts BOTH sign bits into
// accumulators with one v_alignbit_b32 each -- no compare, no add-with-carry, no per-pair border flag: after 32 pairs the
// two words differ iff some pair was borderline.  (v_alignbit_b32 turned out to issue at half rate like v_cmp,
// profiles/r03_valu_bench.csv: 2 fma + 2 alignbit ~ 6 issue slots against fma + mul + 2 v_cmp + v_addc ~ 7.)
// (r is never -0.0: an exact zero rounds to +0.0, and the product -t32 * uni is never zero on a regular frame)
// INTS: the frame's coordinates are integers (kFlagU16) and x2 / y2 arrive with the +1 already added -- every sum below
// is an exact integer whatever the order, so the two "+ 1" of the reference's formula cost nothing.
template <bool XSORTED, bool INTS>
__device__ __forceinline__ void pred_margins(float4 br, float rarea, float4 bc, float carea, float t32, float t_lo, float &r, float &q)
{
    const float xx1 = XSORTED ? bc.x : amax(br.x, bc.x);
    const float yy1 = amax(br.y, bc.y);
    const float xx2 = amin(br.z, bc.z);
    const float yy2 = amin(br.w, bc.w);
    const float w = INTS ? amax(0.0f, xx2 - xx1) : amax(0.0f, (xx2 - xx1) + 1.0f);
    const float h = INTS ? yy2 - yy1 : (yy2 - yy1) + 1.0f;       // (not clamped: see pred_regular)
    const float inter = w * h;
    const float uni = (rarea + carea) - inter;
    r = __builtin_fmaf(-t32, uni, inter);
    q = __builtin_fmaf(-t_lo, uni, inter);
}

// acc = (acc << 1) | sign bit of v, one full-rate VALU instruction
__device__ __forceinline__ void shl1_or_sign(uint32_t &acc, float v)
{
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(v), 31);
}

// ------------------------------------------------------------------------------------------------
// 64 x 64 bit-matrix transpose inside one wave: lane j holds row j as (lo, hi); afterwards lane k
// holds column k.  Six butterfly stages (distance 32, 16, 8, 4, 2, 1), each exchanging the
// off-diagonal blocks of every 2k x 2k sub-matrix between lane pairs (j, j ^ k):
//   32: v_permlane32_swap   lanes 32..63 of lo  <->  lanes 0..31 of hi       (gfx950)
//   16: v_permlane16_swap + a byte permute        8: DPP row_ror:8 + a byte permute
//   4 / 2 / 1: DPP (row_shl/shr:4, quad_perm) + rotate + bit-field insert
// ~31 VALU instructions instead of the 128 v_writelanes (plus their hazard nops) of building the
// transposed words from 64 ballots.  The lane-exchange semantics are verified on the device at
// vdet_create (wave_transpose_probe); K1s falls back to the ballot form if that ever fails.
// ------------------------------------------------------------------------------------------------
struct TransposeConsts {     // per lane, computed once per kernel
    uint32_t sel16, sel8;    // v_perm_b32 selectors
    uint32_t sh4, m4, sh2, m2, sh1, m1;
};

__device__ __forceinline__ TransposeConsts transpose_consts(int lane)
{
    TransposeConsts t;
    // v_perm_b32 D = perm(S0, S1, sel): selector byte 0-3 -> S1.byte, 4-7 -> S0.byte
    t.sel16 = (lane & 16) ? 0x03020706u : 0x01000504u;   // S0 = r0, S1 = r1 (see stage 16)
    t.sel8 = (lane & 8) ? 0x03070105u : 0x06020400u;     // S0 = partner, S1 = own
    t.sh4 = (lane & 4) ? 4u : 28u;  t.m4 = (lane & 4) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
    t.sh2 = (lane & 2) ? 2u : 30u;  t.m2 = (lane & 2) ? 0xCCCCCCCCu : 0x33333333u;
    t.sh1 = (lane & 1) ? 1u : 31u;  t.m1 = (lane & 1) ? 0xAAAAAAAAu : 0x55555555u;
    return t;
}

__device__ __forceinline__ uint32_t transpose_stage_small(uint32_t r, uint32_t partner, uint32_t sh, uint32_t m)
{
    // upper lane of the pair keeps its bits at positions with the stage bit clear and takes the partner's
    // (shifted up); the lower lane the mirror image: one rotate + one bit-field insert
    const uint32_t rot = __builtin_amdgcn_alignbit(partner, partner, sh);
    return (r & m) | (rot & ~m);
}

__device__ __forceinline__ void wave_transpose64(uint32_t &lo, uint32_t &hi, const TransposeConsts &t)
{
    {   // 32: upper lanes' hi <-> lower lanes' lo
        const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
        lo = r[0]; hi = r[1];
    }
    {   // 16 (per 32-bit register): r0 = {own on even rows, partner on odd rows}, r1 the converse
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        lo = __builtin_amdgcn_perm(a[0], a[1], t.sel16);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        hi = __builtin_amdgcn_perm(b[0], b[1], t.sel16);
    }
    {   // 8: partner = lane ^ 8 = row_ror:8
        const uint32_t pl = __builtin_amdgcn_update_dpp(0u, lo, 0x128, 0xf, 0xf, false);
        const uint32_t ph = __builtin_amdgcn_update_dpp(0u, hi, 0x128, 0xf, 0xf, false);
        lo = __builtin_amdgcn_perm(pl, lo, t.sel8);
        hi = __builtin_amdgcn_perm(ph, hi, t.sel8);
    }
    {   // 4: partner = lane ^ 4: row_shl:4 into banks 0,2 and row_shr:4 into banks 1,3
        uint32_t pl = __builtin_amdgcn_update_dpp(0u, lo, 0x104, 0xf, 0x5, false);
        pl = __builtin_amdgcn_update_dpp(pl, lo, 0x114, 0xf, 0xa, false);
        uint32_t ph = __builtin_amdgcn_update_dpp(0u, hi, 0x104, 0xf, 0x5, false);
        ph = __builtin_amdgcn_update_dpp(ph, hi, 0x114, 0xf, 0xa, false);
        lo = transpose_stage_small(lo, pl, t.sh4, t.m4);
        hi = transpose_stage_small(hi, ph, t.sh4, t.m4);
    }
    {   // 2: quad_perm [2,3,0,1]
        const uint32_t pl = __builtin_amdgcn_update_dpp(0u, lo, 0x4E, 0xf, 0xf, false);
        const uint32_t ph = __builtin_amdgcn_update_dpp(0u, hi, 0x4E, 0xf, 0xf, false);
        lo = transpose_stage_small(lo, pl, t.sh2, t.m2);
        hi = transpose_stage_small(hi, ph, t.sh2, t.m2);
    }
    {   // 1: quad_perm [1,0,3,2]
        const uint32_t pl = __builtin_amdgcn_update_dpp(0u, lo, 0xB1, 0xf, 0xf, false);
        const uint32_t ph = __builtin_amdgcn_update_dpp(0u, hi, 0xB1, 0xf, 0xf, false);
        lo = transpose_stage_small(lo, pl, t.sh1, t.m1);
        hi = transpose_stage_small(hi, ph, t.sh1, t.m1);
    }
}

// self-test of wave_transpose64: n matrices of 64 x u64 in, transposed out (compared on the host)
__global__ __launch_bounds__(64) void wave_transpose_probe(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, int n)
{
    const int lane = threadIdx.x;
    const TransposeConsts tc = transpose_consts(lane);
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const uint64_t v = in[(size_t)t * 64 + lane];
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        wave_transpose64(lo, hi, tc);
        out[(size_t)t * 64 + lane] = ((uint64_t)hi << 32) | lo;
    }
}

// acc = (acc << 1) | p in ONE VALU instruction: add-with-carry of acc to itself, the carry-in being the
// compare's own lane mask (instead of v_cndmask + v_or per pair)
__device__ __forceinline__ void shl1_or_pred(uint32_t &acc, bool p)
{
    const unsigned long long m = __ballot(p);
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(carry_out) : "s"(m));
}

// Rows and columns are RANKS of the frame's x1-sorted order (FrameIndex::xbox); adj_build_kernel
// translates back to box indices.  A tile pair whose columns all start to the right of every row's
// IoU >= t reach (x1 + (1-t) * w, the same necessary condition as xwindow) is all-zero and is
// written as such without evaluating a single pair: ~2.8x fewer pair tests at B = 10k.
// Per word-row (64 consecutive ranks) of every regular frame: how far to the right a partner of ANY of its rows can
// start (reach) and where its first box starts (first).  Block (r, c), c > r, of the predicate matrix can hold a set bit
// only if first[c] <= reach[r]; K1s evaluates -- and WRITES -- only such blocks, and K2 reads a word (c, row of r) only
// if the same test, on the same table, says it was written (for c < r it came from block (c, r)'s transposed store).
// The all-zero blocks used to be written out as zeros: 64 % of the 4.5 GB bit matrix of a config-2 video.
// Table slot of word-row r of group g: (box_off >> 6) + g + r   (disjoint: every group adds at most one partial row).
__device__ __forceinline__ int reach_slot(const GroupDesc &gd, int g) { return (gd.box_off >> 6) + g; }

__global__ __launch_bounds__(256) void reach_table_kernel(const float4 *__restrict__ xbox, const GroupDesc *__restrict__ groups,
                                                          const uint32_t *__restrict__ group_flags, float one_minus_t,
                                                          float2 *__restrict__ table)
{
    const int g = blockIdx.x;
    if (!(group_flags[g] & kFlagRegular)) return;
    const GroupDesc gd = groups[g];
    const int B = gd.nbox, W = (B + 63) >> 6;
    const int lane = threadIdx.x & 63;
    for (int r = threadIdx.x >> 6; r < W; r += 4) {
        const int v = r * 64 + lane;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < B) b = xbox[gd.box_off + v];
        // (1 px + 0.1 % margin, cf. xwindow)
        float reach = v < B ? b.x + one_minus_t * ((b.z - b.x) + 1.0f) * 1.001f + 1.0f : -3.0e38f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) reach = fmaxf(reach, __shfl_xor(reach, d, 64));
        const float first = __shfl(b.x, 0, 64);
        if (lane == 0) table[reach_slot(gd, g) + r] = make_float2(reach, first);
    }
}

// (the 32-pair loops are unrolled 8 at a time: fully unrolled, the scheduler hoists the LDS reads of a dozen columns and
//  ends at 158 registers, 3 waves per SIMD and 2.5 ms)
template <bool WT>
__global__ __launch_bounds__(256) void iou_bits_sym_kernel(const float4 *__restrict__ xbox,
                                                           const GroupDesc *__restrict__ groups,
                                                           const uint32_t *__restrict__ group_flags,
                                                           const TilePair *__restrict__ pairs, float t32, float one_minus_t,
                                                           uint64_t *__restrict__ bits, uint32_t *__restrict__ row_deg,
                                                           const float2 *__restrict__ reach_table)
{
    __shared__ float4 sbox[256];
    __shared__ float sarea[256];
    const TilePair tp = pairs[blockIdx.x];
    if (!(group_flags[tp.group] & kFlagRegular)) return;
    const GroupDesc gd = groups[tp.group];
    const int B = gd.nbox;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = tp.rt * 4 + w;                 // word-row of this wave
    const int v = r * 64 + lane;                 // my row (rank)
    // which of this tile pair's blocks can hold a set bit at all (reach_table_kernel); a tile pair without any is done
    const float2 *rtab = reach_table + reach_slot(gd, tp.group);
    const int W = (B + 63) >> 6;
    const float my_reach = r < W ? rtab[r].x : -3.0e38f;
    if (tp.ct > tp.rt) {
        float tr = -3.0e38f;
        for (int k = 0; k < 4; ++k) if (tp.rt * 4 + k < W) tr = fmaxf(tr, rtab[tp.rt * 4 + k].x);
        if (tp.ct * 4 >= W || rtab[tp.ct * 4].y > tr) return;       // (block-uniform)
    }
    const float t32e = t32 * 4.76837158203125e-7f;   // 2^-21
    const float t_lo = t32 * (1.0f - 4.76837158203125e-7f);
    const TransposeConsts tcs = transpose_consts(lane);
    // integer pixel coordinates (kFlagU16): x2 + 1 / y2 + 1 are formed once per box instead of once per pair
    const bool ints = WT && (group_flags[tp.group] & kFlagU16) != 0u;

    float4 br = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < B) br = xbox[gd.box_off + v];
    const float rarea = box_area(br);
    float4 brx = br;
    if (ints) { brx.z += 1.0f; brx.w += 1.0f; }
    {
        const int u = tp.ct * 256 + tid;
        float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u < B) bc = xbox[gd.box_off + u];
        sarea[tid] = box_area(bc);
        if (ints) { bc.z += 1.0f; bc.w += 1.0f; }
        sbox[tid] = bc;
    }
    __syncthreads();
    const int rows_left = B - r * 64;
    const unsigned long long rowvalid = rows_left >= 64 ? ~0ull : (rows_left > 0 ? ((1ull << rows_left) - 1ull) : 0ull);

    uint32_t dsum = 0;
    for (int q = 0; q < 4; ++q) {
        const int c = tp.ct * 4 + q;
        if (c < r) continue;                     // lower triangle: produced by the transposed stores
        const int cols_left = B - c * 64;
        if (cols_left <= 0) break;
        const unsigned long long colvalid = cols_left >= 64 ? ~0ull : ((1ull << cols_left) - 1ull);
        uint32_t lo = 0, hi = 0, tlo = 0, thi = 0;
        // the reach test per 64 x 64 block (the columns of block c start at first[c]): a block out of reach is neither
        // evaluated nor written -- K2 applies the same test before it reads
        if (rows_left <= 0 || (c > r && rtab[c].y > my_reach)) continue;
        {
            bool anyb = false;
            if (WT) {
                // sign bits of the two margins, shifted in (descending k leaves column k in bit k); lo / hi = the complement
                uint32_t nr0 = 0, nq0 = 0, nr1 = 0, nq1 = 0;
#define VDET_MARGIN_BLOCK(XS, IN) \
                _Pragma("unroll 1") for (int g = 0; g < 4; ++g) { \
                    _Pragma("unroll") for (int j = 0; j < 8; ++j) { \
                        const int k = 31 - 8 * g - j; \
                        float mr, mq; \
                        pred_margins<XS, IN>(brx, rarea, sbox[q * 64 + k], sarea[q * 64 + k], t32, t_lo, mr, mq); \
                        shl1_or_sign(nr0, mr); shl1_or_sign(nq0, mq); \
                    } \
                } \
                _Pragma("unroll 1") for (int g = 0; g < 4; ++g) { \
                    _Pragma("unroll") for (int j = 0; j < 8; ++j) { \
                        const int k = 31 - 8 * g - j; \
                        float mr, mq; \
                        pred_margins<XS, IN>(brx, rarea, sbox[q * 64 + 32 + k], sarea[q * 64 + 32 + k], t32, t_lo, mr, mq); \
                        shl1_or_sign(nr1, mr); shl1_or_sign(nq1, mq); \
                    } \
                }
                // (off-diagonal block: every column starts at or to the right of every row in the x1 order)
                if (ints) { if (c > r) { VDET_MARGIN_BLOCK(true, true) } else { VDET_MARGIN_BLOCK(false, true) } }
                else { if (c > r) { VDET_MARGIN_BLOCK(true, false) } else { VDET_MARGIN_BLOCK(false, false) } }
#undef VDET_MARGIN_BLOCK
                lo = ~nr0; hi = ~nr1;
                anyb = (nr0 != nq0) | (nr1 != nq1);
            } else {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const int k = 31 - kk;      // descending: acc = 2*acc + p (one v_addc) leaves column k in bit k
                bool border;
                const bool p = pred_regular(br, rarea, sbox[q * 64 + k], sarea[q * 64 + k], t32, t32e, border);
                anyb |= border;
                shl1_or_pred(lo, p);
                if (!WT) {
                    const unsigned long long b = __ballot(p);
                    tlo = (lane == k) ? (uint32_t)b : tlo;
                    thi = (lane == k) ? (uint32_t)(b >> 32) : thi;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const int k = 31 - kk;
                bool border;
                const bool p = pred_regular(br, rarea, sbox[q * 64 + 32 + k], sarea[q * 64 + 32 + k], t32, t32e, border);
                anyb |= border;
                shl1_or_pred(hi, p);
                if (!WT) {
                    const unsigned long long b = __ballot(p);
                    tlo = (lane == 32 + k) ? (uint32_t)b : tlo;
                    thi = (lane == 32 + k) ? (uint32_t)(b >> 32) : thi;
                }
            }
            }
            if (__builtin_expect(__ballot(anyb) != 0ull, 0)) {
                // rare (~1e-6 of the pairs sit in the half-ulp band, e.g. IoU exactly 3/10):
                // redo this 64 x 64 block with the IEEE quotient
                lo = hi = tlo = thi = 0;
                for (int k = 0; k < 64; ++k) {
                    float4 bk = sbox[q * 64 + k];
                    if (ints) { bk.z -= 1.0f; bk.w -= 1.0f; }        // (exact: integers)
                    const bool p = pair_pred_exact_slow(br, rarea, bk, sarea[q * 64 + k], t32);
                    if (k < 32) lo |= p ? (1u << k) : 0u; else hi |= p ? (1u << (k - 32)) : 0u;
                    if (!WT) {
                        const unsigned long long b = __ballot(p);
                        if (lane == k) { tlo = (uint32_t)b; thi = (uint32_t)(b >> 32); }
                    }
                }
            }
            if (WT && c > r) {   // the transposed block from the row-major words, on the lane-exchange network
                tlo = lo; thi = hi;
                wave_transpose64(tlo, thi, tcs);
            }
        }
        unsigned long long m = (((unsigned long long)hi << 32) | lo) & colvalid;
        if (c == r) m &= ~(1ull << lane);        // no self edge
        if (v < B) { bits[gd.bits_off + bit_word(B, v, c)] = m; dsum += (uint32_t)__popcll(m); }
        if (c > r) {
            const int u = c * 64 + lane;
            if (u < B) {
                const unsigned long long tm = (((unsigned long long)thi << 32) | tlo) & rowvalid;
                bits[gd.bits_off + bit_word(B, u, r)] = tm;
                const uint32_t cnt = (uint32_t)__popcll(tm);
                if (cnt) atomicAdd(&row_deg[gd.box_off + u], cnt);
            }
        }
    }
    // the rows' degrees (list lengths) come for free here; adj_build_kernel does not have to count them
    if (dsum) atomicAdd(&row_deg[gd.box_off + v], dsum);
}

// ------------------------------------------------------------------------------------------------
// K2: bit rows -> adjacency lists.  grid = 2 * n_tiles; block = 128 (one lane per row).
// Each block reserves one contiguous slab of the u16 pool with a single atomicAdd; the slab layout
// is irrelevant to the result (lists are sets).  Zero-union partners are appended with kZTag.
// ------------------------------------------------------------------------------------------------
// One block = kAdjRows rows = HALF a 256-row tile (grid = 2 * n_tiles): the rows' lists are staged in LDS and leave in
// one coalesced copy; a slab that does not fit takes the slow direct path (a dependent index load per edge), so the
// stage is sized for ~2x the graph degree of 10 000 random boxes (92) while leaving room for 4 blocks per CU.
// what the packed walk needs to queue one alive candidate, in ONE 32-byte record per box (its coordinates, its list's
// offset and length) instead of a gather per table: that walk is bound by the cache lines it requests (the L1's
// outstanding misses and the address unit: profiles/r02_pmc_tcp.csv), not by bytes
struct WalkMeta {
    float4 box;
    uint4 row;           // x: list offset in the pool (u16 units, a multiple of 8), y: its length
};

constexpr int kMaxWordRows = 320;       // 64-box word-rows of a regular frame (B <= 17 408: 272)
constexpr int kAdjRows = 128;
constexpr int kAdjBatch = 8;         // bit-matrix words loaded per memory round trip (a row's window is ~42 words)
constexpr int kAdjStage = 16384;     // u16 entries staged in LDS per block (32 KB)

// ROWS = rows per block: kAdjRows (two blocks per 256-row tile), or kRowsPerTile on small frames (one block per tile: a third
// fewer blocks for frames of 257..384 boxes -- there the kernel is a cost per BLOCK)
template <int ROWS>
__global__ __launch_bounds__(ROWS) void adj_build_kernel(const float4 *__restrict__ boxes,
                                                        const GroupDesc *__restrict__ groups,
                                                        const TileDesc *__restrict__ tiles,
                                                        const uint64_t *__restrict__ bits,
                                                        const uint32_t *__restrict__ row_z,
                                                        uint2 *__restrict__ row_meta,
                                                        uint16_t *__restrict__ adj,
                                                        unsigned long long *__restrict__ pool_used,
                                                        unsigned long long pool_cap, int *__restrict__ status,
                                                        const uint32_t *__restrict__ group_flags,
                                                        const FrameIndex ix, float one_minus_t, int pool_bits,
                                                        WalkMeta *__restrict__ wmeta, const float2 *__restrict__ reach_table,
                                                        int skip_regular)
{
    // (skip_regular: the regular groups of this launch are adj_rows_kernel's, adjrows_kernels.hpp -- block-uniform exit)
    if (skip_regular && group_flags && (group_flags[tiles[ROWS == kAdjRows ? (blockIdx.x >> 1) : blockIdx.x].group] & kFlagRegular)) return;
    __shared__ float2 srt[kMaxWordRows];       // regular group: its reach table (which words of a row were written at all)
    __shared__ uint32_t sscan[8];
    __shared__ unsigned long long sbase;
    __shared__ __attribute__((aligned(16))) uint16_t sstage[kAdjStage];
    const TileDesc td = tiles[ROWS == kAdjRows ? (blockIdx.x >> 1) : blockIdx.x];
    const GroupDesc gd = groups[td.group];
    const int B = gd.nbox;
    const int tid = threadIdx.x;
    const int v = td.row_tile * kRowsPerTile + (ROWS == kAdjRows ? (int)(blockIdx.x & 1) * kAdjRows : 0) + tid;
    const int W = (B + 63) >> 6;
    const uint64_t *col = bits + gd.bits_off + bit_word(B, v, 0);      // word c of my row: col[c * 64]
    // regular groups were evaluated in x1-rank space (iou_bits_sym_kernel): translate back
    const uint16_t *tr = (group_flags && (group_flags[td.group] & kFlagRegular)) ? ix.xord + gd.box_off : nullptr;
    const int vo = (tr && v < B) ? (int)tr[v] : v;       // the box this row belongs to
    // ... and only the words inside the row's IoU >= t window can be non-zero (the rest was
    // zero-filled by the tile skipping or is zero anyway): read just those
    int w0 = 0, w1 = W;
    // (round 3) the block is a chain of dependent memory round trips at 2 waves per SIMD: everything that only needs the
    // group's descriptor is requested HERE, in one go, before the first barrier -- the row's box, the frame's extrema, the
    // row's degree -- instead of one wait after the other further down
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    float ixmin = 0.f, iscale = 0.f, iwmax = 0.f;
    uint32_t rz = 0u;
    if (v < B) rz = row_z[gd.box_off + v];
    if (tr && v < B) {
        bx = ix.xbox[gd.box_off + v];
        ixmin = ix.info[td.group * 4 + 0]; iscale = ix.info[td.group * 4 + 1]; iwmax = ix.info[td.group * 4 + 2];
    }
    if (tr) {
        const float2 *rt = reach_table + reach_slot(gd, td.group);
        for (int i = tid; i < W; i += ROWS) srt[i] = rt[i];
        __syncthreads();
    }
    const int wr = __builtin_amdgcn_readfirstlane(v >> 6);     // my word-row (one per wave: 64 aligned rows)
    // word c of my row exists iff block (min, max) of the upper triangle was in reach (iou_bits_sym_kernel's own test):
    // load_batch below
    if (tr && v < B) {
        const float xmin = ixmin, scale = iscale, wmax = iwmax;
        const float wrow = (bx.z - bx.x) + 1.0f;
        // (partners starting to the left are at most wrow / t wide, see xwindow; 1 - one_minus_t <= t)
        const float wleft = fminf(wmax, wrow / fmaxf(1.0f - one_minus_t, 1.0e-6f) * 1.001f);
        const float lo = bx.x - one_minus_t * wleft * 1.001f - 2.0f;
        const float hi = bx.x + one_minus_t * wrow * 1.001f + 2.0f;
        const int r0 = (int)ix.cum[(int64_t)td.group * 257 + xbucket(lo, xmin, scale)];
        const int r1 = (int)ix.cum[(int64_t)td.group * 257 + xbucket(hi, xmin, scale) + 1];
        w0 = r0 >> 6;
        w1 = min(W, (r1 + 63) >> 6);
    }

    // the first batch of the row's words is requested before the scan and the slab reservation (a global atomic): their
    // round trips overlap; inside the loop below the NEXT batch is in flight while this one is taken apart
    // (round 4) TWO register sets that swap roles, and loads without a branch of their own (a word outside the row's
    // window is requested at a clamped address and masked afterwards): with a conditional load per word the compiler
    // cannot count the requests in flight and waits for ALL of them before the first use, and a set that is copied into
    // place at the loop's end has to have landed by then -- either way the "prefetch" was a round trip per batch
    uint64_t ma[kAdjBatch], mb[kAdjBatch];
    uint32_t oka = 0u, okb = 0u;               // which words of the set exist (bit j: word wb + j)
    const float2 swr = tr ? srt[wr] : make_float2(0.f, 0.f);
    auto load_batch = [&](uint64_t (&dst)[kAdjBatch], uint32_t &ok, int wb) {
        uint32_t okm = 0u;
#pragma unroll
        for (int j = 0; j < kAdjBatch; ++j) {      // (no branch, no short circuit: LDS reads first, then sixteen requests in a row)
            const int c = wb + j;
            const float2 sc = srt[min(c, kMaxWordRows - 1)];
            const bool lv = (tr == nullptr) | (c == wr) | (c > wr ? sc.y <= swr.x : swr.y <= sc.x);
            okm |= ((c < w1) & lv) ? (1u << j) : 0u;
        }
        ok = okm;
#pragma unroll
        for (int j = 0; j < kAdjBatch; ++j) dst[j] = col[min(wb + j, w1 - 1) * 64];
    };
    if (tr && v < B && w0 < w1) load_batch(ma, oka, w0);
    uint32_t deg = 0, zc = 0;
    if (v < B && tr) {
        deg = rz;                                // regular group: the degree was accumulated by iou_bits_sym_kernel
    } else if (v < B) {
        for (int wb = w0; wb < w1; wb += 8) {
            uint64_t mm[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) mm[j] = (wb + j < w1) ? col[(wb + j) * 64] : 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j) deg += __popcll(mm[j]);
        }
        zc = rz;
    }
    const uint32_t tot = deg + zc;
    // every list starts at an even pool offset (slabs are sums of even sizes): the walks read two
    // u16 entries with one 4-byte load
    // ... and is padded to a multiple of 8 entries (16 bytes): the packed walk reads it with aligned 16-byte loads
    const uint32_t tot_al = (tot + 7u) & ~7u;
    // block inclusive scan of the 256 list sizes: a DPP scan inside each wave (row_shr 1/2/4/8, then the
    // row_bcast 15/31 carries), the four wave totals combined through LDS -- two barriers instead of the
    // sixteen of a Hillis-Steele scan over LDS
    uint32_t incl = tot_al;
#define VDET_SCAN_STEP(CTRL, ROWMASK) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, CTRL, ROWMASK, 0xf, false);
    VDET_SCAN_STEP(0x111, 0xf) VDET_SCAN_STEP(0x112, 0xf) VDET_SCAN_STEP(0x114, 0xf) VDET_SCAN_STEP(0x118, 0xf)
    VDET_SCAN_STEP(0x142, 0xa) VDET_SCAN_STEP(0x143, 0xc)
#undef VDET_SCAN_STEP
    if ((tid & 63) == 63) sscan[tid >> 6] = incl;
    __syncthreads();
    {
        const int wv = tid >> 6;
        uint32_t carry = 0;
        for (int k = 0; k < wv; ++k) carry += sscan[k];
        incl += carry;
    }
    if (tid == ROWS - 1) { sscan[4] = incl; sbase = atomicAdd(pool_used, (unsigned long long)incl); }
    __syncthreads();
    const unsigned long long base = sbase;
    const uint32_t tile_total = sscan[4];
    if (base + tile_total > pool_cap || base + tile_total > 0xFFFFFFFFull) {
        if (tid == 0) atomicOr(status, pool_bits);
        if (v < B) row_meta[gd.box_off + vo] = make_uint2(0u, 0u);
        if (v < B && tr && wmeta) { wmeta[gd.box_off + vo].box = make_float4(0.f, 0.f, 0.f, 0.f); wmeta[gd.box_off + vo].row = make_uint4(0u, 0u, 0u, 0u); }
        return;
    }
    const bool staged = tile_total <= (uint32_t)kAdjStage;     // block-uniform
    const uint32_t lofs = incl - tot_al;                        // my list's offset inside the tile slab
    if (v < B) {
        uint32_t p = (uint32_t)base + lofs;
        row_meta[gd.box_off + vo] = make_uint2(p, tot);
        if (tr && wmeta) {        // regular group: the packed walk's record of this box
            wmeta[gd.box_off + vo].box = bx;
            wmeta[gd.box_off + vo].row = make_uint4(p, tot, 0u, 0u);
        }
        uint32_t q = lofs;
        if (!tr && w0 < w1) load_batch(ma, oka, w0);
        if (staged) {
            // (round 3) every half word of the batch gets its place in the stage from a prefix sum of the popcounts, so the
            // 2 * kAdjBatch extraction chains are independent of each other: at 2 waves per SIMD the kernel is bound by the
            // LATENCY of its dependent instructions (find-first-set -> store -> clear), not by their number.  A row has
            // ~1.2 neighbours per 32 columns: four unconditional slots per half word, a loop for what is left.
            auto stage_batch = [&](const uint64_t (&m)[kAdjBatch], uint32_t ok, int wb) {
                uint32_t qs = q;
#pragma unroll
                for (int j = 0; j < kAdjBatch; ++j) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint32_t h = hf ? (uint32_t)(m[j] >> 32) : (uint32_t)m[j];
                        h = ((ok >> j) & 1u) ? h : 0u;
                        const uint32_t col0 = (uint32_t)((wb + j) * 64 + hf * 32);
                        uint32_t qx = qs;
                        qs += (uint32_t)__popc(h);
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (h != 0u) sstage[qx] = (uint16_t)(col0 + (uint32_t)(__ffs((int)h) - 1));
                            qx += h != 0u ? 1u : 0u;
                            h &= h - 1u;
                        }
                        while (h) {
                            sstage[qx++] = (uint16_t)(col0 + (uint32_t)(__ffs((int)h) - 1));
                            h &= h - 1u;
                        }
                    }
                }
                q = qs;
            };
            for (int wb = w0; wb < w1; wb += 2 * kAdjBatch) {
                // (a prefetch past the wave's last window is skipped: a wave-uniform branch, the requests stay countable)
                if (__ballot(wb + kAdjBatch < w1)) load_batch(mb, okb, wb + kAdjBatch); else okb = 0u;
                stage_batch(ma, oka, wb);
                if (wb + kAdjBatch >= w1) break;
                if (__ballot(wb + 2 * kAdjBatch < w1)) load_batch(ma, oka, wb + 2 * kAdjBatch); else oka = 0u;
                stage_batch(mb, okb, wb + kAdjBatch);
            }
        } else {
            // a slab too large for the stage (rare): entries go straight to the pool, a dependent translation per edge
            for (int wb = w0; wb < w1; wb += kAdjBatch) {
                if (wb != w0) load_batch(ma, oka, wb);
#pragma unroll
                for (int j = 0; j < kAdjBatch; ++j) {
                    uint64_t m = ((oka >> j) & 1u) ? ma[j] : 0ull;
                    const int w = wb + j;
                    while (m) {
                        const int k = __ffsll((unsigned long long)m) - 1;
                        adj[p++] = tr ? tr[w * 64 + k] : (uint16_t)(w * 64 + k);
                        m &= m - 1;
                    }
                }
            }
        }
        if (zc) {  // rare: degenerate boxes.  Recompute which partners have a zero union.
            const float4 brow = boxes[gd.box_off + v];
            const float rarea = box_area(brow);
            for (int u = 0; u < B; ++u) {
                if (u == v) continue;
                const float4 bu = boxes[gd.box_off + u];
                if (pair_pred(brow, rarea, bu, box_area(bu), 0.0f) & 2u) {
                    const uint16_t e = (uint16_t)u | kZTag;
                    if (staged) sstage[q++] = e; else adj[p++] = e;
                }
            }
        }
        // odd lists are padded to even length with a DUPLICATE of their last entry: the walks apply
        // entries in pairs (a second OR of the same bit is harmless)
        // (the rest of the padding repeats it too: the staged copy-out below translates every entry of the slab)
        for (uint32_t e = tot; e < tot_al && tot > 0u; ++e) { if (staged) { sstage[q] = sstage[q - 1]; ++q; } else { adj[p] = adj[p - 1]; ++p; } }
    }
    if (staged) {   // one coalesced copy of the tile's slab instead of 256 interleaved 2-byte streams
        __syncthreads();
        // (round 4) eight entries per thread and step: lists are padded to multiples of 8 entries and slabs are sums of such,
        // so the slab is a whole number of aligned 16-byte groups -- one LDS read, eight independent translations in
        // flight, one store (before: one 2-byte store and one dependent translation per entry, four in flight)
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sstage);
        uint4 *d4 = reinterpret_cast<uint4 *>(adj + base);
        const uint32_t n8 = tile_total >> 3;
        if (tr) {   // (zero-union entries only exist on irregular frames, which have no x-index: tr == null)
            for (uint32_t i = tid; i < n8; i += ROWS) {
                const uint4 e = s4[i];
                const uint32_t a0 = tr[e.x & 0xFFFFu], a1 = tr[e.x >> 16], a2 = tr[e.y & 0xFFFFu], a3 = tr[e.y >> 16];
                const uint32_t a4 = tr[e.z & 0xFFFFu], a5 = tr[e.z >> 16], a6 = tr[e.w & 0xFFFFu], a7 = tr[e.w >> 16];
                d4[i] = make_uint4(a0 | (a1 << 16), a2 | (a3 << 16), a4 | (a5 << 16), a6 | (a7 << 16));
            }
        } else {
            for (uint32_t i = tid; i < n8; i += ROWS) d4[i] = s4[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: per-problem descending argsort of the scores, entirely in LDS.
// One workgroup per (frame, class).  Stable LSD radix sort, 4 passes x 8 bits, on the inverted
// sortable key; the initial arrangement is DESCENDING index, so equal scores come out by
// descending index (= scores.argsort(kind='stable')[::-1], the build's tie rule).  Only the u16
// index list is permuted (ping-pong), the keys stay put and are gathered through it.
// Non-candidates (score <= thr / excluded) get the largest inverted key and land at the tail;
// ncand[p] = number of real candidates.
// Stable ranks inside a 64-key chunk come from an 8-ballot "match" (lanes with the same digit),
// chunks of one wave are processed in order against per-(wave,digit) running bases.
// ------------------------------------------------------------------------------------------------
struct SortParams {
    int mode;                 // 0: volume [F,B,C]  1: volume [F,C,B]  2: flat, problem p == group p
                              // 3: transposed keys [F*C, B] (from transpose_keys_kernel)
    int P;
    int B, C;
    const float *scores;
    const uint32_t *keys;     // mode 2: explicit priorities (caller-supplied order) or null; mode 3: keys
    const uint8_t *excl;      // mode 2: flat [Ntot] nonzero = not a candidate
    int use_thr;
    float thr;
    const GroupDesc *groups;
    uint16_t *order;          // mode 0/1: [P,B]; mode 2: flat [Ntot] at the group's box_off
    int32_t *ncand;           // [P]
    int lds_idxa_off, lds_idxb_off, lds_base_off;   // dynamic-LDS carve, multiples of 16
    int topk;                 // > 0: only the topk best candidates stay candidates (vdet/video_det.py:93-95)
    int32_t *nover;           // [P] or null: candidates BEFORE the topk cut (vdet/video_det.py:93 tests len(cls_scores) > max_per_image)
};

struct ProblemRef { int g, N, rb; int64_t sbase, sstride, obase; };

__device__ __forceinline__ ProblemRef decode_problem(int mode, int p, int B, int C, const GroupDesc *groups)
{
    ProblemRef r;
    if (mode == 0) { r.g = p / C; const int c = p - r.g * C; r.sbase = (int64_t)r.g * B * C + c; r.sstride = C; r.obase = (int64_t)p * B; }
    else if (mode == 1 || mode == 3) { r.g = p / C; r.sbase = (int64_t)p * B; r.sstride = 1; r.obase = (int64_t)p * B; }
    else { r.g = p; r.sbase = groups[p].box_off; r.sstride = 1; r.obase = groups[p].box_off; }
    r.N = groups[r.g].nbox;
    r.rb = groups[r.g].box_off;
    return r;
}

// XCD-aware problem order: the dispatcher places block b on XCD b % 8; give each XCD a contiguous
// run of problems so the classes of one frame (which share that frame's score cache lines and
// adjacency lists) meet in the same L2.  Placement only affects speed.  grid is a multiple of 8.
__device__ __forceinline__ int xcd_problem(int bid, int nblocks)
{
    const int per = nblocks >> 3;
    return (bid & 7) * per + (bid >> 3);
}

// 64-lane "match": mask of the valid lanes holding the same 8-bit digit
__device__ __forceinline__ unsigned long long match8(uint32_t d, bool valid)
{
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// CPW = chunks (of 64 keys) per wave, compile-time so the per-chunk state lives in registers:
// each pass gathers its keys ONCE (phase 1: digit, stable rank inside the chunk, and -- through one
// returning LDS atomic per distinct digit -- the count of that digit in the wave's earlier chunks),
// then, after the cross-wave scan, scatters with a single independent LDS read per key (phase 2).
//
// ARANK = true: the stable rank comes straight from ONE returning LDS atomic per key
// (ds_add_rtn_u32 on the wave's private digit counter): lanes of one instruction that hit the same
// address are served in ascending lane order on gfx950, and a wave's LDS instructions execute in
// order, so the returned value IS "keys of this digit before me in this wave".  That ordering is
// measured, not documented: vdet_create runs lds_atomic_order_probe on thousands of conflict
// patterns and the host only selects ARANK when it holds; otherwise the 8-ballot match is used.
// (1024-thread variants up to CPW = 10, i.e. N <= 10 240, are register-capped to 8 waves per SIMD
// -- hipcc's second __launch_bounds__ argument -- so that two workgroups are resident per CU)
template <int BLOCK, int CPW, bool ARANK>
__device__ __forceinline__ void lsd_sort_problem(const SortParams &prm, const int p, unsigned char *smem)
{
    constexpr int NW = BLOCK / 64;
    // Only 16 key bits live in LDS at a time (passes 0-1 sort by the low half, passes 2-3 by the
    // high half, which waits in registers meanwhile): 6 B instead of 8 B per key, so TWO workgroups
    // fit in the CU's 160 KiB at B = 10 000 and one covers the other's barrier stalls.
    uint16_t *keys0 = reinterpret_cast<uint16_t *>(smem);
    uint16_t *src = reinterpret_cast<uint16_t *>(smem + prm.lds_idxa_off);
    uint16_t *dst = reinterpret_cast<uint16_t *>(smem + prm.lds_idxb_off);
    uint32_t *bases = reinterpret_cast<uint32_t *>(smem + prm.lds_base_off);   // [NW][256]
    uint32_t *tot = bases + NW * 256;                                           // [256] + [1] excluded count
    __shared__ uint32_t swt[4];                                                 // digit totals of the four scanning waves

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const ProblemRef pr = decode_problem(prm.mode, p, prm.B, prm.C, prm.groups);
    const int N = pr.N;

    if (tid == 0) tot[256] = 0;
    __syncthreads();
    uint32_t nx = 0;
    uint32_t khi[(CPW + 1) / 2];               // high halves of my keys (key v = tid + k * BLOCK), two per register
#pragma unroll
    for (int k = 0; k < (CPW + 1) / 2; ++k) khi[k] = 0xFFFFFFFFu;
    // Every key of the thread is requested BEFORE the first one is looked at (round 4, found in the ISA): with the load inside
    // the per-key `if (v < N) { if (prm.keys) ... }` hipcc kept each load in its own basic block and waited for it right there
    // -- CPW dependent memory round trips per list (10 at config 2) at the head of a kernel that has nothing else to do yet.
    uint32_t raw[CPW];
    uint32_t exb[CPW];
    {
        const int lastv = max(N - 1, 0);
        if (prm.keys) {
#pragma unroll
            for (int k = 0; k < CPW; ++k) raw[k] = prm.keys[pr.sbase + min(tid + k * BLOCK, lastv)];
        } else {
#pragma unroll
            for (int k = 0; k < CPW; ++k) raw[k] = __float_as_uint(prm.scores[pr.sbase + (int64_t)min(tid + k * BLOCK, lastv) * pr.sstride]);
        }
#pragma unroll
        for (int k = 0; k < CPW; ++k) exb[k] = 0u;
        if (prm.excl) {
#pragma unroll
            for (int k = 0; k < CPW; ++k) exb[k] = prm.excl[pr.rb + min(tid + k * BLOCK, lastv)];
        }
    }
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int v = tid + k * BLOCK;         // BLOCK * CPW >= N (host)
        if (v < N) {
            uint32_t ik;
            bool x = false;
            if (prm.keys) {                    // explicit priorities; 0 marks "not a candidate"
                const uint32_t kk = raw[k];
                ik = ~kk;
                x = (kk == 0u);
            } else {
                const float sc = __uint_as_float(raw[k]);
                ik = ~score_key(sc);
                if (prm.use_thr && !(sc > prm.thr)) x = true;
            }
            if (exb[k]) x = true;
            if (x) { ik = 0xFFFFFFFFu; ++nx; } // real inverted keys are <= 0xFF800000
            keys0[v] = (uint16_t)(ik & 0xFFFFu);
            khi[k >> 1] = (k & 1) ? ((khi[k >> 1] & 0x0000FFFFu) | (ik & 0xFFFF0000u))
                                  : ((khi[k >> 1] & 0xFFFF0000u) | (ik >> 16));
            src[v] = (uint16_t)(N - 1 - v);
        }
    }
    if (nx) atomicAdd(&tot[256], nx);
    __syncthreads();
    const int ncand = N - (int)tot[256];

    const int nchunks = (N + 63) >> 6;
    const int c0 = w * CPW;                      // host guarantees NW * CPW >= nchunks

    for (int pass = 0; pass < 4; ++pass) {
        const int shift = (pass & 1) * 8;
        if (pass == 2) {                       // every gather of the low halves is done (barrier at the end of pass 1)
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const int v = tid + k * BLOCK;
                if (v < N) keys0[v] = (uint16_t)((k & 1) ? (khi[k >> 1] >> 16) : (khi[k >> 1] & 0xFFFFu));
            }
        }
        for (int i = tid; i < NW * 256; i += BLOCK) bases[i] = 0;
        __syncthreads();
        uint32_t ra[CPW];        // index | digit << 16
        uint32_t rl[CPW];        // offset inside this wave's run of the digit
        bool peel_on = false;    // this pass's digits are concentrated: peel them (see below)
#pragma unroll
        for (int ch = 0; ch < CPW; ++ch) {
            const int q = (c0 + ch) * 64 + lane;
            const bool valid = (c0 + ch) < nchunks && q < N;
            const uint32_t i = valid ? src[q] : 0u;
            const uint32_t d = valid ? (((uint32_t)keys0[i] >> shift) & 255u) : 0u;
            ra[ch] = i | (d << 16);
            // The top byte (sign + exponent bits) takes only a handful of values: 64 lanes adding to 2-3
            // addresses serialise inside the LDS atomic unit (measured: that pass cost 1.75 ms, the
            // other three 0.5 ms each), so the last pass always ranks by ballots.
            if (ARANK) {
                // 64 lanes adding to 2-3 LDS addresses serialise inside the atomic unit.  That is the rule
                // for the top byte (sign + exponent bits: a handful of values; measured: that pass cost
                // 1.75 ms, the other three 0.5 ms each) and happens in any pass on quantised scores.  So a
                // digit shared by many lanes is peeled with one ballot (rank = position among its lanes,
                // ONE atomic by one lane), at most twice; the lanes left use their own atomics.  In the
                // other passes the first probe (one ballot + popcount) decides whether peeling pays.
                uint32_t r = 0u;
                bool pending = valid;
#pragma unroll
                for (int peel = 0; peel < 2; ++peel) {
                    if (ch > 0 && !peel_on) break;
                    const unsigned long long rem = __ballot(pending);
                    if (rem) {                                   // (scalar)
                        const int l0 = __ffsll((unsigned long long)rem) - 1;
                        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, l0);
                        const bool mine = pending && d == d0;
                        const unsigned long long m = __ballot(mine);
                        if (ch == 0 && peel == 0) peel_on = (pass == 3) || __popcll(m) >= 12;   // (the wave's first chunk decides for the pass)
                        if (peel_on) {
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            uint32_t old = 0u;
                            if (lane == l0) old = atomicAdd(&bases[w * 256 + d0], (uint32_t)__popcll(m));
                            old = (uint32_t)__builtin_amdgcn_readlane((int)old, l0);
                            if (mine) { r = old + rank; pending = false; }
                        } else {
                            break;                               // digits look spread out: per-lane atomics
                        }
                    }
                }
                if (pending) r = atomicAdd(&bases[w * 256 + d], 1u);
                rl[ch] = r;
            } else {
                const unsigned long long peers = match8(d, valid);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
                uint32_t old = 0;
                if (valid && rank == 0) old = atomicAdd(&bases[w * 256 + d], (uint32_t)__popcll(peers));
                const int leader = valid ? (__ffsll((unsigned long long)peers) - 1) : lane;
                old = __shfl(old, leader, 64);
                rl[ch] = old + rank;
            }
        }
        __syncthreads();
        // digit offsets folded into the per-wave bases in two steps (round 3; three before, the middle one a single wave's
        // shuffle scan): thread d sums digit d over the waves and the 64 digits of a wave are scanned on the DPP network;
        // after the barrier every thread adds the earlier waves' totals and writes base[wave][d] = offset of digit d +
        // keys of digit d in earlier waves -- the scatter then makes ONE table look-up per key
        uint32_t dsum = 0, dincl = 0;
        if (tid < 256) {
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) dsum += bases[ww * 256 + tid];
            dincl = dsum;
#define VDET_SCAN_STEP(CTRL, ROWMASK) dincl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dincl, CTRL, ROWMASK, 0xf, false);
            VDET_SCAN_STEP(0x111, 0xf) VDET_SCAN_STEP(0x112, 0xf) VDET_SCAN_STEP(0x114, 0xf) VDET_SCAN_STEP(0x118, 0xf)
            VDET_SCAN_STEP(0x142, 0xa) VDET_SCAN_STEP(0x143, 0xc)
#undef VDET_SCAN_STEP
            if (lane == 63) swt[w] = dincl;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t run = dincl - dsum;
            for (int k = 0; k < w; ++k) run += swt[k];
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const uint32_t c = bases[ww * 256 + tid];
                bases[ww * 256 + tid] = run;
                run += c;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < CPW; ++ch) {
            const int q = (c0 + ch) * 64 + lane;
            const bool valid = (c0 + ch) < nchunks && q < N;
            if (valid) {
                const uint32_t d = ra[ch] >> 16;
                dst[bases[w * 256 + d] + rl[ch]] = (uint16_t)(ra[ch] & 0xFFFFu);
            }
        }
        __syncthreads();
        uint16_t *t = src; src = dst; dst = t;
    }
    int ncand_out = ncand;
    if (prm.topk > 0 && ncand > prm.topk) {
        // vdet/video_det.py:93-95: top = argsort(-cls_scores)[:max_per_image] keeps, among candidates tied
        // with the k-th best score, the LOWEST indices (stable argsort of the negated scores), while this
        // list holds ties by DESCENDING index: move the last (k - a) entries of the tied run [a, b) that
        // straddles position k to [a, k).  One wave; the run is one entry long unless scores tie.
        const int k = prm.topk;
        if (w == 0) {
            auto key_at = [&](int q) -> uint32_t {
                const int i = src[q];
                if (prm.keys) return prm.keys[pr.sbase + i];
                return score_key(prm.scores[pr.sbase + (int64_t)i * pr.sstride]);
            };
            const uint32_t kk = key_at(k - 1);
            int a = k - 1, b = k;
            for (;;) {          // extend the run downwards
                const int q = a - 1 - lane;
                const bool eq = q >= 0 && key_at(q) == kk;
                const unsigned long long m = __ballot(eq);
                const int n = m == ~0ull ? 64 : (__ffsll((unsigned long long)~m) - 1);
                a -= n;
                if (n < 64) break;
            }
            for (;;) {          // ... and upwards (candidates only)
                const int q = b + lane;
                const bool eq = q < ncand && key_at(q) == kk;
                const unsigned long long m = __ballot(eq);
                const int n = m == ~0ull ? 64 : (__ffsll((unsigned long long)~m) - 1);
                b += n;
                if (n < 64) break;
            }
            const int nsel = k - a, s0 = b - nsel;          // s0 >= a: an ascending chunked copy never reads what it wrote
            if (s0 > a)
                for (int t = 0; t < nsel; t += 64) {
                    const bool v = t + lane < nsel;
                    const uint16_t e = v ? src[s0 + t + lane] : (uint16_t)0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (v) src[a + t + lane] = e;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
        }
        ncand_out = k;
        __syncthreads();
    }
    uint16_t *out = prm.order + pr.obase;
    if (((pr.obase | (int64_t)N) & 1) == 0) {      // two indices per lane: 256-B instead of 128-B stores per wave
        uint32_t *out2 = reinterpret_cast<uint32_t *>(out);
        const uint32_t *src2 = reinterpret_cast<const uint32_t *>(src);
        for (int v = tid; v < (N >> 1); v += BLOCK) out2[v] = src2[v];
    } else {
        for (int v = tid; v < N; v += BLOCK) out[v] = src[v];
    }
    if (tid == 0) {
        prm.ncand[p] = ncand_out;
        if (prm.nover) prm.nover[p] = ncand;
    }
}

template <int BLOCK, int CPW, bool ARANK>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 && CPW <= 10) ? 8 : 1) void sort_kernel(const SortParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int p = xcd_problem(blockIdx.x, gridDim.x);
    if (p >= prm.P) return;
    lsd_sort_problem<BLOCK, CPW, ARANK>(prm, p, smem);
}

// The same sort for a LIST of problems (the ones binsort_kernel could not spread, binsort_kernels.hpp): persistent
// workgroups stride over list[0 .. *count).  An empty list costs one launch of workgroups that leave at once.
template <int BLOCK, int CPW, bool ARANK>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 && CPW <= 10) ? 8 : 1) void sort_list_kernel(const SortParams prm, const int32_t *__restrict__ list,
                                                                                                   const int *__restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = *count;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        lsd_sort_problem<BLOCK, CPW, ARANK>(prm, list[i], smem);
        __syncthreads();             // the next problem rewrites the LDS this one's final copy reads
    }
}

// Self-test for ARANK (see sort_kernel): for each of the n patterns, every lane adds 1 to
// tbl[pat[lane]] with a returning LDS atomic, twice in a row; the returned values must equal the
// number of lower lanes (+ the whole first instruction for the second) with the same target.
__global__ __launch_bounds__(64) void lds_atomic_order_probe(const uint8_t *__restrict__ pats, int n, int *__restrict__ bad)
{
    __shared__ uint32_t tbl[256];
    const int lane = threadIdx.x;
    int nbad = 0;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        for (int i = lane; i < 256; i += 64) tbl[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t d = pats[t * 64 + lane];
        const bool act = d != 255;                 // 255 = lane inactive in this pattern
        uint32_t r1 = 0, r2 = 0;
        if (act) r1 = atomicAdd(&tbl[d], 1u);
        if (act) r2 = atomicAdd(&tbl[d], 1u);
        const unsigned long long peers = match8(d, act);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        const uint32_t cnt = (uint32_t)__popcll(peers);
        if (act && (r1 != rank || r2 != cnt + rank)) nbad = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (__ballot(nbad) && lane == 0) atomicOr(bad, 1);
}

// ------------------------------------------------------------------------------------------------
// Score volume [F,B,C] (class innermost) -> sortable keys [F,C,B]: a coalesced 64x64 LDS transpose,
// so that the per-(frame,class) sort reads contiguous keys instead of one 4-B element per 128-B
// line.  key = score_key(s); 0 = "not a candidate" (score <= thr).  HBM-bound: 4 B in + 4 B out.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_keys_kernel(const float *__restrict__ scores, uint32_t *__restrict__ keys,
                                                             int B, int C, int use_thr, float thr)
{
    __shared__ uint32_t tile[64][65];
    const int f = blockIdx.z;
    const int b0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    const float *src = scores + (int64_t)f * B * C;
    for (int r = ty; r < 64; r += 4) {
        const int b = b0 + r, c = c0 + tx;
        uint32_t k = 0;
        if (b < B && c < C) {
            const float s = src[(int64_t)b * C + c];
            k = score_key(s);
            if (use_thr && !(s > thr)) k = 0u;
        }
        tile[r][tx] = k;
    }
    __syncthreads();
    uint32_t *dst = keys + (int64_t)f * C * B;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r, b = b0 + tx;
        if (c < C && b < B) dst[(int64_t)c * B + b] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// K4: the greedy walk.  One WAVE per (frame, class): candidates in descending score order, a
// "dead" bitmask (suppressed | already kept | not a candidate) in LDS, and for every survivor one
// coalesced read of its adjacency list whose lanes OR their neighbour's bit.  Survivors come out
// already in descending score order.  Latency (one L2 round trip per survivor) is hidden by
// prefetching the lists of the next GRP alive candidates and by 32 resident waves per CU.
// Zero-union rule: a tagged entry (u,v) raises iff v is not dead when u is kept -- exactly the
// pairs the reference evaluates (utils/nms.pyx:52-64).
// ------------------------------------------------------------------------------------------------
struct WalkParams {
    int mode, P, B, C;
    const GroupDesc *groups;
    const uint16_t *order;
    const int32_t *ncand;
    const uint2 *row_meta;         // per box: x = offset of its adjacency list, y = its length
    const uint16_t *adj;
    const uint32_t *group_z;
    int32_t *keep_idx;        // mode 0/1: [P,cap]; mode 2: flat [Ntot] at box_off (cap = nbox)
    int32_t *keep_cnt;        // [P]
    int64_t cap;
    int *status;
    int mask_words;           // u32 words of one wave's dead mask
    const uint32_t *group_flags;   // kFlagRegular per group (the K1s path ran), or null
    int wave_words;           // u32 words of LDS per wave (mask + the packed walk's ring)
    int packed;               // regular frames take walk_list_packed (eight candidates per pass)
    const WalkMeta *wmeta;    // per box: coordinates + list (adj_build_kernel), and the graph's threshold: the packed walk
    float t32;                // tests the members of a group against each other geometrically
    int adj32;                // the adjacency pool is smaller than 4 GB (byte offsets of its lists fit 32 bits)
    const uint4 *wmeta16;     // frames of integer coordinates (kFlagU16) whose records adj_rows_kernel wrote: the same record in 16
                              // bytes {x1 | y1 << 16, x2 | y2 << 16, list offset, degree} (null: none)
};

// LDS words through which the lanes of one wave talk to each other (the walks' dead masks): every
// access must be a real LDS instruction.  A `volatile` access through a GENERIC pointer is not
// rewritten to the LDS address space by hipcc (InferAddressSpaces skips volatile accesses): it
// becomes flat_load ... sc0 sc1 + s_waitcnt vmcnt(0), which drains every global load in flight --
// measured: it serialised the walk's whole prefetch pipeline.  So the masks are typed
// address_space(3): ds_read_b32 / ds_write_b32 / ds_or_b32, ordered by the wave's in-order LDS queue.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef volatile lds_u32_t *lds_mask_t;

__device__ __forceinline__ lds_mask_t lds_mask_ptr(unsigned char *dyn_smem, int word_off)
{
    return (lds_mask_t)((__attribute__((address_space(3))) unsigned char *)dyn_smem) + word_off;
}
__device__ __forceinline__ void lds_or(lds_mask_t m, int word, uint32_t bits)
{
    __hip_atomic_fetch_or((lds_u32_t *)(m + word), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void lds_and(lds_mask_t m, int word, uint32_t bits)
{
    __hip_atomic_fetch_and((lds_u32_t *)(m + word), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// Two adjacency entries (one dword) -> two dead bits.  The word index comes from v_bfe_u32 (hipcc
// otherwise rewrites (e & 0xFFFF) >> 5 << 2 as shift + mask and needs a separate add for the mask's LDS base; bit-field
// extract + v_lshl_add_u32 is one instruction less per entry in the walk's hottest sequence), the bit from a shift that
// only looks at the operand's low five bits.
__device__ __forceinline__ void lds_or_pair(lds_mask_t m, uint32_t d)
{
    uint32_t w0, w1;
    asm("v_bfe_u32 %0, %1, 5, 11" : "=v"(w0) : "v"(d));      // (inline asm: the builtin is folded back into shift + mask)
    asm("v_bfe_u32 %0, %1, 21, 11" : "=v"(w1) : "v"(d));
    lds_or(m, (int)w0, 1u << (d & 31u));
    lds_or(m, (int)w1, 1u << ((d >> 16) & 31u));
}

constexpr int kWalkGrp = 4;

// One 64-entry slice of a survivor's adjacency list -> dead bits.  HASZ = false (every regular
// frame): entries are plain indices, nothing to test -- this is the instruction-issue hot spot of
// the walk (one wave does ~1 400 survivors x 2 slices per problem), so it is kept to the bone.
template <bool HASZ>
__device__ __forceinline__ void walk_apply_slice(lds_mask_t mask, uint16_t e, bool in_range, int &bad)
{
    if (HASZ) {
        if (in_range) {
            const int v = e & 0x7FFF;
            if (e & kZTag) { if (!((mask[v >> 5] >> (v & 31)) & 1u)) bad = 1; }
            else lds_or(mask, v >> 5, 1u << (v & 31));
        }
    } else {
        // (exec-masked on purpose: letting the lanes past the end OR 0 into their clamped word makes
        // them all hit ONE address, and same-address LDS atomics serialise -- measured 6x slower)
        if (in_range) lds_or(mask, e >> 5, 1u << (e & 31));
    }
}

// the survivors of one group of up to kWalkGrp alive candidates (lanes ls[0..ng) of c / off / deg).
// pre[k] = entries 2*lane and 2*lane+1 of candidate k's adjacency list (one 4-byte load; lists start
// at even offsets and are padded to even length with a duplicate entry, so a lane's two entries are
// valid together).  A survivor is recorded by setting its lane's bit in the SCALAR mask kept_lanes
// (the caller writes the chunk's survivors with one compacting store): the walk is instruction-issue
// bound (~63 % of a list's time is this loop, measured), every instruction per survivor counts.
template <bool HASZ>
__device__ __forceinline__ void walk_group(lds_mask_t mask, const uint16_t *__restrict__ adj, int lane, int c,
                                           uint32_t off, int deg, const int (&ls)[kWalkGrp], int ng,
                                           const uint32_t (&pre)[kWalkGrp], unsigned long long &kept_lanes, int &bad)
{
#pragma unroll
    for (int k = 0; k < kWalkGrp; ++k) {
        if (k >= ng) break;
        const int cu = __builtin_amdgcn_readlane(c, ls[k]);
        // (wave-uniform value made scalar: scalar branch)
        if (__builtin_amdgcn_readfirstlane((mask[cu >> 5] >> (cu & 31)) & 1u)) continue;   // suppressed by an earlier survivor of this chunk
        kept_lanes |= 1ull << ls[k];
        const int d = __builtin_amdgcn_readlane(deg, ls[k]);
        if (HASZ && lane == 0) lds_or(mask, cu >> 5, 1u << (cu & 31));   // a kept box reads as dead (zero-union rule)
        if (2 * lane < d) {
            walk_apply_slice<HASZ>(mask, (uint16_t)(pre[k] & 0xFFFFu), true, bad);
            walk_apply_slice<HASZ>(mask, (uint16_t)(pre[k] >> 16), true, bad);
        }
        if (d > 128) {   // rare: long lists
            const uint32_t o = __builtin_amdgcn_readlane(off, ls[k]);
            for (int e0 = 128; e0 < d; e0 += 64)
                walk_apply_slice<HASZ>(mask, adj[o + min(e0 + lane, d - 1)], e0 + lane < d, bad);
        }
    }
}


// ------------------------------------------------------------------------------------------------
// The PACKED walk of a REGULAR frame's list (finite boxes, positive areas; graph built by iou_bits_sym_kernel: symmetric,
// no zero-union tags).  The general walk (walk_group, below in walk_one) spends ~50 instructions per survivor, half of
// them scalar (one survivor at a time: broadcast its id, look at its bit, select its length, mask the lanes), and
// the scalar unit is its bottleneck.  Here EIGHT alive candidates are handled by one pass of vector code, 8 lanes
// each: a lane loads 16 entries (32 B) of its member's adjacency list -- once the member is known to survive -- and ORs
// them into the dead mask.  What makes
// that legal is that the order inside a group only matters for members that suppress EACH OTHER, and that is a
// property of their two boxes: lane (i, j) of the 8 x 8 lane grid evaluates the exact pair predicate of members
// i < j (the graph's own edge rule), one ballot yields the group's conflict matrix, and only if it is non-zero
// (rare: a few % of the groups) a short scalar loop decides who survives.  Everything else is the same greedy
// rule: a member is alive iff its bit is clear when its group starts (all earlier groups are in the mask) and no
// earlier SURVIVING member of its group suppresses it.
// Alive candidates are queued in a small LDS ring (with their boxes and lists: one 32-byte record per candidate, fetched
// one chunk ahead) as the chunks of 64 candidates are filtered, so every group but the last is full; a candidate filtered "alive" while earlier ones were still queued is simply found dead when its
// group starts.  Lists longer than 128 entries finish in a (rare) per-member loop.
// ------------------------------------------------------------------------------------------------
constexpr int kPackRing = 72;                  // ring slots of 32 B: {box index, list offset, length, -, x1 y1 x2 y2}
struct __attribute__((aligned(16))) AdjVec { uint32_t v[4]; };    // 8 entries (lists are 16-byte aligned)
typedef float lds_f4v __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) lds_f4v *lds_f4_t;

__device__ __forceinline__ int ring_wrap(int s) { return s >= kPackRing ? s - kPackRing : s; }

// The groups of the packed walks: while the ring holds a full group of eight alive candidates (or, at the end of the list,
// anything at all), decide the group's survivors and OR their adjacency lists into the dead mask.  qh / qn = ring head and
// fill, nk = survivors so far (all wave-uniform).
__device__ __forceinline__ void walk_ring_drain(const WalkParams &prm, lds_mask_t mask, lds_mask_t ring, lds_f4_t ringb, const int lane,
                                                const float t32, const bool flush, int &qh, int &qn, int &nk,
                                                int32_t *__restrict__ out, const int64_t cap)
{
    const int k = lane >> 3, sub = lane & 7;
    while (qn >= 8 || (flush && qn > 0)) {
        const int ng = min(8, qn);
        const int s = ring_wrap(qh + k);
        const bool vk = k < ng;
        // (the slot's four meta words -- candidate, list offset, length, area -- in ONE 16-byte LDS read; the areas were formed
        //  once when the candidate was queued instead of twice per lane and group here)
        const int sj = ring_wrap(qh + sub);
        const lds_f4v mi = ringb[2 * s];
        const int cm = vk ? (int)__float_as_uint(mi.x) : 0;
        // who survives: alive when the group starts, and not suppressed by an earlier surviving member (boxes from the ring)
        const lds_f4v vi = ringb[2 * s + 1], vj = ringb[2 * sj + 1];
        const float4 bi = make_float4(vi.x, vi.y, vi.z, vi.w), bj = make_float4(vj.x, vj.y, vj.z, vj.w);
        const bool live = vk && !((mask[cm >> 5] >> (cm & 31)) & 1u);
        const unsigned long long lm = __ballot(live);
        // the graph's own edge rule without the division (regular frames only get here): the sign of fma(-t32, union, inter)
        // decides, the band just below the threshold falls back to the IEEE quotient (pred_regular, as in iou_bits_sym_kernel)
        const float ai = mi.w, aj = __uint_as_float(ring[8 * sj + 3]);
        bool border;
        bool hit = pred_regular(bi, ai, bj, aj, t32, t32 * 4.76837158203125e-7f, border);
        if (__ballot(border) != 0ull) {
            if (border) hit = (pair_pred(bi, ai, bj, aj, t32) & 1u) != 0u;
        }
        const unsigned long long cmask = __ballot(hit && k < sub && sub < ng && live && ((lm >> (8 * sub)) & 1ull));
        unsigned long long surv_s = lm & 0x0101010101010101ull;      // bit 8k <=> member k survives
        if (cmask) {
            unsigned long long sv = 0ull;
            for (int j = 0; j < 8; ++j) {
                const unsigned long long col = (cmask >> j) & 0x0101010101010101ull;   // bit 8i <=> i suppresses j
                if (((lm >> (8 * j)) & 1ull) && !(col & sv)) sv |= 1ull << (8 * j);
            }
            surv_s = sv;
        }
        const bool surv = (surv_s >> (lane & 56)) & 1ull;
        if (surv_s) {
            const int pos = nk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(surv_s >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)surv_s, 0u));
            if (surv && sub == 0 && (int64_t)pos < cap) out[(uint32_t)pos] = cm;
            nk += __popcll(surv_s);
            // the survivors' lists: lane `sub` owns entries [8 sub, 8 sub + 8) and [64 + 8 sub, 64 + 8 sub + 8) -- 8 lanes
            // read 128 contiguous, aligned bytes per load; a lane whose share lies past the list's end re-reads the
            // list's first 16 bytes (no extra cache line, no divergent load); only entries that exist go to the LDS
            const uint32_t off = surv ? __float_as_uint(mi.y) : 0u;
            const int deg = surv ? (int)__float_as_uint(mi.z) : 0;
            AdjVec a0, a1;
            if (prm.adj32) {
                // (a pool of < 4 GB: 32-bit byte offsets next to the scalar base -- two 64-bit address computations less per
                //  lane and group; A/B on one box: 2.79 -> 2.745 ms)
                const uint32_t bo = off * 2u;
                const char *ab = reinterpret_cast<const char *>(prm.adj);
                a0 = *reinterpret_cast<const AdjVec *>(ab + (bo + (8 * sub < deg ? 16u * (uint32_t)sub : 0u)));
                a1 = *reinterpret_cast<const AdjVec *>(ab + (bo + (64 + 8 * sub < deg ? 128u + 16u * (uint32_t)sub : 0u)));
            } else {
                const AdjVec *pa = reinterpret_cast<const AdjVec *>(prm.adj + off);
                a0 = pa[8 * sub < deg ? sub : 0];
                a1 = pa[64 + 8 * sub < deg ? 8 + sub : 0];
            }
            // (a piece that starts inside the list is applied whole: K2 pads every list to a multiple of 8 entries with
            //  copies of its last entry -- one guard per piece instead of one per entry)
            const bool has0 = 8 * sub < deg, has1 = 64 + 8 * sub < deg;
            if (has0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t d = a0.v[t];
                    lds_or_pair(mask, d);
                }
            }
            if (__ballot(has1) != 0ull) {
                if (has1) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t d = a1.v[t];
                        lds_or_pair(mask, d);
                    }
                }
                unsigned long long lg = __ballot(sub == 0 && deg > 128);
                while (lg) {                                             // rare: long lists
                    const int l = __ffsll((unsigned long long)lg) - 1;
                    lg &= lg - 1;
                    const uint32_t o = __builtin_amdgcn_readlane(off, l);
                    const int dl = __builtin_amdgcn_readlane(deg, l);
                    for (int e0 = 128; e0 < dl; e0 += 64) {
                        const uint32_t e = prm.adj[o + min(e0 + lane, dl - 1)];
                        if (e0 + lane < dl) lds_or(mask, (int)(e >> 5), 1u << (e & 31u));
                    }
                }
            }
            // every list load of this turn has landed before the next turn starts (vmcnt(0) only).  Without it the loads of a
            // turn whose lanes all skipped their pieces are still pending on the back edge, and hipcc then waits for vmcnt(0) at
            // the loop HEADER -- where it also drains the next chunk's prefetched records before this chunk's first lists can
            // even be requested (one exposed round trip per chunk)
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        qh = ring_wrap(qh + ng);
        qn -= ng;
    }
}

// U16: the records come from wmeta16 -- ONE 16-byte load per candidate and chunk instead of two.  The walk's busiest unit is the
// address unit of the L1 (TA_BUSY 71 % of the kernel's cycles, profiles/r06_pmc_tcp.csv), and the record prefetch -- issued for all
// 64 lanes of every one of the 157 chunks -- is ~40 % of the bytes it generates addresses for.  The raw words stay in registers
// until the candidate is queued (a conversion at load time would wait for the load on the spot).
template <bool U16>
__device__ __forceinline__ void walk_list_packed(const WalkParams &prm, lds_mask_t mask, const int lane, const int rb,
                                                 const uint16_t *__restrict__ order, const int ncand,
                                                 int32_t *__restrict__ out, const int64_t cap, int &nk_out)
{
    lds_mask_t ring = mask + prm.mask_words;
    lds_f4_t ringb = (lds_f4_t)ring;                              // slot s: words 8s .. 8s+3 = meta, float4 2s+1 = box
    const WalkMeta *__restrict__ wmeta = prm.wmeta + rb;
    const uint4 *__restrict__ wm16 = U16 ? prm.wmeta16 + rb : nullptr;
    const float t32 = prm.t32;
    int qh = 0, qn = 0, nk = 0;                                   // ring head, queued candidates, survivors (wave-uniform)
    const int last = max(ncand - 1, 0);
    // Software pipeline over the chunks of 64 candidates, three register sets taken in turn (the loop is unrolled by three):
    // chunk k's pass uses set k % 3, requests the records of chunk k + 1 (for the lanes alive NOW) into set (k + 1) % 3 and
    // the candidate ids of chunk k + 2 into set (k + 2) % 3.  With ONE set of names rotated by register copies at the bottom of
    // the loop (round 2/3) hipcc has to wait for every load in flight right there -- a copy reads its source -- so the
    // "prefetch" was drained once per chunk; a chunk that queues fewer than eight candidates (no lists to fetch, nothing to
    // hide behind) then paid the records' whole round trip (found in the ISA, round 4: s_waitcnt vmcnt(0) in front of the copies).
    int cc[3];
    uint2 mm[3];
    float4 bb[3];
    uint4 rr[3];                                                  // U16: the raw 16-byte records
    cc[0] = (int)order[(uint32_t)min(lane, last)];
    cc[1] = (int)order[(uint32_t)min(64 + lane, last)];
    cc[2] = 0;
    mm[0] = mm[1] = mm[2] = make_uint2(0u, 0u);
    bb[0] = bb[1] = bb[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    rr[0] = rr[1] = rr[2] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < ncand) {                                           // chunk 0: everything is alive
        if (U16) rr[0] = wm16[(uint32_t)cc[0]];
        else { const WalkMeta *wm = wmeta + (uint32_t)cc[0]; bb[0] = wm->box; const uint4 rw = wm->row; mm[0] = make_uint2(rw.x, rw.y); }
    }
#define VDET_WALK_PASS(Q0, CUR, NXT, NN)                                                                                             \
    {                                                                                                                                \
        const int q0_ = (Q0);                                                                                                        \
        const int c = cc[CUR];                                                                                                       \
        cc[NN] = (int)order[(uint32_t)min(q0_ + 128 + lane, last)];                                                                  \
        {   /* (no branch around the loads: a lane that needs no record reads record 0 -- one shared line -- so that the number */ \
            /*  of loads in flight does not depend on the path and hipcc can wait for exactly the older ones)                   */ \
            const bool want = (q0_ + 64 + lane) < ncand && !((mask[cc[NXT] >> 5] >> (cc[NXT] & 31)) & 1u);                           \
            if (U16) rr[NXT] = wm16[want ? (uint32_t)cc[NXT] : 0u];                                                                  \
            else {                                                                                                                   \
                const WalkMeta *wm = wmeta + (want ? (uint32_t)cc[NXT] : 0u);                                                        \
                bb[NXT] = wm->box; const uint4 rw = wm->row; mm[NXT] = make_uint2(rw.x, rw.y);                                       \
            }                                                                                                                        \
        }                                                                                                                            \
        const bool alive = (q0_ + lane) < ncand && !((mask[c >> 5] >> (c & 31)) & 1u);                                               \
        const unsigned long long am = __ballot(alive);                                                                               \
        if (alive) {                                                                                                                 \
            const int s = ring_wrap(qh + qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u))); \
            ring[8 * s] = (uint32_t)c;                                                                                               \
            lds_f4v bv;                                                                                                              \
            if (U16) {                                                                                                               \
                ring[8 * s + 1] = rr[CUR].z;                                                                                         \
                ring[8 * s + 2] = rr[CUR].w;                                                                                         \
                bv.x = (float)(rr[CUR].x & 0xFFFFu); bv.y = (float)(rr[CUR].x >> 16);                                                \
                bv.z = (float)(rr[CUR].y & 0xFFFFu); bv.w = (float)(rr[CUR].y >> 16);                                                \
            } else {                                                                                                                 \
                ring[8 * s + 1] = mm[CUR].x;                                                                                         \
                ring[8 * s + 2] = mm[CUR].y;                                                                                         \
                bv.x = bb[CUR].x; bv.y = bb[CUR].y; bv.z = bb[CUR].z; bv.w = bb[CUR].w;                                              \
            }                                                                                                                        \
            ringb[2 * s + 1] = bv;                                                                                                   \
            ring[8 * s + 3] = __float_as_uint(box_area(make_float4(bv.x, bv.y, bv.z, bv.w)));                                        \
        }                                                                                                                            \
        qn += __popcll(am);                                                                                                          \
        walk_ring_drain(prm, mask, ring, ringb, lane, t32, q0_ + 64 >= ncand, qh, qn, nk, out, cap);                                 \
    }
    for (int q0 = 0; q0 < ncand; q0 += 192) {
        VDET_WALK_PASS(q0, 0, 1, 2)
        if (q0 + 64 >= ncand) break;
        VDET_WALK_PASS(q0 + 64, 1, 2, 0)
        if (q0 + 128 >= ncand) break;
        VDET_WALK_PASS(q0 + 128, 2, 0, 1)
    }
#undef VDET_WALK_PASS
    nk_out = nk;
}

// one list (problem p) walked by one wave (w = its index in the block: its slice of the dynamic LDS)
__device__ __forceinline__ void walk_one(const WalkParams &prm, const int p, unsigned char *smem, const int lane, const int w)
{
    lds_mask_t mask = lds_mask_ptr(smem, w * prm.wave_words);
    const ProblemRef pr = decode_problem(prm.mode, p, prm.B, prm.C, prm.groups);
    const int N = pr.N, rb = pr.rb;
    const uint16_t *order = prm.order + pr.obase;
    const int ncand = prm.ncand[p];
    const bool has_z = prm.group_z[pr.g] != 0;
    const bool regular = !has_z && prm.group_flags && (prm.group_flags[pr.g] & kFlagRegular);    // (wave-uniform)
    int32_t *out = prm.keep_idx + (prm.mode == 2 ? (int64_t)rb : (int64_t)p * prm.cap);
    const int64_t cap = prm.mode == 2 ? (int64_t)N : prm.cap;

    for (int i = lane; i < ((N + 31) >> 5); i += 64) mask[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (has_z) {   // non-candidates must read as dead for the zero-union rule
        for (int q = ncand + lane; q < N; q += 64) {
            const int v = order[q];
            lds_or(mask, v >> 5, 1u << (v & 31));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    int nk = 0;
    int bad = 0;
    if (regular && prm.packed && N >= 2) {     // (singleton groups have no graph: adj_build_kernel never saw them)
        // (wave-uniform) integer frames whose records exist in 16 bytes take them
        if (prm.wmeta16 && (prm.group_flags[pr.g] & kFlagU16)) walk_list_packed<true>(prm, mask, lane, rb, order, ncand, out, cap, nk);
        else walk_list_packed<false>(prm, mask, lane, rb, order, ncand, out, cap, nk);
        if (lane == 0) prm.keep_cnt[p] = nk;
        if ((int64_t)nk > cap && lane == 0) atomicOr(prm.status, kStCap);
        return;
    }
    const uint32_t *adjw = reinterpret_cast<const uint32_t *>(prm.adj);   // adjacency lists start at even offsets
    // Two-deep software pipeline over the chunks of 64 candidates: the ids of chunk i+2 (one
    // coalesced load) and the row meta of chunk i+1 (an 8-B gather, issued only for the lanes that
    // are still alive NOW -- a dead candidate never revives, so this is a superset of what will be
    // needed) are in flight while chunk i is walked.  Per chunk only the adjacency-list loads of
    // its survivor groups remain on the critical path.  (Also prefetching the next chunk's first
    // group of lists was measured: +11 % time -- the walk is instruction-issue bound, ~35
    // instructions per survivor and ~60 per chunk, not latency-bound; so were two lists per wave
    // (+20 %) and byte flags instead of the bit mask (+40 %: half the occupancy).)
    const int last = max(ncand - 1, 0);
    int c_cur = (int)order[min(lane, last)];
    int c_nxt = (int)order[min(64 + lane, last)];
    uint2 m_cur = make_uint2(0u, 0u);
    if (lane < ncand) m_cur = prm.row_meta[rb + c_cur];       // chunk 0: everything is alive
    for (int q0 = 0; q0 < ncand; q0 += 64) {
        const int q = q0 + lane;
        const bool valid = q < ncand;
        const int c = c_cur;
        const int c_nn = (int)order[min(q0 + 128 + lane, last)];
        uint2 m_nxt = make_uint2(0u, 0u);
        if ((q + 64) < ncand && !((mask[c_nxt >> 5] >> (c_nxt & 31)) & 1u)) m_nxt = prm.row_meta[rb + c_nxt];
        const bool alive = valid && !((mask[c >> 5] >> (c & 31)) & 1u);
        unsigned long long am = __ballot(alive);
        unsigned long long kept_lanes = 0ull;
        const uint32_t off = m_cur.x;
        const int deg = (int)m_cur.y;
        while (am) {
            int ls[kWalkGrp];
            int ng = 0;
#pragma unroll
            for (int k = 0; k < kWalkGrp; ++k) {
                // slots past the last alive candidate repeat the previous one (their loads hit cache)
                ls[k] = k ? ls[k - 1] : 0;
                if (am) { ls[k] = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)am) - 1); am &= am - 1; ng = k + 1; }
            }
            // Prefetch the first 128 entries of every list of the group.  The loads are
            // UNCONDITIONAL (clamped lane index, no exec-masked branch): a guarded load forces hipcc
            // to drain vmcnt at every join and serialises the whole group.
            uint32_t pre[kWalkGrp];
#pragma unroll
            for (int k = 0; k < kWalkGrp; ++k) {
                const uint32_t o = __builtin_amdgcn_readlane(off, ls[k]);
                const int d = __builtin_amdgcn_readlane(deg, ls[k]);
                pre[k] = adjw[(o >> 1) + min(lane, (max(d, 1) - 1) >> 1)];
            }
            if (has_z) walk_group<true>(mask, prm.adj, lane, c, off, deg, ls, ng, pre, kept_lanes, bad);
            else walk_group<false>(mask, prm.adj, lane, c, off, deg, ls, ng, pre, kept_lanes, bad);
        }
        if (kept_lanes) {   // the chunk's survivors, in lane (= descending score) order, with one compacting store
            const int pos = nk + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(kept_lanes >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)kept_lanes, 0u));
            if (((kept_lanes >> lane) & 1ull) && (int64_t)pos < cap) out[pos] = c;
            nk += __popcll(kept_lanes);
        }
        c_cur = c_nxt; c_nxt = c_nn; m_cur = m_nxt;
    }
    if (lane == 0) prm.keep_cnt[p] = nk;
    if ((int64_t)nk > cap && lane == 0) atomicOr(prm.status, kStCap);
    if (__ballot(bad != 0) && lane == 0) atomicOr(prm.status, kStDivZero);
}

__global__ __launch_bounds__(256) void walk_kernel(const WalkParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (the wave index through readfirstlane: hipcc then knows that the problem, its list / frame pointers and every
    //  address base below are wave-uniform -- scalar registers and saddr loads instead of 64-bit vector address math)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves_total = gridDim.x * 4;
    // wave-granular XCD mapping: block b -> XCD b % 8; consecutive problems share a frame
    const int per = nwaves_total >> 3;
    const int p = (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + w;
    if (p >= prm.P) return;
    walk_one(prm, p, smem, lane, w);
}

// Caller-supplied candidate lists (vdet_nms_volume_ordered) are walked as they come: a count above B or an index >= B would
// index the walks' LDS dead mask and the adjacency tables out of bounds.  One wave per list checks it first: a bad list
// latches kStBadOrder and is walked as EMPTY (ncand_out = 0), a good one keeps its count.
__global__ __launch_bounds__(256) void check_order_kernel(const uint16_t *__restrict__ order, const int32_t *__restrict__ ncand_in, int P, int B,
                                                          int32_t *__restrict__ ncand_out, int *__restrict__ status)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int n = ncand_in[p];
    bool bad = n < 0 || n > B;
    if (!bad)
        for (int q = lane; q < n; q += 64) bad = bad || (int)order[(int64_t)p * B + q] >= B;
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) {
        ncand_out[p] = any ? 0 : n;
        if (any) atomicOr(status, kStBadOrder);
    }
}

// mode-2 merge helper: survivors of every group -> composites key << 32 | original index
__global__ __launch_bounds__(256) void gather_comp_kernel(const GroupDesc *__restrict__ groups, int P,
                                                          const int32_t *__restrict__ keep_idx,
                                                          const int32_t *__restrict__ keep_cnt,
                                                          const float *__restrict__ scores,
                                                          const uint32_t *__restrict__ keys,
                                                          const uint32_t *__restrict__ orig_idx,
                                                          unsigned long long *__restrict__ comp,
                                                          unsigned int *__restrict__ glob_cnt)
{
    __shared__ unsigned int sbase;
    const int p = blockIdx.x;
    if (p >= P) return;
    const int rb = groups[p].box_off;
    const int K = keep_cnt[p];
    if (threadIdx.x == 0) sbase = atomicAdd(glob_cnt, (unsigned int)K);
    __syncthreads();
    const unsigned int base = sbase;
    for (int k = threadIdx.x; k < K; k += 256) {
        const int v = keep_idx[rb + k];
        const uint32_t key = keys ? keys[rb + v] : score_key(scores[rb + v]);
        comp[base + k] = ((unsigned long long)key << 32) | orig_idx[rb + v];
    }
}

// ------------------------------------------------------------------------------------------------
// Global descending sort of u64 composites (vid_nms merge of the per-frame survivors).
// n2 = power of two >= n, data padded with 0.
// ------------------------------------------------------------------------------------------------
__global__ void bitonic_global_step(unsigned long long *__restrict__ data, uint32_t n2, uint32_t j, uint32_t k)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const uint32_t ixj = i ^ j;
    if (ixj > i) {
        const unsigned long long a = data[i], b = data[ixj];
        const bool desc = (i & k) == 0;
        if (desc ? (a < b) : (a > b)) { data[i] = b; data[ixj] = a; }
    }
}

// all steps with j < 2048 of stage k (or the complete sort of a <= 2048 block when k_lo..k_hi given)
__global__ __launch_bounds__(1024) void bitonic_lds_kernel(unsigned long long *__restrict__ data, uint32_t n2,
                                                           uint32_t k_first, uint32_t k_last)
{
    __shared__ unsigned long long s[2048];
    const uint32_t base = blockIdx.x * 2048u;
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 2048u; i += 1024u) s[i] = (base + i < n2) ? data[base + i] : 0ull;
    __syncthreads();
    for (uint32_t k = k_first; k <= k_last; k <<= 1) {
        uint32_t j0 = k >> 1;
        if (j0 > 1024u) j0 = 1024u;
        for (uint32_t j = j0; j > 0; j >>= 1) {
            for (uint32_t li = tid; li < 2048u; li += 1024u) {
                const uint32_t lixj = li ^ j;
                if (lixj > li) {
                    const uint32_t gi = base + li;
                    const unsigned long long a = s[li], b = s[lixj];
                    const bool desc = (gi & k) == 0;
                    if (desc ? (a < b) : (a > b)) { s[li] = b; s[lixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = tid; i < 2048u; i += 1024u)
        if (base + i < n2) data[base + i] = s[i];
}

// Large-group fallback of the per-problem argsort (groups that do not fit the in-LDS sort, up to the
// 32 767 boxes a u16 index can address): composites key << 32 | index, sorted by the global bitonic
// network, then unpacked.  Same order as sort_kernel: descending key, ties by descending index;
// non-candidates (key 0) at the tail.
__global__ void fill_comp_kernel(const float *__restrict__ scores, const uint32_t *__restrict__ keys,
                                 const uint8_t *__restrict__ excl, int use_thr, float thr, int64_t base, int n,
                                 uint32_t n2, unsigned long long *__restrict__ comp, int32_t *__restrict__ ncand_p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    unsigned long long c = 0ull;
    if ((int)i < n) {
        uint32_t k;
        bool x = false;
        if (keys) { k = keys[base + i]; x = (k == 0u); }
        else {
            const float sc = scores[base + i];
            k = score_key(sc);
            if (use_thr && !(sc > thr)) x = true;
        }
        if (excl && excl[base + i]) x = true;
        if (x) k = 0u; else atomicAdd(ncand_p, 1);
        c = ((unsigned long long)k << 32) | i;
    }
    comp[i] = c;
}

__global__ void comp_to_order_kernel(const unsigned long long *__restrict__ comp, int n, uint16_t *__restrict__ order)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) order[i] = (uint16_t)(comp[i] & 0xFFFFull);
}

// composites -> int64 indices
__global__ void comp_to_index_kernel(const unsigned long long *__restrict__ comp, uint32_t n, int64_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)(comp[i] & 0xFFFFFFFFull);
}

// ------------------------------------------------------------------------------------------------
// track_det_nms round 1 (utils/nms.pyx:163-183): det i (as the "i" box) against the same-frame
// tracks in order; stops at the first suppression, raises on an evaluated zero union.
// dets: rows (frame, x1,y1,x2,y2) packed as frame[] + float4 boxes[]; excl[i] = suppressed.
// ------------------------------------------------------------------------------------------------
__global__ void track_round1_kernel(const float *__restrict__ det_frame, const float4 *__restrict__ det_box, int m,
                                    const float *__restrict__ trk_frame, const float4 *__restrict__ trk_box, int t,
                                    float t32, uint8_t *__restrict__ excl, int *__restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const float4 bi = det_box[i];
    const float iarea = box_area(bi);
    const float fi = det_frame[i];
    uint8_t sup = 0;
    for (int j = 0; j < t; ++j) {
        if (fi != trk_frame[j]) continue;
        const float4 bt = trk_box[j];
        // roles: the DET is box "i", the track is box "j" (iarea + t_areas[j] - inter), but
        // max/min take (det, track) in that order: xx1 = max(ix1, t_x1[j])
        const uint32_t p = pair_pred(bi, iarea, bt, box_area(bt), t32);
        if (p & 2u) { atomicOr(status, kStDivZero); break; }
        if (p & 1u) { sup = 1; break; }
    }
    excl[i] = sup;
}

}  // namespace vdet
