// nms_kernels.hpp -- gfx950 kernels for greedy NMS (utils/nms.pyx:17-189 of the reference).
//
// MI355X-first formulation (not the reference's sequential double loop):
//
//   The suppression predicate  IoU_f32(i, j) >= thresh  depends only on the two boxes, not on the
//   class whose scores order them.  So per FRAME (geometry group) we build the suppression graph
//   once, and every (frame, class) problem is a lexicographically-first maximal independent set on
//   that shared graph under its own priority order -- which is exactly what the reference's greedy
//   loop computes (box v is kept iff no higher-priority neighbour is kept).
//
//   K1 iou_bits_kernel   all-pairs predicate of one frame, one lane per row, column boxes broadcast
//                        from LDS; 64 predicates packed per u64, written transposed ([word][row]) so
//                        the stores and the later loads are coalesced.  VALU-bound.
//   K2 adj_build_kernel  bit rows -> compact u16 adjacency lists (CSR slabs allocated with one
//                        atomicAdd per 256-row tile).
//   K3 mis_kernel        one workgroup per (frame, class): priorities (sortable score keys) in LDS,
//                        rounds of "decide every vertex whose higher-priority neighbours are all
//                        decided" (deterministic parallel greedy MIS), then an in-LDS bitonic sort
//                        of the survivors into descending-score order.  LDS/latency-bound.
//
//   No sort of the B candidates is needed at all (only the ~K survivors are sorted), and the
//   O(B^2) float work is shared by all C classes.
//
// Exactness notes (all verified against the oracle / golden vectors in tests/):
//   * the predicate reproduces utils/nms.pyx:57-65 operation by operation in f32 (TU is compiled
//     with -ffp-contract=off; '/' is the correctly rounded IEEE division), with the reference's
//     "a if a >= b else b" max/min (NaN: second operand wins) and explicit i/j roles, so the graph
//     is stored as IN-lists of j (who can suppress me) and stays exact for NaN coordinates;
//   * "ovr >= thresh" is an f64 compare of the promoted f32 quotient (thresh is a boxed python
//     float); t32 = min{f in f32 : (double)f >= thresh} makes  ovr >= t32  the same predicate;
//   * a zero union raises ZeroDivisionError in the reference (Cython cdivision=False) only for
//     pairs it actually evaluates; zero-union pairs are kept as TAGGED adjacency entries and the
//     evaluated-pair rule is re-checked after the MIS converged (mis_kernel epilogue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdet {

struct GroupDesc {
    int32_t box_off;   // first row of this group in the flat (grouped) box array
    int32_t nbox;      // boxes in the group (<= 32767)
    int64_t bits_off;  // u64-word offset of the group's [W][nbox] bit matrix inside the batch scratch
};

struct TileDesc {      // one 256-row tile of one group
    int32_t group;
    int32_t row_tile;
};

constexpr int kRowsPerTile = 256;
constexpr uint16_t kZTag = 0x8000;

// status bits latched by kernels into ctx->d_status
constexpr int kStCap = 1;       // survivors > cap
constexpr int kStDivZero = 2;   // evaluated zero-union pair
constexpr int kStPool = 4;      // adjacency pool too small (internal, retried by the host)

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
// bits[g.bits_off + w*nbox + v] bit k  <=>  box u = 64*w+k (as i) suppresses box v (as j), u != v.
// row_z[flat v] += number of zero-union partners of v.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iou_bits_kernel(const float4 *__restrict__ boxes,
                                                       const GroupDesc *__restrict__ groups,
                                                       const TileDesc *__restrict__ tiles, float t32,
                                                       uint64_t *__restrict__ bits,
                                                       uint32_t *__restrict__ row_z,
                                                       uint32_t *__restrict__ group_z)
{
    __shared__ float4 sbox[256];
    __shared__ float sarea[256];
    const TileDesc td = tiles[blockIdx.x];
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
    float4 bj = make_float4(qnan, qnan, qnan, qnan);
    if (v < B) bj = boxes[gd.box_off + v];
    const float jarea = box_area(bj);
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
                const uint32_t p = pair_pred(sbox[q * 64 + k], sarea[q * 64 + k], bj, jarea, t32);
                lo |= (p & 1u) << k;
                zcnt += p >> 1;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const uint32_t p = pair_pred(sbox[q * 64 + 32 + k], sarea[q * 64 + 32 + k], bj, jarea, t32);
                hi |= (p & 1u) << k;
                zcnt += p >> 1;
            }
            uint64_t m = ((uint64_t)hi << 32) | lo;
            const int cbase = (wt + q) * 64;
            if (v >= cbase && v < cbase + 64) {
                m &= ~(1ull << (v - cbase));                       // no self edge
                const uint32_t ps = pair_pred(bj, jarea, bj, jarea, t32);
                zcnt -= ps >> 1;                                   // ... and no self zero-union
            }
            if (v < B) bits[gd.bits_off + (int64_t)(wt + q) * B + v] = m;
        }
    }
    if (v < B && zcnt) {
        atomicAdd(&row_z[gd.box_off + v], zcnt);
        group_z[td.group] = 1u;
    }
}

// ------------------------------------------------------------------------------------------------
// K2: bit rows -> adjacency lists.  grid = n_tiles; block = 256 (one lane per row).
// Each tile reserves one contiguous slab of the u16 pool with a single atomicAdd; the slab layout
// is irrelevant to the result (lists are sets).  Zero-union partners are appended with kZTag.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adj_build_kernel(const float4 *__restrict__ boxes,
                                                        const GroupDesc *__restrict__ groups,
                                                        const TileDesc *__restrict__ tiles,
                                                        const uint64_t *__restrict__ bits,
                                                        const uint32_t *__restrict__ row_z,
                                                        uint32_t *__restrict__ row_off,
                                                        uint16_t *__restrict__ row_deg,
                                                        uint16_t *__restrict__ adj,
                                                        unsigned long long *__restrict__ pool_used,
                                                        unsigned long long pool_cap, int *__restrict__ status)
{
    __shared__ uint32_t sscan[256];
    __shared__ unsigned long long sbase;
    const TileDesc td = tiles[blockIdx.x];
    const GroupDesc gd = groups[td.group];
    const int B = gd.nbox;
    const int tid = threadIdx.x;
    const int v = td.row_tile * kRowsPerTile + tid;
    const int W = (B + 63) >> 6;
    const uint64_t *col = bits + gd.bits_off + v;

    uint32_t deg = 0, zc = 0;
    if (v < B) {
        for (int w = 0; w < W; ++w) deg += __popcll(col[(int64_t)w * B]);
        zc = row_z[gd.box_off + v];
    }
    const uint32_t tot = deg + zc;
    // block exclusive scan (Hillis-Steele over 256 entries)
    sscan[tid] = tot;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t t = (tid >= d) ? sscan[tid - d] : 0u;
        __syncthreads();
        sscan[tid] += t;
        __syncthreads();
    }
    const uint32_t incl = sscan[tid];
    if (tid == 255) sbase = atomicAdd(pool_used, (unsigned long long)incl);
    __syncthreads();
    const unsigned long long base = sbase;
    const uint32_t tile_total = sscan[255];
    if (v >= B) return;
    if (base + tile_total > pool_cap || base + tile_total > 0xFFFFFFFFull) {
        if (tid == 0) atomicOr(status, kStPool);
        row_off[gd.box_off + v] = 0;
        row_deg[gd.box_off + v] = 0;
        return;
    }
    uint32_t p = (uint32_t)base + (incl - tot);
    row_off[gd.box_off + v] = p;
    row_deg[gd.box_off + v] = (uint16_t)tot;
    for (int w = 0; w < W; ++w) {
        uint64_t m = col[(int64_t)w * B];
        while (m) {
            const int k = __ffsll((unsigned long long)m) - 1;
            adj[p++] = (uint16_t)(w * 64 + k);
            m &= m - 1;
        }
    }
    if (zc) {  // rare: degenerate boxes.  Recompute which partners have a zero union.
        const float4 bj = boxes[gd.box_off + v];
        const float jarea = box_area(bj);
        for (int u = 0; u < B; ++u) {
            if (u == v) continue;
            const float4 bi = boxes[gd.box_off + u];
            if (pair_pred(bi, box_area(bi), bj, jarea, 0.0f) & 2u) adj[p++] = (uint16_t)u | kZTag;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3: per-problem parallel greedy MIS + descending sort of the survivors.
// ------------------------------------------------------------------------------------------------
enum : uint8_t { ST_U = 0, ST_K = 1, ST_S = 2, ST_X = 3 };  // undecided / kept / suppressed / not a candidate

struct MisParams {
    // problem -> (group, score vector)
    int mode;                 // 0: volume [F,B,C]  1: volume [F,C,B]  2: flat, problem p == group p
    int P;                    // number of problems
    int B, C;                 // volume dims (mode 0/1)
    const float *scores;      // mode 0/1: the volume; mode 2: flat [Ntot] (may be null if keys given)
    const uint32_t *keys;     // mode 2 only: explicit priorities (caller-supplied order), else null
    const uint8_t *excl;      // mode 2 only: flat [Ntot], nonzero = not a candidate (track_det_nms round 1)
    int use_thr;              // candidates are score > thr  (vdet/video_det.py:90)
    float thr;
    const GroupDesc *groups;
    const uint32_t *row_off;
    const uint16_t *row_deg;
    const uint16_t *adj;
    const uint32_t *group_z;
    // per-problem sorted output
    int32_t *keep_idx;        // [P, cap] or null
    int32_t *keep_cnt;        // [P] or null
    int64_t cap;
    // global append output (vid_nms merge): composite = key << 32 | orig_idx
    unsigned long long *glob_comp;
    unsigned int *glob_cnt;
    const uint32_t *orig_idx; // flat [Ntot]
    int *status;
    // dynamic-LDS carve (byte offsets, all multiples of 16):
    //   [0, 4*nmax) keys | state | cursor (aliased by the sort buffer after the rounds) | scan[BLOCK]
    int lds_state_off, lds_cursor_off, lds_scan_off;
};

__device__ __forceinline__ bool prio_higher(uint32_t ku, int u, uint32_t kv, int v)
{
    return ku > kv || (ku == kv && u > v);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void mis_kernel(const MisParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);
    uint8_t *state = smem + prm.lds_state_off;
    uint16_t *cursor = reinterpret_cast<uint16_t *>(smem + prm.lds_cursor_off);
    unsigned long long *comp = reinterpret_cast<unsigned long long *>(smem + prm.lds_cursor_off);  // aliases cursor
    uint32_t *sred = reinterpret_cast<uint32_t *>(smem + prm.lds_scan_off);   // no static LDS: keeps the
                                                                              // dynamic base 16-B aligned

    const int tid = threadIdx.x;
    // XCD-aware problem order: the dispatcher places block b on XCD b % 8; give each XCD a
    // contiguous run of problems so the classes of one frame (which share that frame's adjacency
    // lists and score cache lines) meet in the same L2.  Placement only affects speed.
    int p;
    {
        const int nb = gridDim.x;
        const int per = (nb + 7) >> 3;
        p = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (p >= prm.P) return;   // (grid is padded to a multiple of 8)
    }
    int g, N;
    int64_t sbase, sstride;
    if (prm.mode == 0) { g = p / prm.C; const int c = p - g * prm.C; sbase = (int64_t)g * prm.B * prm.C + c; sstride = prm.C; }
    else if (prm.mode == 1) { g = p / prm.C; sbase = (int64_t)p * prm.B; sstride = 1; }
    else { g = p; sbase = prm.groups[g].box_off; sstride = 1; }
    const GroupDesc gd = prm.groups[g];
    N = gd.nbox;
    const int rb = gd.box_off;

    // ---- load priorities / candidate mask
    for (int v = tid; v < N; v += BLOCK) {
        uint32_t k;
        uint8_t st = ST_U;
        if (prm.keys) {
            k = prm.keys[sbase + v];
        } else {
            const float s = prm.scores[sbase + (int64_t)v * sstride];
            k = score_key(s);
            if (prm.use_thr && !(s > prm.thr)) st = ST_X;
        }
        if (prm.excl && prm.excl[rb + v]) st = ST_X;
        keys[v] = k;
        state[v] = st;
        cursor[v] = 0;
    }
    __syncthreads();

    // ---- rounds: a vertex is decided as soon as all its higher-priority in-neighbours are
    for (;;) {
        int pending = 0;
        for (int v = tid; v < N; v += BLOCK) {
            if (state[v] != ST_U) continue;
            const uint32_t off = prm.row_off[rb + v];
            const int d = prm.row_deg[rb + v];
            int c = cursor[v];
            const uint32_t kv = keys[v];
            uint8_t ns = ST_U;
            while (c < d) {
                const uint16_t e = prm.adj[off + c];
                if (e & kZTag) { ++c; continue; }
                const int u = e;
                const uint32_t ku = keys[u];
                if (!prio_higher(ku, u, kv, v)) { ++c; continue; }
                const uint8_t su = state[u];
                if (su == ST_K) { ns = ST_S; break; }
                if (su == ST_U) break;      // wait for u
                ++c;                        // u suppressed or not a candidate
            }
            if (ns == ST_S) state[v] = ST_S;
            else if (c >= d) state[v] = ST_K;
            else { cursor[v] = (uint16_t)c; pending = 1; }
        }
        if (!__syncthreads_or(pending)) break;
    }

    // ---- zero-union rule (only groups holding degenerate boxes): the reference raises iff it
    // EVALUATES a zero-union pair (i kept, j later, j not yet suppressed when i is processed).
    if (prm.group_z[g]) {
        int bad = 0;
        for (int v = tid; v < N; v += BLOCK) {
            if (state[v] == ST_X) continue;
            const uint32_t off = prm.row_off[rb + v];
            const int d = prm.row_deg[rb + v];
            const uint32_t kv = keys[v];
            for (int c = 0; c < d; ++c) {
                const uint16_t e = prm.adj[off + c];
                if (!(e & kZTag)) continue;
                const int u = e & 0x7FFF;
                if (state[u] != ST_K || !prio_higher(keys[u], u, kv, v)) continue;
                bool earlier = false;   // was v already suppressed by a kept box processed before u?
                for (int c2 = 0; c2 < d && !earlier; ++c2) {
                    const uint16_t e2 = prm.adj[off + c2];
                    if (e2 & kZTag) continue;
                    const int s = e2;
                    if (state[s] == ST_K && prio_higher(keys[s], s, keys[u], u)) earlier = true;
                }
                if (!earlier) bad = 1;
            }
        }
        if (bad) atomicOr(prm.status, kStDivZero);
    }

    // ---- compact survivors: composite = key << 32 | index  (unique => any sorting network works)
    uint32_t mine = 0;
    for (int v = tid; v < N; v += BLOCK) mine += (state[v] == ST_K);
    sred[tid] = mine;
    __syncthreads();
    for (int d = 1; d < BLOCK; d <<= 1) {
        const uint32_t t = (tid >= d) ? sred[tid - d] : 0u;
        __syncthreads();
        sred[tid] += t;
        __syncthreads();
    }
    const uint32_t K = sred[BLOCK - 1];
    uint32_t pos = sred[tid] - mine;
    __syncthreads();

    if (prm.glob_comp) {   // unsorted append; the caller sorts globally
        if (tid == 0) sred[0] = atomicAdd(prm.glob_cnt, K);
        __syncthreads();
        const uint32_t gbase = sred[0];
        for (int v = tid; v < N; v += BLOCK)
            if (state[v] == ST_K)
                prm.glob_comp[gbase + pos++] = ((unsigned long long)keys[v] << 32) | prm.orig_idx[rb + v];
        return;
    }

    if (tid == 0) prm.keep_cnt[p] = (int32_t)K;
    if ((int64_t)K > prm.cap) {
        if (tid == 0) atomicOr(prm.status, kStCap);
        return;
    }
    if (K == 0) return;
    uint32_t n2 = 1;
    while (n2 < K) n2 <<= 1;
    // comp[] aliases cursor[] (dead since the last round's barrier); keys/state sit below it.
    for (int v = tid; v < N; v += BLOCK)
        if (state[v] == ST_K) comp[pos++] = ((unsigned long long)keys[v] << 32) | (uint32_t)v;
    for (uint32_t i = K + tid; i < n2; i += BLOCK) comp[i] = 0ull;   // real composites are > 0
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = tid; i < n2; i += BLOCK) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = comp[i], b = comp[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { comp[i] = b; comp[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    int32_t *out = prm.keep_idx + (int64_t)p * prm.cap;
    for (uint32_t i = tid; i < K; i += BLOCK) out[i] = (int32_t)(comp[i] & 0xFFFFFFFFull);
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
