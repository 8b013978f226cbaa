// fused_kernels.hpp -- the drop-in calls of utils/cython_nms on SMALL inputs in ONE launch (round 5).
//
// T-CNN calls the reference's three entry points thousands of times on a few hundred rows each: apply_image_nms once per
// (frame, class) on <= max_per_image rows (vdet/image_det.py:117-123), track_det_nms once per tracked box on the still-kept
// detections of one frame (vdet/track.py:238-249).  At that size a call is latency, not work: the general path of
// vdet_nms_f32 (group by frame -> K0 / index / K1s / K1 / K2 -> sort -> walk -> merge, three host waits) costs several
// hundred microseconds against the reference's 0.2-1.0 ms per call on a CPU core.  Up to kFusedMax rows therefore take this
// file: ONE workgroup runs the reference's algorithm end to end with everything in LDS --
//   rows (read straight from host-mapped memory: no staging copy) -> [track_det_nms round 1, utils/nms.pyx:163-183]
//   -> composite keys, bitonic sort (descending score, ties by descending index: the build's rule; or the caller's order)
//   -> the upper triangle of the suppression matrix in sorted order, 32 pairs per word, exact pair predicate (pair_pred:
//      the reference's f32 operation order, IEEE quotient, f32 frame equality of vid_nms :111 / :169)
//   -> one wave walks the candidates with the dead mask in registers (a lane per 32-bit word), ORing a survivor's row in
//   -> kept indices + count + status written straight to host-mapped memory.
// ZeroDivisionError (Cython cdivision=False): a zero-union pair (i, j) counts iff the reference evaluates it -- i kept, j
// later, on the same frame and not yet suppressed when i is visited; rows that have such a partner are flagged while the
// matrix is built and only those re-evaluate their pairs in the walk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"

namespace vdet {

constexpr int kFusedMax = 640;         // rows per call: one CU evaluates n^2 / 2 pairs, and past ~700 rows the general chain's
                                       // chip-wide graph build is faster (measured per call through the python module: 0.05 ms at
                                       // 100 rows, 0.09 at 300, 0.59 at 1 000 against 0.3 for the general chain at 2 000)
constexpr int kFusedMaxTracks = 256;   // track rows of a fused track_det_nms call

struct FusedParams {
    const float *rows;        // [n, ncols] packed rows: (x1,y1,x2,y2,score) or (frame,x1,y1,x2,y2,score)
    const int32_t *rank;      // null, or [n] priority of every row (larger = earlier, distinct): the caller's order
    const float *tracks;      // [t, 5] (frame,x1,y1,x2,y2) or null: track_det_nms round 1
    int n, ncols, t;
    float t32;
    int32_t *hdr;             // [0] status bits (kStDivZero), [1] number kept
    int32_t *kept;            // kept row indices, descending priority
};

// A batch of independent track_det_nms problems in one launch (vdet_track_det_nms_batch): block k takes rows off[k] .. off[k+1]
// of the packed rows and track rows toff[k] .. toff[k+1]; its header is hdr[2k .. 2k+1], its kept list starts at kept[off[k]].
struct FusedBatchParams {
    const float *rows;        // [off[K], 6]
    const float *tracks;      // [toff[K], 5]
    const int32_t *off, *toff;
    float t32;
    int32_t *hdr, *kept;
};

__host__ __device__ __forceinline__ int fused_tri_words(int W) { return 16 * W * (W + 1); }
// LDS bytes of a call with n rows (n2 = n rounded up to a power of two >= 64)
__host__ __device__ __forceinline__ size_t fused_lds_bytes(int n, int n2)
{
    const int W = (n + 31) >> 5;
    const size_t raw = (size_t)n * 20;                                  // raw boxes + frames (dead once the rows are gathered)
    const size_t tri = (size_t)fused_tri_words(W) * 4;
    return (((size_t)n2 * 8 + (size_t)n * 28 + 256 + 15) & ~(size_t)15) + (tri > raw ? tri : raw);
}

template <int BLOCK>
__device__ __forceinline__ void fused_nms_body(const FusedParams &prm, unsigned char *smem, int &s_ncand, int &s_bad)
{
    const int tid = threadIdx.x;
    const int n = prm.n;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int o = prm.ncols == 6 ? 1 : 0;
    unsigned long long *comp = reinterpret_cast<unsigned long long *>(smem);               // [n2]
    float4 *sbox = reinterpret_cast<float4 *>(smem + (size_t)n2 * 8);                       // [n] sorted
    float *sarea = reinterpret_cast<float *>(sbox + n);
    float *sframe = sarea + n;
    int *sidx = reinterpret_cast<int *>(sframe + n);
    uint32_t *zflag = reinterpret_cast<uint32_t *>(sidx + n);                               // [32] rows with a zero-union partner
    uint32_t *sdead = zflag + 32;                                                           // [32] the walk's mask, when it is asked for
    unsigned char *rest = smem + (((size_t)n2 * 8 + (size_t)n * 28 + 256 + 15) & ~(size_t)15);       // (16-byte aligned: float4 rows)
    float4 *rawbox = reinterpret_cast<float4 *>(rest);                                      // [n]  } alias the matrix
    float *rawframe = reinterpret_cast<float *>(rawbox + n);                                // [n]  }
    uint32_t *tri = reinterpret_cast<uint32_t *>(rest);

    if (tid == 0) { s_ncand = 0; s_bad = 0; }
    if (tid < 32) zflag[tid] = 0u;
    __syncthreads();
    // ---- rows in, round 1 of track_det_nms, composite keys
    for (int i = tid; i < n2; i += BLOCK) {
        unsigned long long cv = 0ull;
        if (i < n) {
            const float *row = prm.rows + (size_t)i * prm.ncols;
            const float fr = o ? row[0] : 0.0f;
            const float4 b = make_float4(row[o], row[o + 1], row[o + 2], row[o + 3]);
            const float sc = row[o + 4];
            rawbox[i] = b;
            rawframe[i] = fr;
            bool excl = false;
            if (prm.tracks) {      // utils/nms.pyx:163-183: the det is box "i", tracks in order, stop at the first suppression
                const float ia = box_area(b);
                for (int j = 0; j < prm.t; ++j) {
                    const float *tr = prm.tracks + (size_t)j * 5;
                    if (fr != tr[0]) continue;
                    const float4 tb = make_float4(tr[1], tr[2], tr[3], tr[4]);
                    const uint32_t p = pair_pred(b, ia, tb, box_area(tb), prm.t32);
                    if (p & 2u) { atomicOr(&s_bad, 1); break; }
                    if (p & 1u) { excl = true; break; }
                }
            }
            const uint32_t key = prm.rank ? (uint32_t)prm.rank[i] : score_key(sc);
            if (!excl) cv = ((unsigned long long)key << 32) | (uint32_t)i;       // (key > 0: a valid composite is never 0)
        }
        comp[i] = cv;
    }
    __syncthreads();
    // ---- bitonic sort, descending (the zero padding / excluded rows sink to the end)
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (n2 >> 1); t += BLOCK) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int hi = lo | j;
                const unsigned long long a = comp[lo], b = comp[hi];
                const bool desc = (lo & k) == 0;
                if (desc ? (a < b) : (a > b)) { comp[lo] = b; comp[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int p = tid; p < n2; p += BLOCK)
        if (comp[p] != 0ull && (p == n2 - 1 || comp[p + 1] == 0ull)) s_ncand = p + 1;
    __syncthreads();
    const int nc = s_ncand;
    for (int p = tid; p < nc; p += BLOCK) {
        const int i = (int)(uint32_t)comp[p];
        const float4 b = rawbox[i];
        sidx[p] = i;
        sbox[p] = b;
        sarea[p] = box_area(b);
        sframe[p] = rawframe[i];
    }
    __syncthreads();                   // (the raw rows are dead: the matrix takes their place)
    // ---- upper triangle of the suppression matrix in sorted order: row i (the kept box, "i" of utils/nms.pyx:57-65) x
    // candidates j > i; rows of the 32-row group g keep words g .. W-1
    const int W = (nc + 31) >> 5;
    const int total = fused_tri_words(W);
    for (int e0 = tid; e0 < total; e0 += BLOCK) {
        int g = 0, e = e0;
        while (e >= 32 * (W - g)) { e -= 32 * (W - g); ++g; }         // (<= 32 cheap turns per word of 32 pair tests)
        const int wn = W - g;
        const int r = e / wn, wq = e - r * wn;
        const int i = 32 * g + r;
        uint32_t bits = 0u, zany = 0u;
        if (i < nc) {
            const float4 bi = sbox[i];
            const float ai = sarea[i], fi = sframe[i];
            const int j0 = 32 * (g + wq);
#pragma unroll 4
            for (int k = 0; k < 32; ++k) {
                const int j = j0 + k;
                if (j > i && j < nc && fi == sframe[j]) {
                    const uint32_t p = pair_pred(bi, ai, sbox[j], sarea[j], prm.t32);
                    bits |= (p & 1u) << k;
                    zany |= p >> 1;
                }
            }
            if (zany) atomicOr(&zflag[i >> 5], 1u << (i & 31));
        }
        tri[e0] = bits;              // == tri[32 * (g * W - g * (g - 1) / 2) + r * wn + wq]
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- the greedy walk (utils/nms.pyx:33-66 / :88-124), one wave: lane l owns word l of the dead mask
    const int lane = tid;
    uint32_t dead = 0u;
    const uint32_t zf = lane < 32 ? zflag[lane] : 0u;
    int nk = 0, bad = 0;
    for (int i = 0; i < nc; ++i) {
        const int g = i >> 5;
        const uint32_t dw = (uint32_t)__builtin_amdgcn_readlane((int)dead, g);
        if ((dw >> (i & 31)) & 1u) continue;
        if (lane == 0) prm.kept[nk] = sidx[i];
        ++nk;
        const int wn = W - g;
        const int base = 32 * (g * W - (g * (g - 1)) / 2) + (i & 31) * wn;
        const uint32_t rw = (lane >= g && lane < W) ? tri[base + (lane - g)] : 0u;
        if (((uint32_t)__builtin_amdgcn_readlane((int)zf, g) >> (i & 31)) & 1u) {   // rare: does the reference divide by a zero union on this row?
            if (lane < 32) sdead[lane] = dead;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const float4 bi = sbox[i];
            const float ai = sarea[i], fi = sframe[i];
            for (int j = i + 1 + lane; j < nc; j += 64)
                if (!((sdead[j >> 5] >> (j & 31)) & 1u) && fi == sframe[j] && (pair_pred(bi, ai, sbox[j], sarea[j], prm.t32) & 2u)) bad = 1;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        dead |= rw;
    }
    const bool anybad = __ballot(bad != 0) != 0ull;
    if (lane == 0) {
        // the count doubles as the "done" word the host polls (fused_wait, vdet_capi.hip): kept list and status first, a
        // system-scope fence, then the count with release semantics
        prm.hdr[0] = (anybad || s_bad) ? kStDivZero : 0;
        __threadfence_system();
        __hip_atomic_store(&prm.hdr[1], nk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fused_nms_kernel(const FusedParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_ncand, s_bad;
    fused_nms_body<BLOCK>(prm, smem, s_ncand, s_bad);
}

// one block per problem; an empty problem writes an empty header
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fused_nms_batch_kernel(const FusedBatchParams bp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_ncand, s_bad;
    const int k = blockIdx.x;
    const int r0 = bp.off[k], r1 = bp.off[k + 1], t0 = bp.toff[k], t1 = bp.toff[k + 1];
    FusedParams prm;
    prm.rows = bp.rows + (size_t)r0 * 6;
    prm.rank = nullptr;
    prm.tracks = t1 > t0 ? bp.tracks + (size_t)t0 * 5 : nullptr;
    prm.n = r1 - r0; prm.ncols = 6; prm.t = t1 - t0;
    prm.t32 = bp.t32;
    prm.hdr = bp.hdr + 2 * (size_t)k;
    prm.kept = bp.kept + r0;
    if (prm.n <= 0) {
        if (threadIdx.x == 0) { prm.hdr[0] = 0; __threadfence_system(); __hip_atomic_store(&prm.hdr[1], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        return;
    }
    fused_nms_body<BLOCK>(prm, smem, s_ncand, s_bad);
}

}  // namespace vdet
