// bucket_kernels.hpp -- K3b (round 4): the per-(frame, class) candidate lists are CUT INTO SCORE-ORDERED BUCKETS, not sorted.
//
// sort_kernel orders all 6e8 keys of a config-2 video (4 LSD passes in LDS, the dominant kernel) although the greedy walk
// that follows only ever needs the order of the candidates that are still ALIVE when it reaches them -- ~1 400 of a
// list's 10 000.  Any two boxes the walk meets in different buckets are already ordered by their buckets; inside a
// bucket only the alive members have to be ranked, and that is a handful of lanes of the walking wave
// (walk_list_bucketed, nms_kernels.hpp).  So this kernel does ONE counting pass per list:
//
//   1. histogram of the inverted sortable key's top 14 bits (sign, exponent, 5 mantissa bits: 32 bins per octave,
//      so scores that crowd one exponent still spread over dozens of LDS addresses), u16 counters packed in pairs;
//   2. exclusive scan -> cum[d] = candidates with a smaller digit (= better score);
//   3. every key gets a RANK ESTIMATE by linear interpolation inside its bin, in 1/8192 rank units:
//          fine = cum[d] * 2^13 + (m * count[d] >> 5),  m = the key's low 18 bits,
//      monotone in the key (strictly, inside bins of >= 32 keys); bucket = fine >> 18 (32 estimated ranks each, so a
//      bucket holds ~32 keys whatever the score distribution is -- the map is exact at every bin border and linear
//      over 1/32 octave in between), ord = fine & 0x3FFFF;
//   4. one returning LDS atomic per key on its bucket's counter = arrival slot inside the bucket; scan of the
//      <= 512 bucket counters; entries {ord : 18 | 0x3FFF ^ index : 14} staged in LDS, copied out coalesced.
//
// Entries of one bucket are in arrival order.  Ascending entry value IS the list's order (descending score, ties by
// descending index) except between two entries of a bucket with EQUAL ord (equal keys -- then the index bits already
// say it -- or two keys of a thin bin within 32 / count of each other): consumers detect the equal ord and let the
// full keys decide.  ~9 LDS operations per key and 8 barriers against ~20 and 22 of the LSD sort.
//
// The first kBkHead buckets are also written to `order` in exact order (one wave each, rank by lane broadcasts) for
// the tracking kernels, which read the head of every list (track_kernels.hpp: bucket_extend orders further buckets on
// demand, one at a time).
//
// A list whose buckets this map cannot keep within one wave (a bucket of more than 64 keys: heavily tied / quantised
// scores), a list of an irregular frame (the eager track_det_nms walk reads whole lists) or a tie of ord inside the
// head goes to a fail list and is sorted by the LSD kernel afterwards (sort_list_kernel); nsb[p] says which form a
// list has.  Correctness never depends on the map: it only decides how evenly the buckets fill.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"
#include "binsort_kernels.hpp"

namespace vdet {

constexpr int kBkMaxB = 16384;            // 14 index bits per entry
constexpr int kBkMax = 64;                // keys per bucket the walk can take (one per lane)
constexpr int kBkHead = 8;                // leading buckets ordered exactly for the tracking kernels
constexpr int kBkHistWords = 8192 + 4;    // 16 384 u16 counters (+ cum[16384])
constexpr int kBkCntWords = 512 + 4;      // bucket counters / starts (+ start[512])

inline size_t bucket_lds_bytes(int n) { return (size_t)4 * (kBkHistWords + kBkCntWords + (size_t)((n + 3) & ~3)); }
// u16 bucket starts per list in global memory: start[0 .. nbk] with nbk = ceil(ncand / 32) <= ceil(B / 32), padded to whole words
inline int bucket_nbs(int B) { return ((((B + 31) >> 5) + 1) + 1) & ~1; }

__device__ __forceinline__ int bucket_entry_index(uint32_t e) { return (int)(kBkIdxMask ^ (e & kBkIdxMask)); }
// two entries of one bucket whose order the entry values do not decide: equal ord, different index
__device__ __forceinline__ bool bucket_entries_tied(uint32_t a, uint32_t b) { return ((a ^ b) - 1u) < kBkIdxMask; }

struct BucketParams {
    const uint32_t *raw;           // [P, B] rows of sortable keys (0 = not a candidate), or of float32 scores (FLOATS)
    int P, B, C;
    int use_thr;                   // FLOATS: candidates are the scores > thr
    float thr;
    const GroupDesc *groups;       // one group per frame
    const uint32_t *group_flags;   // kFlagRegular per frame
    uint32_t *ent;                 // [P, B] bucketed entries
    uint16_t *bst;                 // [P, nbs] bucket starts
    int nbs;
    int32_t *ncand;                // [P]
    int32_t *nsb;                  // [P]  -1: sorted list in `order` (LSD kernel);  else buckets (k << 16 | sorted prefix length)
    uint16_t *order;               // [P, B] exact head of every list (tracking), or null
    int32_t *fail_list;            // [P]
    int *nfail;
};

// KPT = keys per thread (key v = tid + k * 1024)
template <int KPT, bool FLOATS>
__global__ __launch_bounds__(1024, KPT <= 10 ? 8 : 4) void bucket_kernel(const BucketParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const uint16_t *cum16 = reinterpret_cast<const uint16_t *>(smem);
    uint32_t *bcnt = hist + kBkHistWords;
    uint32_t *stage = bcnt + kBkCntWords;
    __shared__ uint32_t swt[16];
    __shared__ uint32_t swb[8];
    __shared__ int sfail;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int p = xcd_problem(blockIdx.x, gridDim.x);
    if (p >= prm.P) return;
    const int g = p / prm.C;
    const int N = prm.groups[g].nbox;
    const bool doable = N >= 2 && N <= kBkMaxB && N <= 1024 * KPT && (prm.group_flags[g] & kFlagRegular);
    if (!doable) {                                       // (block-uniform)
        if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
        return;
    }
    const uint32_t *src = prm.raw + (int64_t)p * prm.B;

    // the keys, inverted: ascending = the list's order; 0xFFFFFFFF = not a candidate (real inverted keys are <= 0xFF800000)
    uint32_t ik[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int v = tid + k * 1024;
        uint32_t kk = 0u;
        if (v < N) {
            const uint32_t r = src[v];
            if (FLOATS) {
                const float sc = __uint_as_float(r);
                kk = score_key(sc);
                if (prm.use_thr && !(sc > prm.thr)) kk = 0u;
            } else {
                kk = r;
            }
        }
        ik[k] = ~kk;
    }
    {
        uint4 *h4 = reinterpret_cast<uint4 *>(hist);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        h4[2 * tid] = z; h4[2 * tid + 1] = z;
        if (tid < (kBkHistWords - 8192 + kBkCntWords)) hist[8192 + tid] = 0u;          // cum[16384], bucket counters
        if (tid == 0) sfail = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k)
        if (ik[k] != 0xFFFFFFFFu) {
            const uint32_t d = ik[k] >> 18;
            atomicAdd(&hist[d >> 1], (d & 1u) ? 0x10000u : 1u);
        }
    __syncthreads();
    // exclusive scan of the 16 384 counters, in place: thread t owns digits [16 t, 16 t + 16)
    uint32_t hw[8];
    {
        const uint4 a = reinterpret_cast<const uint4 *>(hist)[2 * tid], b = reinterpret_cast<const uint4 *>(hist)[2 * tid + 1];
        hw[0] = a.x; hw[1] = a.y; hw[2] = a.z; hw[3] = a.w; hw[4] = b.x; hw[5] = b.y; hw[6] = b.z; hw[7] = b.w;
    }
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += (hw[j] & 0xFFFFu) + (hw[j] >> 16);
    const uint32_t incl = wave_incl_scan_u32(tot);
    if (lane == 63) swt[w] = incl;
    __syncthreads();
    uint32_t run = incl - tot;
    uint32_t ncand_u = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t s = swt[k];
        run += k < w ? s : 0u;
        ncand_u += s;
    }
    const int ncand = (int)ncand_u;
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t lo = hw[j] & 0xFFFFu, hi = hw[j] >> 16;
            hw[j] = run | ((run + lo) << 16);              // (cum < 16 384 wherever a key can look: no carry into the next field;
            run += lo + hi;                                //  a full list's last fields hold 16 384 = 0x4000, still 16 bits)
        }
        uint4 *h4 = reinterpret_cast<uint4 *>(hist);
        h4[2 * tid] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        h4[2 * tid + 1] = make_uint4(hw[4], hw[5], hw[6], hw[7]);
        if (tid == 1023) hist[8192] = run | (run << 16);   // cum[16384] = ncand
    }
    __syncthreads();
    // rank estimate -> bucket, slot inside the bucket
    uint32_t br[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        br[k] = 0xFFFFFFFFu;
        if (ik[k] != 0xFFFFFFFFu) {
            const uint32_t d = ik[k] >> 18, m = ik[k] & 0x3FFFFu;
            const uint32_t c0 = cum16[d], c1 = cum16[d + 1];
            const uint32_t fine = (c0 << 13) + (__umul24(m, c1 - c0) >> 5);      // m < 2^18, count <= 2^14: the product fits
            const uint32_t b = fine >> 18;
            const uint32_t r = atomicAdd(&bcnt[b], 1u);
            ik[k] = ((fine & 0x3FFFFu) << 14) | (kBkIdxMask ^ (uint32_t)(tid + k * 1024));   // the entry
            br[k] = (b << 8) | (r < 255u ? r : 255u);
        }
    }
    __syncthreads();
    // bucket starts (<= 512 buckets: waves 0..7)
    uint32_t bn = 0, bincl = 0;
    if (tid < 512) {
        bn = bcnt[tid];
        if (bn > (uint32_t)kBkMax) sfail = 1;
        bincl = wave_incl_scan_u32(bn);
        if (lane == 63) swb[w] = bincl;
    }
    __syncthreads();
    if (sfail) {                                          // (block-uniform) a bucket one wave cannot take: the LSD kernel sorts this list
        if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
        return;
    }
    const int nbk = (ncand + 31) >> 5;                    // buckets in use (est. rank < ncand)
    if (tid < 512) {
        uint32_t st = bincl - bn;
        for (int k = 0; k < w; ++k) st += swb[k];
        bcnt[tid] = st;
        if (tid == 511) bcnt[512] = st + bn;
        uint16_t *bs = prm.bst + (int64_t)p * prm.nbs;
        if (tid <= nbk) bs[tid] = (uint16_t)st;           // start[nbk] = ncand (the counters behind the last bucket are zero)
        if (tid == 511 && nbk == 512) bs[512] = (uint16_t)(st + bn);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k)
        if (br[k] != 0xFFFFFFFFu) stage[bcnt[br[k] >> 8] + (br[k] & 255u)] = ik[k];
    __syncthreads();
    {
        uint32_t *out = prm.ent + (int64_t)p * prm.B;
        if ((((int64_t)p * prm.B) & 3) == 0) {
            const int nv = ncand >> 2;
            for (int i = tid; i < nv; i += 1024) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(stage)[i];
            for (int i = (nv << 2) + tid; i < ncand; i += 1024) out[i] = stage[i];
        } else {
            for (int i = tid; i < ncand; i += 1024) out[i] = stage[i];
        }
    }
    int nhead = 0;
    if (prm.order) {
        nhead = nbk < kBkHead ? nbk : kBkHead;
        if (w < nhead) {                                  // one wave per head bucket: exact order by lane broadcasts
            const int s = (int)bcnt[w], n = (int)bcnt[w + 1] - s;
            const uint32_t e = lane < n ? stage[s + lane] : 0xFFFFFFFFu;
            uint32_t rank = 0;
            bool tie = false;
            for (int l = 0; l < n; ++l) {
                const uint32_t el = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
                rank += el < e ? 1u : 0u;
                tie = tie || bucket_entries_tied(el, e);
            }
            if (lane < n) prm.order[(int64_t)p * prm.B + s + (int)rank] = (uint16_t)bucket_entry_index(e);
            if (__ballot(tie && lane < n) != 0ull && lane == 0) sfail = 1;
        }
        __syncthreads();
        if (sfail) {
            if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
            return;
        }
    }
    if (tid == 0) {
        prm.ncand[p] = ncand;
        prm.nsb[p] = (nhead << 16) | (int)bcnt[nhead];
    }
}

}  // namespace vdet
