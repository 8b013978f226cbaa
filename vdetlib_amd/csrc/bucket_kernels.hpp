// bucket_kernels.hpp -- K3b (round 4): the per-(frame, class) candidate lists are CUT INTO SCORE-ORDERED BUCKETS, not sorted.
//
// sort_kernel orders all 6e8 keys of a config-2 video (4 LSD passes in LDS, the dominant kernel) although the greedy walk
// that follows only ever needs the order of the candidates that are still ALIVE when it reaches them -- ~1 400 of a
// list's 10 000.  Any two boxes the walk meets in different buckets are already ordered by their buckets; inside a
// bucket only the alive members have to be ranked, and that is a handful of lanes of the walking wave
// (walk_list_bucketed, nms_kernels.hpp).  So this kernel does ONE counting pass per list:
//
//   1. histogram of the inverted sortable key's top 13 bits (sign, exponent, 4 mantissa bits: 16 bins per octave,
//      so scores that crowd one exponent still spread over the LDS), u16 counters packed in pairs;
//   2. exclusive scan -> cum[d] = candidates with a smaller digit (= better score);
//   3. every key gets a RANK ESTIMATE by linear interpolation inside its bin, in 1/8192 rank units:
//          fine = cum[d] * 2^13 + (m * count[d] >> 5),  m = the next 18 bits of the key,
//      monotone in the key; bucket = fine >> 16 (8 estimated ranks each, so a bucket holds ~8 keys whatever the score
//      distribution is -- the map is exact at every bin border and linear over 1/16 octave in between), ord = fine & 0xFFFF;
//   4. one returning LDS atomic per key on its bucket's counter = arrival slot inside the bucket; scan of the
//      <= 2 048 bucket counters; entries {ord : 16 | first-of-bucket : 1 | 0x3FFF ^ index : 14} staged in LDS;
//   5. a bucket that straddles a 64-entry chunk border (one per chunk) is put in exact order by one wave (rank by lane
//      broadcasts) and all its entries are flagged, i.e. become buckets of their own: the consumer reads the list in
//      aligned chunks of 64 and every chunk holds whole buckets only.  Coalesced copy-out.
//
// Entries of one bucket are in arrival order.  Ascending (ord, flag-less entry) IS the list's order (descending score,
// ties by descending index) except between two entries of a bucket with EQUAL ord and different keys (two keys of a thin
// bin within 32 / count of each other): whoever orders entries detects equal ords and lets the full keys decide.
// ~8 LDS operations per key and 10 barriers against ~20 and 22 of the LSD sort.
//
// The first kBkHead buckets are also written to `order` in exact order for the tracking kernels, which read the head of
// every list (track_kernels.hpp: bucket_extend orders further buckets on demand, one at a time).
//
// A list whose buckets this map cannot keep small (a bucket of more than 32 keys: heavily tied / quantised scores) or a
// list of an irregular frame (the eager track_det_nms walk reads whole lists) goes to a fail list and is sorted by the
// LSD kernel afterwards (sort_list_kernel); nsb[p] says which form a list has.  Correctness never depends on the map:
// it only decides how evenly the buckets fill.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"
#include "binsort_kernels.hpp"

namespace vdet {

constexpr int kBkMaxB = 16384;            // 14 index bits per entry
constexpr int kBkMax = 32;                // keys per bucket (a bucket and its neighbours fit one wave)
constexpr int kBkHead = 32;               // leading buckets ordered exactly for the tracking kernels (two per wave)
constexpr int kBkDigitBits = 13;
constexpr int kBkHistWords = (1 << kBkDigitBits) / 2 + 4;    // u16 counters (+ cum[last + 1])
constexpr int kBkCntWords = 2048 + 8;     // bucket counters / starts (+ start[2048])

inline size_t bucket_lds_bytes(int n) { return (size_t)4 * (kBkHistWords + kBkCntWords + (size_t)((n + 3) & ~3)); }

struct BucketParams {
    const uint32_t *raw;           // [P, B] rows of sortable keys (0 = not a candidate), or of float32 scores (FLOATS)
    int P, B, C;
    int use_thr;                   // FLOATS: candidates are the scores > thr
    float thr;
    const GroupDesc *groups;       // one group per frame
    const uint32_t *group_flags;   // kFlagRegular per frame
    uint32_t *ent;                 // [P, B] bucketed entries
    int32_t *ncand;                // [P]
    int32_t *nsb;                  // [P]  -1: sorted list in `order` (LSD kernel);  else entries of `order` in exact order so far
    uint16_t *order;               // [P, B] exact head of every list (tracking), or null
    int32_t *fail_list;            // [P]
    int *nfail;
};

template <bool FLOATS>
__device__ __forceinline__ uint32_t bucket_full_ikey(const uint32_t *__restrict__ src, int idx)
{
    const uint32_t r = src[idx];
    return ~(FLOATS ? score_key(__uint_as_float(r)) : r);
}

// Exact rank of lane's entry among the member lanes [ls, le) of one wave (one bucket, <= 64 entries): by flag-less entry
// value, and -- if two members share an ord without sharing an index -- by the full keys (global loads, rare).
template <bool FLOATS>
__device__ __forceinline__ uint32_t bucket_rank_members(uint32_t ekey, bool member, int ls, int le, const uint32_t *__restrict__ src)
{
    uint32_t rank = 0;
    bool tie = false;
    for (int l = ls; l < le; ++l) {
        const uint32_t el = (uint32_t)__builtin_amdgcn_readlane((int)ekey, l);
        rank += el < ekey ? 1u : 0u;
        tie = tie || bucket_entries_tied(el, ekey);
    }
    if (__ballot(tie && member) != 0ull) {          // (wave-uniform)
        const uint32_t kf = member ? bucket_full_ikey<FLOATS>(src, bucket_entry_index(ekey)) : 0u;
        rank = 0;
        for (int l = ls; l < le; ++l) {
            const uint32_t il = (uint32_t)__builtin_amdgcn_readlane((int)kf, l);
            const uint32_t el = (uint32_t)__builtin_amdgcn_readlane((int)ekey, l);
            rank += (il < kf || (il == kf && el < ekey)) ? 1u : 0u;      // equal keys: the higher index first (its entry is smaller)
        }
    }
    return rank;
}

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
    __shared__ uint32_t swb[16];
    __shared__ int sfail;
    constexpr int kMBits = 32 - kBkDigitBits;            // low key bits interpolated inside a bin
    constexpr int kHistW = (1 << kBkDigitBits) / 2;      // counter words
    static_assert(kHistW == 4 * 1024, "one uint4 of counters per thread");

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
        reinterpret_cast<uint4 *>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint2 *>(bcnt)[tid] = make_uint2(0u, 0u);
        if (tid < 4) hist[kHistW + tid] = 0u;             // cum[last + 1] ...
        if (tid < 8) bcnt[2048 + tid] = 0u;
        if (tid == 0) sfail = 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k)
        if (ik[k] != 0xFFFFFFFFu) {
            const uint32_t d = ik[k] >> kMBits;
            atomicAdd(&hist[d >> 1], (d & 1u) ? 0x10000u : 1u);
        }
    __syncthreads();
    // exclusive scan of the counters, in place: thread t owns digits [8 t, 8 t + 8)
    uint32_t hw[4];
    {
        const uint4 a = reinterpret_cast<const uint4 *>(hist)[tid];
        hw[0] = a.x; hw[1] = a.y; hw[2] = a.z; hw[3] = a.w;
    }
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) tot += (hw[j] & 0xFFFFu) + (hw[j] >> 16);
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
        for (int j = 0; j < 4; ++j) {
            const uint32_t lo = hw[j] & 0xFFFFu, hi = hw[j] >> 16;
            hw[j] = run | ((run + lo) << 16);              // (cum <= 16 384: 16 bits, no carry into the next field)
            run += lo + hi;
        }
        reinterpret_cast<uint4 *>(hist)[tid] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if (tid == 1023) hist[kHistW] = run | (run << 16);   // cum[last + 1] = ncand
    }
    __syncthreads();
    // rank estimate -> bucket, slot inside the bucket
    uint32_t br[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        br[k] = 0xFFFFFFFFu;
        if (ik[k] != 0xFFFFFFFFu) {
            const uint32_t d = ik[k] >> kMBits, m = (ik[k] & ((1u << kMBits) - 1u)) >> (kMBits - 18);   // 18 bits of the remainder
            const uint32_t c0 = cum16[d], c1 = cum16[d + 1];
            const uint32_t fine = (c0 << 13) + (__umul24(m, c1 - c0) >> 5);      // m < 2^18, count <= 2^14: the product fits
            const uint32_t b = fine >> 16;
            const uint32_t r = atomicAdd(&bcnt[b], 1u);
            ik[k] = ((fine & 0xFFFFu) << 15) | (r == 0u ? kBkFlag : 0u) | (kBkIdxMask ^ (uint32_t)(tid + k * 1024));   // the entry
            br[k] = (b << 8) | (r < 255u ? r : 255u);
        }
    }
    __syncthreads();
    // bucket starts (<= 2 048 buckets, two per thread)
    uint32_t bn0, bn1, bincl;
    {
        const uint2 a = reinterpret_cast<const uint2 *>(bcnt)[tid];
        bn0 = a.x; bn1 = a.y;
        if (bn0 > (uint32_t)kBkMax || bn1 > (uint32_t)kBkMax) sfail = 1;
        bincl = wave_incl_scan_u32(bn0 + bn1);
        if (lane == 63) swb[w] = bincl;
    }
    __syncthreads();
    if (sfail) {                                          // (block-uniform) a crowded bucket: the LSD kernel sorts this list
        if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
        return;
    }
    const int nbk = (ncand + 7) >> 3;                     // buckets in use (est. rank < ncand)
    {
        uint32_t st = bincl - (bn0 + bn1);
        for (int k = 0; k < w; ++k) st += swb[k];
        reinterpret_cast<uint2 *>(bcnt)[tid] = make_uint2(st, st + bn0);
        if (tid == 1023) bcnt[2048] = st + bn0 + bn1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KPT; ++k)
        if (br[k] != 0xFFFFFFFFu) stage[bcnt[br[k] >> 8] + (br[k] & 255u)] = ik[k];
    __syncthreads();
    // the bucket across every 64-entry chunk border: exact order, every entry a bucket of its own
    {
        const int nchunk = (ncand + 63) >> 6;
        for (int j = 1 + w; j < nchunk; j += 16) {
            const int pos0 = 64 * j - 32, q = pos0 + lane;
            const bool valid = q < ncand;
            const uint32_t e = valid ? stage[q] : 0xFFFFFFFFu;
            const unsigned long long fm = __ballot(!valid || (e & kBkFlag) != 0u);     // lanes where a bucket starts
            if ((fm >> 32) & 1ull) continue;               // a bucket starts right at the border: nothing straddles
            const unsigned long long low = fm & 0xFFFFFFFFull, high = fm >> 33;
            const int ls = 63 - __builtin_clzll(low);      // (buckets hold <= 32 entries: the start is within reach)
            const int le = high ? 33 + (__ffsll((unsigned long long)high) - 1) : 64;
            const bool member = lane >= ls && lane < le;
            const uint32_t ekey = e & ~kBkFlag;
            const uint32_t rank = bucket_rank_members<FLOATS>(ekey, member, ls, le, src);
            if (member) stage[pos0 + ls + (int)rank] = ekey | kBkFlag;
        }
    }
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
        for (int b = w; b < nhead; b += 16) {             // exact order of the head buckets, one wave each
            const int s = (int)bcnt[b], n = (int)bcnt[b + 1] - s;
            const bool member = lane < n;
            const uint32_t ekey = member ? (stage[s + lane] & ~kBkFlag) : 0xFFFFFFFFu;
            const uint32_t rank = bucket_rank_members<FLOATS>(ekey, member, 0, n, src);
            if (member) prm.order[(int64_t)p * prm.B + s + (int)rank] = (uint16_t)bucket_entry_index(ekey);
        }
    }
    if (tid == 0) {
        prm.ncand[p] = ncand;
        prm.nsb[p] = (int)bcnt[nhead];
    }
}

}  // namespace vdet
