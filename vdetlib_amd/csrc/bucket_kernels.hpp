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
//      <= 2 048 bucket counters; entries {ord : 16 | 0x3FFF ^ index : 14 | first-of-bucket : 1} staged in LDS;
//   5. a bucket that straddles a 64-entry chunk border (one per chunk) is put in exact order (its entries go on a work list;
//      every thread ranks a few of them by counting over the bucket in LDS) and all its entries are flagged, i.e. become
//      buckets of their own: the consumer reads the list in aligned chunks of 64 and every chunk holds whole buckets only.
//      Coalesced copy-out.
//
// Entries of one bucket are in arrival order.  Ascending (ord, flag-less entry) IS the list's order (descending score,
// ties by descending index) except between two entries of a bucket with EQUAL ord and different keys (two keys of a thin
// bin within 32 / count of each other): whoever orders entries detects equal ords and lets the full keys decide.
// ~8 LDS operations per key and 10 barriers against ~20 and 22 of the LSD sort.
//
// The first kBkHead buckets are put in exact order as well (a chunk of one-entry buckets is taken in lane order by the
// walk); with `order` given they are also written there as plain indices for the tracking kernels, which read the head of
// every list (track_kernels.hpp: bucket_extend orders further buckets on demand).  Ordering a longer head here instead of
// ranking its alive entries in the walk moves the cost from one kernel to the other, one for one (measured, kBkHead).
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
constexpr int kBkHead = 32;               // leading buckets (~250 entries) put in exact order: the tracking kernels read the head of every list,
                                          // and the walk takes a chunk of one-entry buckets in lane order (measured with 32 / 96 / 192 / 320
                                          // buckets: bucket kernel 2.14 / 2.60 / 2.70 / 3.27 ms, walk 3.95 / 3.85 / 3.72 / 3.62 ms)
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
    int head;                      // leading buckets put in exact order (kBkHead; VDET_BUCKET_HEAD is an A-B knob)
    int dbg;                       // VDET_BK_DBG (timing experiments only; results invalid): 1 no exact-order phase, 2 no copy-out, 4 stop after the scan, 8 stop after the load
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

__device__ __forceinline__ void bucket_lds_barrier() { lds_only_barrier(); }   // (nms_kernels.hpp)

constexpr uint32_t kBkNeedsRank = 0x80000000u;   // bucket start word: its entries are put in exact order (chunk border / head)
constexpr uint32_t kBkStartMask = 0x0000FFFFu;
constexpr int kBkWorkMax = 4096;                 // entries ranked per list

// BLOCK threads, KPT = keys per thread (key v = tid + k * BLOCK).  Persistent workgroups: a workgroup walks its share of the
// lists and requests the next list's keys before it starts on the current one.  512 threads x 20 keys at B = 10 000: the LDS
// (65 KB) admits two workgroups per CU whatever their size, and 8 waves each leave 128 VGPRs per lane -- 1 024 threads x 10
// keys under the 64-register cap spill the prefetched keys.
template <int BLOCK, int KPT, bool FLOATS>
__global__ __launch_bounds__(BLOCK, (BLOCK == 1024 && KPT <= 10) ? 8 : 4) void bucket_kernel(const BucketParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const uint16_t *cum16 = reinterpret_cast<const uint16_t *>(smem);
    uint32_t *work = hist;                                // (the histogram is dead once every key has its bucket)
    uint32_t *bcnt = hist + kBkHistWords;
    uint32_t *stage = bcnt + kBkCntWords;
    constexpr int NW = BLOCK / 64;                        // waves
    constexpr int HPT = (1 << kBkDigitBits) / 2 / BLOCK;  // histogram words per thread (4 or 8)
    constexpr int CPT = 2048 / BLOCK;                     // bucket counters per thread (2 or 4)
    constexpr int WPT = kBkWorkMax / BLOCK;               // ranked entries per thread
    __shared__ uint32_t swt[NW];
    __shared__ uint32_t swb[NW];
    __shared__ int sfail;
    __shared__ uint32_t nwork, ntied;
    __shared__ uint32_t tied[64];
    constexpr int kMBits = 32 - kBkDigitBits;            // low key bits interpolated inside a bin
    constexpr int kHistW = (1 << kBkDigitBits) / 2;      // counter words
    static_assert(HPT * BLOCK == kHistW && (HPT == 4 || HPT == 8) && CPT * BLOCK == 2048, "whole vectors of counters per thread");
    static_assert(kBkWorkMax <= kHistW, "the work list lives in the histogram");

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-contiguous shares (block b runs on XCD b % 8): the lists of one frame meet in one L2
    const int per = (prm.P + 7) >> 3;
    const int p_end = min(prm.P, ((int)(blockIdx.x & 7) + 1) * per);
    const int stride = (int)gridDim.x >> 3;
    int pn = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);

    uint32_t nxt[KPT];
    int n_nxt = 0;                                        // boxes of the requested list; 0: the kernel does not take it
    auto request = [&](int q) {
        n_nxt = 0;
        if (q >= p_end) return;
        const int g = q / prm.C;
        const int N = prm.groups[g].nbox;
        if (!(N >= 2 && N <= kBkMaxB && N <= BLOCK * KPT && (prm.group_flags[g] & kFlagRegular))) return;
        n_nxt = N;
        const uint32_t *src = prm.raw + (int64_t)q * prm.B;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int v = tid + k * BLOCK;
            nxt[k] = v < N ? src[v] : 0u;
        }
    };
    request(pn);
    while (pn < p_end) {
        const int p = pn, N = n_nxt;
        // the keys, inverted: ascending = the list's order; 0xFFFFFFFF = not a candidate (real inverted keys are <= 0xFF800000)
        uint32_t ik[KPT];
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int v = tid + k * BLOCK;
            uint32_t kk = 0u;
            if (v < N) {
                if (FLOATS) {
                    const float sc = __uint_as_float(nxt[k]);
                    kk = score_key(sc);
                    if (prm.use_thr && !(sc > prm.thr)) kk = 0u;
                } else {
                    kk = nxt[k];
                }
            }
            ik[k] = ~kk;
        }
        pn += stride;
        request(pn);                                      // in flight while this list is cut
        if (prm.dbg & 8) { if (ik[0] == 0x12345u) prm.ncand[p] = 1; continue; }
        if (N == 0) {                                     // (block-uniform) irregular frame / size: the LSD kernel sorts this list
            if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
            continue;
        }
        const uint32_t *src = prm.raw + (int64_t)p * prm.B;
        {
#pragma unroll
            for (int j = 0; j < HPT / 4; ++j) reinterpret_cast<uint4 *>(hist)[tid * (HPT / 4) + j] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int j = 0; j < CPT / 2; ++j) reinterpret_cast<uint2 *>(bcnt)[tid * (CPT / 2) + j] = make_uint2(0u, 0u);
            if (tid < 4) hist[kHistW + tid] = 0u;         // cum[last + 1] ...
            if (tid < 8) bcnt[2048 + tid] = 0u;
            if (tid == 0) { sfail = 0; nwork = 0u; ntied = 0u; }
        }
        bucket_lds_barrier();
#pragma unroll
        for (int k = 0; k < KPT; ++k)
            if (ik[k] != 0xFFFFFFFFu) {
                const uint32_t d = ik[k] >> kMBits;
                atomicAdd(&hist[d >> 1], (d & 1u) ? 0x10000u : 1u);
            }
        bucket_lds_barrier();
        // exclusive scan of the counters, in place: thread t owns digits [2 HPT t, 2 HPT (t + 1))
        uint32_t hw[HPT];
#pragma unroll
        for (int j = 0; j < HPT / 4; ++j) {
            const uint4 a = reinterpret_cast<const uint4 *>(hist)[tid * (HPT / 4) + j];
            hw[4 * j] = a.x; hw[4 * j + 1] = a.y; hw[4 * j + 2] = a.z; hw[4 * j + 3] = a.w;
        }
        uint32_t tot = 0;
#pragma unroll
        for (int j = 0; j < HPT; ++j) tot += (hw[j] & 0xFFFFu) + (hw[j] >> 16);
        const uint32_t incl = wave_incl_scan_u32(tot);
        if (lane == 63) swt[w] = incl;
        bucket_lds_barrier();
        uint32_t run = incl - tot;
        uint32_t ncand_u = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const uint32_t s = swt[k];
            run += k < w ? s : 0u;
            ncand_u += s;
        }
        const int ncand = (int)ncand_u;
        {
#pragma unroll
            for (int j = 0; j < HPT; ++j) {
                const uint32_t lo = hw[j] & 0xFFFFu, hi = hw[j] >> 16;
                hw[j] = run | ((run + lo) << 16);          // (cum <= 16 384: 16 bits, no carry into the next field)
                run += lo + hi;
            }
#pragma unroll
            for (int j = 0; j < HPT / 4; ++j)
                reinterpret_cast<uint4 *>(hist)[tid * (HPT / 4) + j] = make_uint4(hw[4 * j], hw[4 * j + 1], hw[4 * j + 2], hw[4 * j + 3]);
            if (tid == BLOCK - 1) hist[kHistW] = run | (run << 16);   // cum[last + 1] = ncand
        }
        bucket_lds_barrier();
        if (prm.dbg & 4) { if (tid == 0) prm.ncand[p] = ncand; continue; }
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
                ik[k] = ((fine & 0xFFFFu) << 15) | ((kBkIdxMask ^ (uint32_t)(tid + k * BLOCK)) << 1) | (r == 0u ? kBkFlag : 0u);   // the entry
                br[k] = (b << 8) | (r < 255u ? r : 255u);
            }
        }
        bucket_lds_barrier();
        // bucket starts (<= 2 048 buckets, CPT per thread)
        uint32_t bn[CPT], bsum = 0, bincl;
        {
#pragma unroll
            for (int j = 0; j < CPT / 2; ++j) {
                const uint2 a = reinterpret_cast<const uint2 *>(bcnt)[tid * (CPT / 2) + j];
                bn[2 * j] = a.x; bn[2 * j + 1] = a.y;
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                if (bn[j] > (uint32_t)kBkMax) sfail = 1;
                bsum += bn[j];
            }
            bincl = wave_incl_scan_u32(bsum);
            if (lane == 63) swb[w] = bincl;
        }
        bucket_lds_barrier();
        if (sfail) {                                      // (block-uniform) a crowded bucket: the LSD kernel sorts this list
            if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
            bucket_lds_barrier();                         // (every wave has read sfail before the next list clears it)
            continue;
        }
        const int nbk = (ncand + 7) >> 3;                 // buckets in use (est. rank < ncand)
        const int nhead = nbk < prm.head ? nbk : prm.head;
        {
            // a bucket across a 64-entry chunk border, or one of the head the tracking kernels read, is put in exact order
            uint32_t st = bincl - bsum;
            for (int k = 0; k < w; ++k) st += swb[k];
            uint32_t sw[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                const bool need = bn[j] && ((CPT * tid + j < nhead) || (bn[j] > 1u && (st >> 6) != ((st + bn[j] - 1u) >> 6)));
                sw[j] = st | (need ? kBkNeedsRank : 0u);
                st += bn[j];
            }
#pragma unroll
            for (int j = 0; j < CPT / 2; ++j) reinterpret_cast<uint2 *>(bcnt)[tid * (CPT / 2) + j] = make_uint2(sw[2 * j], sw[2 * j + 1]);
            if (tid == BLOCK - 1) bcnt[2048] = st;
        }
        bucket_lds_barrier();
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            bool needs = false;
            uint32_t item = 0u;
            if (br[k] != 0xFFFFFFFFu) {
                const uint32_t b = br[k] >> 8, sw = bcnt[b];
                const uint32_t pos = (sw & kBkStartMask) + (br[k] & 255u);
                stage[pos] = ik[k];
                needs = (sw & kBkNeedsRank) != 0u;
                item = pos | (b << 14);
            }
            // (one LDS atomic per wave and key slot: ~1 500 lanes adding to ONE counter serialise)
            const unsigned long long nm = __ballot(needs);
            if (nm) {
                const int l0 = __ffsll((unsigned long long)nm) - 1;
                uint32_t base = 0u;
                if (lane == l0) base = atomicAdd(&nwork, (uint32_t)__popcll(nm));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, l0);
                const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
                if (needs && slot < (uint32_t)kBkWorkMax) work[slot] = item;
            }
        }
        bucket_lds_barrier();
        const int nw = (prm.dbg & 1) ? 0 : (int)nwork;
        if (nw > kBkWorkMax) {                            // (block-uniform; does not happen with buckets of ~8)
            if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
            bucket_lds_barrier();
            continue;
        }
        // exact rank of the listed entries inside their buckets: rank by counting over the bucket (<= 32 entries); the flag is the
        // entries' lowest bit, so whole entries compare like (ord, index)
        uint32_t we[WPT], ws[WPT];                        // entry; start : 15 | size : 6 | rank : 6 | head : 1
#pragma unroll
        for (int h = 0; h < WPT; ++h) {
            const int i = tid + BLOCK * h;
            we[h] = 0xFFFFFFFFu; ws[h] = 0u;
            if (i < nw) {
                const uint32_t it = work[i], b = it >> 14;
                const uint32_t s = bcnt[b] & kBkStartMask, n = (bcnt[b + 1] & kBkStartMask) - s;
                const uint32_t e = stage[it & 0x3FFFu];
                // (measured: eight independent reads per turn instead of one did NOT help, 2.85 vs 2.70 ms -- more instructions)
                uint32_t rank = 0u;
                for (uint32_t j = 0; j < n; ++j) rank += stage[s + j] < e ? 1u : 0u;
                we[h] = e; ws[h] = s | (n << 15) | (rank << 21) | ((int)b < nhead ? 1u << 27 : 0u);
            }
        }
        bucket_lds_barrier();                             // every rank is counted before an entry moves
        uint16_t *head = prm.order ? prm.order + (int64_t)p * prm.B : nullptr;
#pragma unroll
        for (int h = 0; h < WPT; ++h)
            if (we[h] != 0xFFFFFFFFu) {
                const uint32_t s = ws[h] & 0x7FFFu, r = (ws[h] >> 21) & 63u;
                stage[s + r] = we[h] | kBkFlag;           // (every entry of an ordered bucket is a bucket of its own)
                if (head && (ws[h] >> 27)) head[s + r] = (uint16_t)bucket_entry_index(we[h]);
            }
        bucket_lds_barrier();
        // two neighbours with the same ord?  then the entry values did not decide their order: that bucket again, by full keys
#pragma unroll
        for (int h = 0; h < WPT; ++h)
            if (we[h] != 0xFFFFFFFFu) {
                const uint32_t s = ws[h] & 0x7FFFu, n = (ws[h] >> 15) & 63u, r = (ws[h] >> 21) & 63u;
                if (r + 1u < n && ((stage[s + r + 1u] ^ we[h]) >> 15) == 0u) {
                    const uint32_t t = atomicAdd(&ntied, 1u);
                    if (t < 64u) tied[t] = ws[h] & 0x083FFFFFu; else sfail = 1;      // start | size | head (a bucket may be listed twice)
                }
            }
        bucket_lds_barrier();
        if (sfail) {
            if (tid == 0) { prm.nsb[p] = -1; prm.fail_list[atomicAdd(prm.nfail, 1)] = p; }
            bucket_lds_barrier();
            continue;
        }
        const int nt = (int)ntied;                        // (block-uniform) rare
        if (nt) {
            if (w == 0)                                   // one wave, one listed bucket after the other (a bucket listed twice is simply redone)
                for (int i = 0; i < nt; ++i) {
                    const uint32_t t = tied[i];
                    const int s = (int)(t & 0x7FFFu), n = (int)((t >> 15) & 63u);
                    const bool member = lane < n;
                    const uint32_t ekey = member ? (stage[s + lane] & ~kBkFlag) : 0xFFFFFFFFu;
                    const uint32_t rank = bucket_rank_members<FLOATS>(ekey, member, 0, n, src);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (member) {
                        stage[s + (int)rank] = ekey | kBkFlag;
                        if (head && (t >> 27)) head[s + (int)rank] = (uint16_t)bucket_entry_index(ekey);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            bucket_lds_barrier();
        }
        if (tid == 0) {
            prm.ncand[p] = ncand;
            prm.nsb[p] = prm.order ? (int)(bcnt[nhead] & kBkStartMask) : 0;       // entries of `order` written in exact order
        }
        if (!(prm.dbg & 2)) {
            uint32_t *out = prm.ent + (int64_t)p * prm.B;
            if ((((int64_t)p * prm.B) & 3) == 0) {
                const int nv = ncand >> 2;
                for (int i = tid; i < nv; i += BLOCK) reinterpret_cast<uint4 *>(out)[i] = reinterpret_cast<const uint4 *>(stage)[i];
                for (int i = (nv << 2) + tid; i < ncand; i += BLOCK) out[i] = stage[i];
            } else {
                for (int i = tid; i < ncand; i += BLOCK) out[i] = stage[i];
            }
        }
    }
}

}  // namespace vdet
