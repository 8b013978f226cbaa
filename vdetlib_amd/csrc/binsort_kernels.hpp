// binsort_kernels.hpp -- K3', the per-(frame, class) descending argsort as ONE equalised counting pass (round 3).
//
// sort_kernel (nms_kernels.hpp) is a stable LSD radix sort: 4 passes x (gather, returning atomic, table look-up,
// scatter) over every key plus ~22 workgroup barriers per problem -- ~20 LDS operations per key, the LDS pipe and the
// barriers are its bound (profiles/r02_pmc_sq2.csv).  The order it produces is fully determined by the keys
// (descending key, ties by descending index), so ANY algorithm that ends in that order is a drop-in.  This one needs
// ~6 LDS operations per key and 7 barriers:
//
//   1. level 1: histogram of the keys' top 9 bits (sign + exponent of the inverted sortable key), 8 copies per value
//      spread over the lanes (scores concentrate on a handful of exponents; same-address LDS atomics serialise);
//   2. every exponent value e that occurs gets m_e = ceil(count_e * S / N) sub-bins of EQUAL mantissa width,
//      S = 32 256: a piecewise-linear, monotone map key -> bin in [0, 32 768) that follows the key distribution octave
//      by octave (uniform, normal, exponential / softmax-like scores all end with ~0.3 keys per bin);
//   3. one returning byte-wide LDS atomic per key counts the bin and hands the key its arrival rank r inside it;
//   4. an exclusive scan over the 32 768 byte counters (8 192 words; 8 words per thread) gives every bin its start;
//   5. the key's entry {sub-bin fraction (16 bits) | first-of-bin flag | index} goes to start + r;
//   6. fix-up: bins are contiguous runs of 1-3 entries (never more than kBinMax) in ARRIVAL order; every entry looks
//      at its run's other members (coalesced neighbour reads) and counts how many of them belong before it -- by
//      fraction, and in the rare case of equal fractions (equal keys, or an exponent with < 128 sub-bins) by the full
//      key from global memory and then by descending index.  It then stores its index at its exact final position.
//
// The result is bit-identical to sort_kernel's.  A problem the map cannot spread (a bin with more than kBinMax keys:
// heavily tied / quantised scores; or any excluded key) is appended to a fail list and sorted by the LSD kernel
// afterwards (sort_list_kernel) -- the decision is made per problem, on the device, from the counts alone.
// Workgroups are persistent and claim problems from a global counter; the next problem's keys are in flight while
// the current one is scanned, scattered and fixed up.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"

namespace vdet {

// counter words of a list of up to 512 * CPW keys (4 byte-wide bins each).  Round 6: half as many bins as before for lists of <=
// 10 240 keys (16 384 bins, ~0.6 keys per bin at config 2): since the bins are put in order by their owners (phase 6) a
// multi-key bin costs little, and the scan and the clears are half as long (1.70 -> 1.54 ms per c2 video).  Floors: the
// level-1 histogram needs 4 096 words, phase 6 keeps one byte per list position in them (>= N bytes).
__host__ __device__ constexpr int binsort_words(int cpw) { return cpw <= 20 ? 4096 : 8192; }
constexpr int kBinMax = 10;              // keys per bin (three of them must fit a 5-bit field of the scanned counter word)

struct BinSortCtl {
    int next;            // next problem to claim
    int nfail;           // problems handed to the LSD kernel
};

struct BinSortParams {
    const uint32_t *raw; // [P, N] rows of sortable keys (0 = not a candidate), or of float32 scores (FLOATS)
    int P, N;            // problems = (frame, class) pairs; boxes per frame (the same for every problem of a volume)
    uint16_t *order;     // [P, N]
    int32_t *ncand;      // [P]
    BinSortCtl *ctl;
    int32_t *fail_list;  // [P]
};

// dynamic LDS: [counter words: 32 KB][16 B][entries: N + kBinTail words]
constexpr int kBinTail = 72;             // flagged sentinels behind the last entry + slack for the last chunk's window reads
inline size_t binsort_lds_bytes(int n, int cpw) { return (size_t)(4 * binsort_words(cpw) + 16) + (size_t)4 * (size_t)(((n + 63) & ~63) + kBinTail) + (size_t)(((n + 63) & ~63) + 16); }

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x)
{
#define VDET_BS_STEP(CTRL, ROWMASK) x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xf, false);
    VDET_BS_STEP(0x111, 0xf) VDET_BS_STEP(0x112, 0xf) VDET_BS_STEP(0x114, 0xf) VDET_BS_STEP(0x118, 0xf)
    VDET_BS_STEP(0x142, 0xa) VDET_BS_STEP(0x143, 0xc)
#undef VDET_BS_STEP
    return x;
}

// Two entries of one bin with EQUAL fractions (equal keys, or an exponent with < 128 sub-bins): the full keys decide, equal keys by
// descending index.  Out of line on purpose: inlined, hipcc if-converts the rare branch and issues the two global loads for
// EVERY comparison of phase 6 (a memory round trip per bin: measured 3.0 instead of 1.9 ms per video on random scores).
template <bool FLOATS>
__device__ __attribute__((noinline)) bool binsort_tie_before(const uint32_t *__restrict__ row, uint32_t ia, uint32_t ib)
{
    const uint32_t xa = row[ia], xb = row[ib];
    const uint32_t ka = FLOATS ? score_key(__uint_as_float(xa)) : xa, kb = FLOATS ? score_key(__uint_as_float(xb)) : xb;
    if (ka != kb) return ka > kb;
    return ia > ib;
}

// CPW = keys per thread (key v = tid + k * BLOCK) = chunks of 64 positions per wave in the fix-up (even).
// 512 threads x 20 keys at B = 10 000: the LDS (72 KB) admits two workgroups per CU whatever their size, and 8 waves
// each leave 128 VGPRs per lane -- with 1 024 threads x 10 keys the 64-register cap spilled the loop's invariants.
// The kernel is bound by VALU issue (profiles/r03_pmc_sort.csv), so every phase is written for instruction count:
// full-rate 24-bit multiplies, compile-time key format, immediate LDS offsets, flags tested as lane masks.
template <int CPW, bool FLOATS>
__global__ __launch_bounds__(512, CPW <= 20 ? 4 : 2) void binsort_kernel(const BinSortParams prm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BLOCK = 512, NW = BLOCK / 64;
    constexpr int kBinWords = binsort_words(CPW), kBinTotal = 4 * kBinWords, kBinEntOff = 4 * kBinWords + 16;
    constexpr int kBinSub = kBinTotal - 512;        // sub-bins handed out proportionally (every exponent value adds < 1 by rounding up)
    static_assert(kBinWords >= 4096 && 4 * kBinWords >= 512 * (CPW <= 20 ? 20 : 36), "level-1 table and one byte per position");
    constexpr int WPT = kBinWords / BLOCK;          // counter words per thread in the scan
    static_assert(CPW % 2 == 0, "the fix-up takes two chunks per round");
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);                        // [kBinWords]; level-1 table in its first 4 096 words
    uint32_t *ent = reinterpret_cast<uint32_t *>(smem + kBinEntOff);            // [N] fraction << 16 | first << 15 | index
    uint8_t *blen = reinterpret_cast<uint8_t *>(ent + (((prm.N + 63) & ~63) + kBinTail));      // [N] size of the bin that starts at a position
    __shared__ uint32_t tab[512];        // per exponent value: first bin << 16 | number of sub-bins
    __shared__ uint32_t wsum[NW];
    __shared__ int snext;
    __shared__ int sbad;

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = prm.P, N = prm.N;

    // (keys are INVERTED sortable keys: ascending = descending score; 0xFFFFFFFF = not a candidate)
    // Round 4 (found in the ISA): with `ik[k] = v < N ? ikey(row, v) : ...` every key's load sat in its own basic block with an
    // s_waitcnt vmcnt(0) behind it -- CPW dependent memory round trips per list -- and the "prefetch" of the next list's keys was
    // drained by the next __syncthreads() anyway (its fence waits for every global load on gfx9).  Now the RAW words are requested
    // with unconditional, clamped loads (all in flight together), the barriers of the loop order LDS traffic only
    // (lds_only_barrier), and the words are turned into inverted keys when the list's turn comes (finish_keys).
    auto load_keys = [&](int p, uint32_t (&ik)[CPW]) {
        const uint32_t *row = prm.raw + (int64_t)p * N;          // (scalar base + 32-bit lane offsets)
        uint32_t off = (uint32_t)tid;
        asm volatile("" : "+v"(off));                            // not hoisted out of the problem loop as CPW 64-bit pointers
#pragma unroll
        for (int k = 0; k < CPW; ++k) ik[k] = row[min(off + (uint32_t)(k * BLOCK), (uint32_t)(N - 1))];
    };
    auto finish_keys = [&](uint32_t (&ik)[CPW]) {
        uint32_t off = (uint32_t)tid;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const uint32_t x = ik[k];
            ik[k] = off + (uint32_t)(k * BLOCK) < (uint32_t)N ? ~(FLOATS ? score_key(__uint_as_float(x)) : x) : 0xFFFFFFFFu;
        }
    };

    if (tid == 0) snext = atomicAdd(&prm.ctl->next, 1);
    for (int i = tid; i < kBinWords / 4; i += BLOCK) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < kBinTail) ent[N + tid] = 0x8000u;                  // sentinels: "a new bin starts here" closes every walk to the right
    __syncthreads();
    int p = __builtin_amdgcn_readfirstlane(snext);      // (wave-uniform by construction: scalar addressing)
    __syncthreads();                 // (thread 0 overwrites snext at the top of the loop)
    uint32_t ik[CPW];
    if (p < P) load_keys(p, ik);

    // (every phase re-derives what it needs from an OPAQUE copy of the thread index: left alone, hipcc hoists each
    //  phase's addresses and predicates out of the problem loop and then spills them -- seen in the ISA)
    auto fresh_tid = [&]() -> int { int t = tid; asm volatile("" : "+v"(t)); return t; };
    while (p < P) {
        int bad = 0;
        finish_keys(ik);
        if (tid == 0) sbad = 0;
        // ---- 1. level 1: exponent histogram, 8 copies per value (lane & 7)
        {
            const int t = fresh_tid();
            const uint32_t copy = (uint32_t)(t & 7);
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const bool none = ik[k] == 0xFFFFFFFFu;                          // (slots past N hold 0xFFFFFFFF too)
                bad |= (int)(none & (t + k * BLOCK < N));                        // excluded keys: the LSD kernel's business
                if (!none) atomicAdd(&hist[((ik[k] >> 23) << 3) | copy], 1u);
            }
        }
        if (tid == 0) snext = atomicAdd(&prm.ctl->next, 1);                     // (read after the barriers below)
        lds_only_barrier();
        // ---- 2. sub-bins per exponent value: thread e owns value e (BLOCK == 512 values); its level-1 words are zeroed again
        {
            static_assert(BLOCK == 512, "one thread per exponent value");
            const int e = fresh_tid();
            uint4 *src = reinterpret_cast<uint4 *>(hist) + e * 2;
            const uint4 a = src[0], b = src[1];
            src[0] = make_uint4(0u, 0u, 0u, 0u); src[1] = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t cnt = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
            const uint32_t m = cnt ? (cnt * (uint32_t)kBinSub + (uint32_t)N - 1u) / (uint32_t)N : 0u;      // (cnt <= N < 2^15, kBinSub < 2^15)
            const uint32_t incl = wave_incl_scan_u32(m);
            if ((e & 63) == 63) wsum[w] = incl;
            lds_only_barrier();
            uint32_t run = incl - m;
            for (int k = 0; k < w; ++k) run += wsum[k];
            tab[e] = (run << 16) | m;
        }
        lds_only_barrier();
        const int pnext = __builtin_amdgcn_readfirstlane(snext);
        // ---- 3. bin + arrival rank of every key.  Each stage of the phase runs over ALL the thread's keys before the next
        //         one starts (table reads, then atomics, then their returns): 16 waves per CU hide no LDS latency, the
        //         20 independent operations per thread do
        uint32_t cs[CPW];            // bin << 16 | fraction
        uint32_t rr[(CPW + 3) / 4];  // ranks, one byte each
        {
#pragma unroll
            for (int k = 0; k < CPW; ++k) cs[k] = tab[ik[k] >> 23];
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                // bin = first + (mant * m) >> 23, fraction = the next 16 bits: 23 x 16-bit product from the full-rate 24-bit
                // multipliers (v_mul_lo / v_mul_hi_u32 are quarter rate)
                const uint32_t t = cs[k], mant = ik[k] & 0x7FFFFFu, m = t & 0xFFFFu;
                uint32_t lo, hi;
                asm("v_mul_u32_u24 %0, %1, %2" : "=v"(lo) : "v"(mant), "v"(m));
                asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(hi) : "v"(mant), "v"(m));
                const uint32_t binv = (t >> 16) + __builtin_amdgcn_alignbit(hi, lo, 23);
                cs[k] = ik[k] == 0xFFFFFFFFu ? 0u : ((binv << 16) | ((lo >> 7) & 0xFFFFu));
            }
            uint32_t old[CPW];
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                old[k] = 0u;
                if (ik[k] != 0xFFFFFFFFu) old[k] = atomicAdd(&hist[cs[k] >> 18], 1u << ((cs[k] >> 13) & 24u));
            }
#pragma unroll
            for (int k = 0; k < (CPW + 3) / 4; ++k) rr[k] = 0u;
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const uint32_t r = (old[k] >> ((cs[k] >> 13) & 24u)) & 0xFFu;
                if (r >= (uint32_t)kBinMax) bad = 1;
                rr[k >> 2] |= r << ((k & 3) * 8);
            }
        }
        // which of my slots hold keys: remembered as a bit per slot (ik is about to be reused)
        unsigned long long have = 0ull;
#pragma unroll
        for (int k = 0; k < CPW; ++k) have |= (unsigned long long)(ik[k] != 0xFFFFFFFFu ? 1u : 0u) << k;
        // the next problem's keys travel while this one is scanned, scattered and fixed up
        if (pnext < P) load_keys(pnext, ik);
        if (bad) sbad = 1;
        lds_only_barrier();
        if (sbad) {
            // not spreadable (ties / quantised scores / exclusions): hand the problem to the LSD kernel
            if (tid == 0) prm.fail_list[atomicAdd(&prm.ctl->nfail, 1)] = p;
            for (int i = tid; i < kBinWords / 4; i += BLOCK) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
            lds_only_barrier();
            p = pnext;
            continue;
        }
        // ---- 4. exclusive scan of the byte counters; word <- start | s1 << 17 | s2 << 22 | s3 << 27, s_j = keys in the
        //         word's first j bins (<= 3 * kBinMax = 30: five bits)
        //         (two reads of the thread's 16 words instead of 16 live registers + their shifted copies across the barrier)
        {
            uint4 *hw = reinterpret_cast<uint4 *>(hist) + fresh_tid() * (WPT / 4);
            auto bytesum = [](uint32_t x) -> uint32_t {
                const uint32_t h2 = (x & 0x00FF00FFu) + ((x >> 8) & 0x00FF00FFu);
                return (h2 & 0xFFFFu) + (h2 >> 16);
            };
            uint32_t tot = 0;
#pragma unroll
            for (int j = 0; j < WPT / 4; ++j) { const uint4 a = hw[j]; tot += bytesum(a.x) + bytesum(a.y) + bytesum(a.z) + bytesum(a.w); }
            const uint32_t incl = wave_incl_scan_u32(tot);
            if (lane == 63) wsum[w] = incl;
            lds_only_barrier();
            uint32_t run = incl - tot;
            for (int k = 0; k < w; ++k) run += wsum[k];
            // word <- start (15 bits: N <= 18 432) | the four bins' counts, a nibble each (<= kBinMax): the scatter needs the keys
            // in the word's earlier bins AND the size of the key's own bin (phase 6 sorts bins of two or more)
            auto enc = [&](uint32_t x) -> uint32_t {
                const uint32_t c = (x & 0xFu) | ((x >> 4) & 0xF0u) | ((x >> 8) & 0xF00u) | ((x >> 12) & 0xF000u);
                const uint32_t r = run | (c << 15);
                run += bytesum(x);
                return r;
            };
            uint4 *hw2 = reinterpret_cast<uint4 *>(hist) + fresh_tid() * (WPT / 4);
#pragma unroll 2
            for (int j = 0; j < WPT / 4; ++j) {
                const uint4 a = hw2[j];
                uint4 o;
                o.x = enc(a.x); o.y = enc(a.y); o.z = enc(a.z); o.w = enc(a.w);
                hw2[j] = o;
            }
        }
        lds_only_barrier();
        // ---- 5. scatter: position = word start + keys in the word's earlier bins + arrival rank
        {
            const int t5 = fresh_tid();
            uint32_t e[CPW];
#pragma unroll
            for (int k = 0; k < CPW; ++k) e[k] = hist[cs[k] >> 18];
#pragma unroll
            for (int k = 0; k < CPW; ++k) {
                const uint32_t j = (cs[k] >> 16) & 3u;
                const uint32_t r = (rr[k >> 2] >> ((k & 3) * 8)) & 0xFFu;
                const uint32_t c = e[k] >> 15;                                  // the word's four counts
                const uint32_t lowc = c & ((1u << (4u * j)) - 1u);             // ... of the bins in front of mine
                const uint32_t before = (lowc & 0xFu) + ((lowc >> 4) & 0xFu) + (lowc >> 8);
                const uint32_t cnt = (c >> (4u * j)) & 0xFu;                   // ... and of my own
                const uint32_t pos = (e[k] & 0x7FFFu) + r + before;
                if ((have >> k) & 1ull) {
                    ent[pos] = (cs[k] << 16) | (r == 0u ? 0x8000u : 0u) | (uint32_t)(t5 + k * BLOCK);
                    blen[pos] = (uint8_t)(r == 0u ? cnt : 0u);                  // at a bin's first place: its size; 0 elsewhere
                }
            }
        }
        lds_only_barrier();
        // ---- 6. the bins in exact order, then the store; the counters are cleared for the next problem meanwhile.
        // Bins are contiguous runs of entries in ARRIVAL order; with ~0.6 keys per bin about half of the keys share theirs.
        // (Until round 6 every entry looked at three neighbours to its left and four to its right and counted which of them belong
        // on its other side: 8 LDS reads and ~55 VALU instructions per entry, 1.0 of the kernel's 2.36 ms by cut-off launches.)
        // The scatter left the size of every bin at its first place: the lane that finds a size of two or more at one of its
        // positions puts that bin in order in place -- two or three LDS words read together, compared, written back; the rare
        // longer bin by insertion -- by fraction, equal fractions (equal keys, or an exponent with < 128 sub-bins) by the full key
        // from global memory, equal keys by descending index.  Then one coalesced store of the whole list.
        for (int i = fresh_tid(); i < kBinWords / 4; i += BLOCK) reinterpret_cast<uint4 *>(hist)[i] = make_uint4(0u, 0u, 0u, 0u);
        {
            const int lane6 = fresh_tid() & 63;
            uint16_t *out = prm.order + (int64_t)p * N;
            const uint32_t *row = prm.raw + (int64_t)p * N;
            // does entry a belong in front of entry b?  (smaller fraction of the inverted key = larger key = earlier)
            auto before = [&](uint32_t a, uint32_t b) -> bool {
                const uint32_t fa = a >> 16, fb = b >> 16;
                if (__builtin_expect(fa != fb, 1)) return fa < fb;
                return binsort_tie_before<FLOATS>(row, a & 0x7FFFu, b & 0x7FFFu);
            };
            unsigned long long own = 0ull;       // bit ch: my position of chunk ch starts a bin of >= 2 entries (CPW <= 36 chunks)
            static_assert(CPW <= 64, "one bit per chunk");
#pragma unroll
            for (int ch = 0; ch < CPW; ++ch) {
                const int q = (w * CPW + ch) * 64 + lane6;
                const uint32_t nb = blen[min(q, N - 1)];
                own |= (q < N && nb >= 2u) ? (1ull << ch) : 0ull;
            }
            while (own) {
                const int ch = __ffsll(own) - 1;
                own &= own - 1ull;
                const int q = (w * CPW + ch) * 64 + lane6;
                uint32_t *bin = ent + q;
                const int n = (int)blen[q];
                uint32_t x0 = bin[0], x1 = bin[1], x2 = bin[n > 2 ? 2 : 1];       // (read together: one LDS round trip)
                if (n <= 3) {
                    if (before(x1, x0)) { const uint32_t t = x0; x0 = x1; x1 = t; }
                    if (n == 3) {
                        if (before(x2, x1)) { const uint32_t t = x1; x1 = x2; x2 = t; }
                        if (before(x1, x0)) { const uint32_t t = x0; x0 = x1; x1 = t; }
                        bin[2] = x2;
                    }
                    bin[0] = x0; bin[1] = x1;
                } else if (n <= 6) {
                    // four to six entries (~60 bins of a 10 000-key list of random scores): all read together, insertion by
                    // compare-exchange in registers -- as an insertion sort on LDS words this path alone cost 0.7 ms per video
                    // (one dependent LDS round trip per comparison while the rest of the wave waits)
                    uint32_t v[6];
                    v[0] = x0; v[1] = x1; v[2] = x2; v[3] = bin[3]; v[4] = bin[n > 4 ? 4 : 3]; v[5] = bin[n > 5 ? 5 : 3];
#pragma unroll
                    for (int i = 1; i < 6; ++i) {
#pragma unroll
                        for (int j = i; j > 0; --j) {
                            if (i < n && before(v[j], v[j - 1])) { const uint32_t t = v[j]; v[j] = v[j - 1]; v[j - 1] = t; }
                        }
                    }
                    bin[0] = v[0]; bin[1] = v[1]; bin[2] = v[2]; bin[3] = v[3];
                    if (n > 4) bin[4] = v[4];
                    if (n > 5) bin[5] = v[5];
                } else {
                    for (int i = 1; i < n; ++i) {
                        const uint32_t x = bin[i];
                        int j = i;
                        while (j > 0) {
                            const uint32_t y = bin[j - 1];
                            if (!before(x, y)) break;
                            bin[j] = y;
                            --j;
                        }
                        bin[j] = x;
                    }
                }
            }
            lds_only_barrier();
#pragma unroll
            for (int ch = 0; ch < CPW; ++ch) {
                const int q = (w * CPW + ch) * 64 + lane6;
                if (q < N) out[q] = (uint16_t)(ent[q] & 0x7FFFu);
            }
            // every key of the list is a candidate HERE: a list with an excluded key (NaN score, key 0xFFFFFFFF -- `bad` in
            // phase 1, decided on the device, not by the host's gating) was handed to the LSD kernel, which counts its own
            if (tid == 0) prm.ncand[p] = N;
        }
        lds_only_barrier();
        p = pnext;
    }
}

}  // namespace vdet
