// small_kernels.hpp -- (round 4) frames of at most 384 proposals: the ILSVRC-VID shape (<= 300 per frame, BASELINE configs[0]
// and [4]) and every other small problem of utils/nms.pyx:17-68 / vdet/video_det.py:89-99.
//
// The per-list kernels of nms_kernels.hpp are built for lists of ~10 000 candidates: a 256-thread LSD sort (22 workgroup
// barriers per list) and a walk with one WAVE per list, eight survivors per pass of vector code.  On a batch of 64 VID-shaped
// videos (972 000 lists of <= 300 candidates) they were 3.7 + 4.3 of 20.3 ms (profiles/r04_vid_batch_kernel_stats.csv before
// this file): latency per list, not work.  Here:
//
//   small_sort_kernel   the same stable LSD passes run by ONE WAVE per list: no workgroup barrier, no cross-wave scan, four
//                       times the lists resident (3.7 -> 2.2 ms).
//   small_walk_kernel   ONE LANE per list.  A wave takes the lists of one frame (C > 32) or of several (64 / C frames); the
//                       frames' suppression ROWS (bit v of row u <=> u suppresses v) are built in LDS from the adjacency
//                       lists once and serve all their classes.  Every lane runs the reference's loop (utils/nms.pyx:33-66) on
//                       its own list: next candidate, test its bit in the lane's dead mask (registers), and -- if it is
//                       clear -- keep it and OR its row into the mask (the row: three 16-byte LDS reads, requested a step
//                       ahead).  ~60 instructions per step of 64 lists instead of ~2 400 per list (4.3 -> 1.7 ms).
//   walk_rest_kernel    what small_walk_kernel leaves: the lists of irregular frames (NaN / degenerate boxes: zero-union
//                       tags, asymmetric rows), through the general walk, one wave per list.
//
// Results are identical to the large-list kernels' (tests/test_small_gpu.py runs both: VDET_SMALL_LISTS=0).
//
// Measured and dropped in the same round: ONE WAVE per list sorting it by counting (every lane counts, for each of its <= 6
// keys, the keys before it: N^2 / 64 subtract-with-borrow + add-with-carry pairs per lane) -- 11.5 ms against the LSD kernel's
// 3.7 on that batch (a wave64 instruction takes four cycles: 3 500 of them per list are 6 ms of pure issue), and the first small
// walk, one block per frame with a scalar loop over the alive candidates of one list per wave (7.6 ms against 4.3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"
#include "binsort_kernels.hpp"

namespace vdet {

constexpr int kSmallMax = 384;          // boxes per frame: rows of at most 12 words

// ------------------------------------------------------------------------------------------------
// small_sort_kernel: the LSD radix sort of nms_kernels.hpp (sort_kernel: 4 passes x 8 bits over the inverted sortable key,
// stable, from an initial arrangement by DESCENDING index -- so equal scores come out by descending index and excluded keys
// end behind the candidates) run by ONE WAVE per list instead of a 256-thread block: no workgroup barrier, no cross-wave
// scan, 32 lists resident per CU instead of 8.  A key's rank inside its digit is what a returning LDS atomic on the digit's
// counter hands back: chunks in program order, and inside one instruction the lanes in ascending order (probed at
// vdet_create: lds_atomic_order_probe; the host only takes this kernel when the probe passed).
// KPL = chunks of 64 positions (N <= 64 * KPL); grid = waves / 4, a multiple of 8.
// ------------------------------------------------------------------------------------------------
template <int KPL>
__global__ __launch_bounds__(256) void small_sort_kernel(const SortParams prm)
{
    constexpr int NMAX = 64 * KPL;
    __shared__ uint32_t skey[4][NMAX];                                     // inverted keys by box index
    __shared__ uint16_t sidx[4][2][NMAX];                                  // the arrangement, ping-pong
    __shared__ __attribute__((aligned(16))) uint32_t shist[4][256];        // digit counters, then digit bases
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = (gridDim.x * 4) >> 3;                                  // XCD-contiguous runs of lists (block b runs on XCD b % 8)
    const int p = (blockIdx.x & 7) * per + (blockIdx.x >> 3) * 4 + w;
    if (p >= prm.P) return;
    const ProblemRef pr = decode_problem(prm.mode, p, prm.B, prm.C, prm.groups);
    const int N = pr.N;
    const int lastv = max(N - 1, 0);
    uint32_t raw[KPL];
    if (prm.keys) {
#pragma unroll
        for (int k = 0; k < KPL; ++k) raw[k] = prm.keys[pr.sbase + min(lane + 64 * k, lastv)];
    } else {
#pragma unroll
        for (int k = 0; k < KPL; ++k) raw[k] = __float_as_uint(prm.scores[pr.sbase + (int64_t)min(lane + 64 * k, lastv) * pr.sstride]);
    }
    int nx = 0;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const int v = lane + 64 * k;
        uint32_t key;
        bool x;
        if (prm.keys) { key = ~raw[k]; x = raw[k] == 0u; }                 // explicit priorities; 0 marks "not a candidate"
        else { const float sc = __uint_as_float(raw[k]); key = ~score_key(sc); x = prm.use_thr && !(sc > prm.thr); }
        x = x && v < N;
        if (x) key = 0xFFFFFFFFu;                                          // (real inverted keys are <= 0xFF800000)
        nx += __popcll(__ballot(x));
        if (v < N) { skey[w][v] = key; sidx[w][0][v] = (uint16_t)(N - 1 - v); }
    }
    const int ncand = N - nx;
    lds_mask_t hist = (lds_mask_t)((__attribute__((address_space(3))) uint32_t *)&shist[w][0]);
    int cur = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        *reinterpret_cast<uint4 *>(&shist[w][4 * lane]) = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint32_t e[KPL], rk[KPL];
        if (pass < 3) {
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int q = 64 * c + lane;
                const bool valid = q < N;
                const uint32_t i = valid ? (uint32_t)sidx[w][cur][q] : 0u;
                const uint32_t d = valid ? ((skey[w][i] >> shift) & 255u) : 0u;
                e[c] = i | (d << 16);
                rk[c] = 0u;
                if (valid) rk[c] = __hip_atomic_fetch_add((lds_u32_t *)(hist + d), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        } else {
            // the top byte (sign + exponent bits) takes a handful of values -- 64 lanes on 2-3 counters serialise in the LDS
            // (profiles/r04_pmc_vid_sq2.csv: two thirds of this kernel's LDS cycles were bank conflicts).  Up to three digits
            // seen at the head of the list are counted in registers instead: their lanes rank themselves by ballot (chunks in
            // order, lanes in order: the same stable rank), their totals reach the counters once, at the end of the pass
            uint32_t dg[3] = {0x100u, 0x100u, 0x100u}, dc[3] = {0u, 0u, 0u};
#pragma unroll
            for (int c = 0; c < KPL; ++c) {
                const int q = 64 * c + lane;
                const bool valid = q < N;
                const uint32_t i = valid ? (uint32_t)sidx[w][cur][q] : 0u;
                const uint32_t d = valid ? ((skey[w][i] >> shift) & 255u) : 0x1FFu;
                e[c] = i | ((d & 255u) << 16);
                if (c == 0) {     // (wave-uniform) the digits of the first lanes that differ
                    unsigned long long rest = __ballot(valid);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (rest) {
                            dg[k] = (uint32_t)__builtin_amdgcn_readlane((int)d, __ffsll((unsigned long long)rest) - 1);
                            rest &= ~__ballot(d == dg[k]);
                        }
                    }
                }
                rk[c] = 0u;
                bool taken = false;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const unsigned long long m = __ballot(valid && d == dg[k]);
                    if ((m >> lane) & 1ull) {
                        rk[c] = dc[k] + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        taken = true;
                    }
                    dc[k] += (uint32_t)__popcll(m);
                }
                if (valid && !taken) rk[c] = __hip_atomic_fetch_add((lds_u32_t *)(hist + d), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            {
                const uint32_t dd = lane == 0 ? dg[0] : lane == 1 ? dg[1] : dg[2], cc = lane == 0 ? dc[0] : lane == 1 ? dc[1] : dc[2];
                if (lane < 3 && dd < 256u)
                    __hip_atomic_fetch_add((lds_u32_t *)(hist + dd), cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        {   // counters -> exclusive bases (lane l owns digits 4 l .. 4 l + 3)
            const uint4 h = *reinterpret_cast<const uint4 *>(&shist[w][4 * lane]);
            const uint32_t s4 = h.x + h.y + h.z + h.w;
            const uint32_t ex = wave_incl_scan_u32(s4) - s4;
            *reinterpret_cast<uint4 *>(&shist[w][4 * lane]) = make_uint4(ex, ex + h.x, ex + h.x + h.y, ex + h.x + h.y + h.z);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < KPL; ++c) {
            if (64 * c + lane < N) sidx[w][cur ^ 1][shist[w][e[c] >> 16] + rk[c]] = (uint16_t)(e[c] & 0xFFFFu);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        cur ^= 1;
    }
    uint16_t *out = prm.order + pr.obase;
#pragma unroll
    for (int c = 0; c < KPL; ++c)
        if (64 * c + lane < N) out[64 * c + lane] = sidx[w][cur][64 * c + lane];
    if (lane == 0) prm.ncand[p] = ncand;
}

__device__ __forceinline__ bool small_walk_takes(const WalkParams &prm, int g)
{
    return prm.group_flags && (prm.group_flags[g] & kFlagRegular) && prm.group_z[g] == 0u;
}

// word `wi` of a lane's dead mask (every lane its own index): a binary tree of bit-field inserts over the mask's registers
// (written as mask arithmetic: a `b ? d[i + 1] : d[i]` on array elements makes hipcc keep the array in scratch memory and
// index it -- seen in the ISA)
__device__ __forceinline__ uint32_t pick2(uint32_t m, uint32_t hi, uint32_t lo) { return (m & hi) | (~m & lo); }   // v_bfi_b32

template <int W32>
__device__ __forceinline__ uint32_t pick_word(const uint32_t (&d)[W32], const int wi)
{
    static_assert(W32 == 4 || W32 == 8 || W32 == 12, "rows of 4, 8 or 12 words");
    const uint32_t m0 = (uint32_t)(((int)((uint32_t)wi << 31)) >> 31), m1 = (uint32_t)(((int)((uint32_t)wi << 30)) >> 31);
    const uint32_t m2 = (uint32_t)(((int)((uint32_t)wi << 29)) >> 31), m3 = (uint32_t)(((int)((uint32_t)wi << 28)) >> 31);
    uint32_t v[W32 / 2];
#pragma unroll
    for (int k = 0; k < W32 / 2; ++k) v[k] = pick2(m0, d[2 * k + 1], d[2 * k]);
    uint32_t u[W32 / 4];
#pragma unroll
    for (int k = 0; k < W32 / 4; ++k) u[k] = pick2(m1, v[2 * k + 1], v[2 * k]);
    if (W32 == 4) return u[0];
    const uint32_t x = pick2(m2, u[1], u[0]);
    if (W32 == 8) return x;
    return pick2(m3, u[W32 / 4 - 1], x);
}

// four consecutive candidates of a lane's list.  VEC4 (B % 4 == 0: every list starts on an 8-byte boundary): ONE 8-byte load,
// taken apart only when a candidate is used (unpacking right behind the load would wait for it there); otherwise four loads
template <bool VEC4> struct Cand4;
template <> struct Cand4<true> {
    uint2 x;
    __device__ __forceinline__ void load(const uint16_t *order, int q4, int last)
    {   // (positions past the list re-read its last aligned group: never walked, only their rows are prefetched)
        x = *reinterpret_cast<const uint2 *>(order + min(q4, last & ~3));
    }
    template <int I> __device__ __forceinline__ int get() const
    {
        return (int)(I == 0 ? (x.x & 0xFFFFu) : I == 1 ? (x.x >> 16) : I == 2 ? (x.y & 0xFFFFu) : (x.y >> 16));
    }
};
template <> struct Cand4<false> {
    int c[4];
    __device__ __forceinline__ void load(const uint16_t *order, int q4, int last)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = (int)order[min(q4 + i, last)];
    }
    template <int I> __device__ __forceinline__ int get() const { return c[I]; }
};

// NQ = 16-byte pieces of a row (N <= 128 * NQ).  One wave per block; it serves `fpw` frames (fpw * C <= 64, or fpw == 1 and
// the lanes take the classes in rounds of 64).  Dynamic LDS: [fpw][nmax][4 NQ] row words.
// The lane's loop is a chain of dependent steps, so everything a step needs is requested ahead of it: the candidates two
// groups of four ahead (three register sets taken in turn), the NEXT candidate's row before this one is decided (two sets) --
// whether it will be needed or not; the dead mask lives in registers (4 NQ per lane), a step's LDS traffic is that one row.
template <int NQ, bool VEC4>
__global__ __launch_bounds__(64) void small_walk_kernel(const WalkParams prm, int G, int fpw, int nmax)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slds[];
    constexpr int W32 = 4 * NQ, RG = 2 * NQ;
    typedef uint32_t lds_u4v __attribute__((ext_vector_type(4)));
    typedef volatile __attribute__((address_space(3))) lds_u4v *lds_row_t;      // (volatile: the reads stay where they are written)
    const int lane = threadIdx.x;
    const int g0 = blockIdx.x * fpw;
    uint32_t *rows = slds;
    // ---- the frames' rows, from the adjacency lists (16-byte pieces: lists are aligned and padded to 8 entries with copies).
    // lane = row (groups of 64 rows); the records of all groups first, then piece i of every group's lists together
    for (int j = 0; j < fpw; ++j) {
        const int g = g0 + j;
        if (g >= G || !small_walk_takes(prm, g)) continue;              // (wave-uniform)
        const GroupDesc gd = prm.groups[g];
        const int N = gd.nbox, rb = gd.box_off;
        if (N <= 0) continue;
        uint32_t *rj = rows + (size_t)j * nmax * W32;
        uint2 meta[RG];
        int np[RG];
        int maxnp = 0;
#pragma unroll
        for (int k = 0; k < RG; ++k) meta[k] = prm.row_meta[rb + min(lane + 64 * k, N - 1)];       // (all requested, then looked at)
#pragma unroll
        for (int k = 0; k < RG; ++k) {
            if (lane + 64 * k >= N) meta[k] = make_uint2(0u, 0u);
            np[k] = ((int)meta[k].y + 7) >> 3;
            maxnp = max(maxnp, np[k]);
        }
        for (int i = lane; i < N * W32; i += 64) rj[i] = 0u;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) maxnp = max(maxnp, __shfl_xor(maxnp, d, 64));
        maxnp = __builtin_amdgcn_readfirstlane(maxnp);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < maxnp; ++i) {
            AdjVec a[RG];
#pragma unroll
            for (int k = 0; k < RG; ++k)
                a[k] = reinterpret_cast<const AdjVec *>(prm.adj + meta[k].x)[min(i, max(np[k] - 1, 0))];
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                if (i < np[k]) {
                    // (LDS atomics without a return value: sixteen read-modify-writes of a row's words would each wait for
                    //  the one before -- they may hit the same word -- and were most of this kernel's time)
                    lds_mask_t rv = (lds_mask_t)((__attribute__((address_space(3))) uint32_t *)rj) + (lane + 64 * k) * W32;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const uint32_t e0 = a[k].v[t] & 0x7FFFu, e1 = (a[k].v[t] >> 16) & 0x7FFFu;
                        lds_or(rv, (int)(e0 >> 5), 1u << (e0 & 31u));
                        lds_or(rv, (int)(e1 >> 5), 1u << (e1 & 31u));
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- one list per lane
    const int C = prm.C;
    const int rounds = fpw > 1 ? 1 : (C + 63) >> 6;
    for (int r = 0; r < rounds; ++r) {
        const int j = fpw > 1 ? lane / C : 0;
        const int cls = fpw > 1 ? lane - j * C : lane + 64 * r;
        const int g = g0 + j;
        const bool mine = j < fpw && cls < C && g < G && small_walk_takes(prm, g);
        const int p = mine ? g * C + cls : 0;
        const int ncand = mine ? prm.ncand[p] : 0;
        const uint16_t *order = prm.order + (int64_t)p * prm.B;
        int32_t *out = prm.keep_idx + (int64_t)p * prm.cap;
        const uint32_t jrow = (uint32_t)(mine ? j : 0) * (uint32_t)nmax * W32;      // (word offset of my frame's rows)
        const int last = max(ncand - 1, 0);
        uint32_t dm[W32];
#pragma unroll
        for (int i = 0; i < W32; ++i) dm[i] = 0u;
        int nk = 0;
        int maxn = ncand;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) maxn = max(maxn, __shfl_xor(maxn, d, 64));
        maxn = __builtin_amdgcn_readfirstlane(maxn);
#define VDET_SMALL_ROW(R, CIDX)                                                                               \
        {                                                                                                     \
            lds_row_t pr_ = (lds_row_t)((__attribute__((address_space(3))) uint32_t *)slds + jrow + (uint32_t)(CIDX) * W32); \
            _Pragma("unroll") for (int i_ = 0; i_ < NQ; ++i_) {                                               \
                const lds_u4v t_ = pr_[i_];                                                                   \
                R[4 * i_ + 0] = t_.x; R[4 * i_ + 1] = t_.y; R[4 * i_ + 2] = t_.z; R[4 * i_ + 3] = t_.w;       \
            }                                                                                                 \
        }
        Cand4<VEC4> ca, cb, cc;
        uint32_t ra[W32], rb2[W32];
        ca.load(order, 0, last);
        cb.load(order, 4, last);
#define VDET_SMALL_STEP(CAND, ROW, Q, CNEXT, ROWNEXT)                                                       \
        {                                                                                                     \
            VDET_SMALL_ROW(ROWNEXT, min((CNEXT), nmax - 1))                                                   \
            const int c_ = (CAND);                                                                            \
            const uint32_t wd_ = pick_word<W32>(dm, c_ >> 5);                                                 \
            if ((Q) < ncand && !((wd_ >> (c_ & 31)) & 1u)) {                                                  \
                if ((int64_t)nk < prm.cap) out[nk] = c_;                                                      \
                ++nk;                                                                                         \
                _Pragma("unroll") for (int i_ = 0; i_ < W32; ++i_) dm[i_] |= ROW[i_];                         \
            }                                                                                                 \
        }
#define VDET_SMALL_GROUP(CUR, NXT, FAR, Q)                                                                    \
        {                                                                                                     \
            FAR.load(order, (Q) + 8, last);                                                                   \
            VDET_SMALL_STEP(CUR.template get<0>(), ra, (Q) + 0, CUR.template get<1>(), rb2)                   \
            VDET_SMALL_STEP(CUR.template get<1>(), rb2, (Q) + 1, CUR.template get<2>(), ra)                   \
            VDET_SMALL_STEP(CUR.template get<2>(), ra, (Q) + 2, CUR.template get<3>(), rb2)                   \
            VDET_SMALL_STEP(CUR.template get<3>(), rb2, (Q) + 3, NXT.template get<0>(), ra)                   \
        }
        VDET_SMALL_ROW(ra, min(ca.template get<0>(), nmax - 1))
        for (int q = 0; q < maxn; q += 12) {
            VDET_SMALL_GROUP(ca, cb, cc, q)
            if (q + 4 >= maxn) break;
            VDET_SMALL_GROUP(cb, cc, ca, q + 4)
            if (q + 8 >= maxn) break;
            VDET_SMALL_GROUP(cc, ca, cb, q + 8)
        }
#undef VDET_SMALL_GROUP
#undef VDET_SMALL_STEP
#undef VDET_SMALL_ROW
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
