// graphlists_kernels.hpp -- K1d: the suppression graph of REGULAR large frames straight into adjacency lists (round 6).
//
// iou_bits_sym_kernel + adj_rows_kernel (nms_kernels.hpp, adjrows_kernels.hpp) write the predicate as a 2.1 GB bit matrix per
// config-2 video and read it back to re-encode it as u16 lists.  Here the wave that finishes a 64 x 64 block of the predicate
// takes its words apart on the spot:
//   * lists live in FIXED slots of slot_cap entries per row (rank-row v of a group at (box_off + v) * slot_cap); the row's
//     degree counter doubles as the list's cursor -- ONE returning atomicAdd per (row, non-zero word) reserves the places.  The
//     atomics of a block are issued when the block is done and used one block later (the next block's ~80 000 issue slots of
//     pair tests hide the round trip; measured: the atomics cost nothing);
//   * a pair is evaluated ONCE, by the wave that holds the box of lower x1 rank as a row; both directions are emitted -- the
//     row words give the rows' entries, their in-wave 64 x 64 transpose (wave_transpose64) the columns' entries;
//   * entries are box indices (x1 rank -> index through LDS copies of xord) and leave as 2-byte stores from straight-line code:
//     a wave's lanes hold ~54 entries per 32-bit half, seldom more than 4 in one lane, so the first four bits of every lane are
//     handled without a loop (all bit positions, all translations in flight, stores under their own exec masks); the loop form --
//     a ballot, three branches and an exposed LDS round trip per turn -- cost 0.42 ms per video, this form 0.38;
//   * no bit matrix means a wave may hold ANY 64 rows: row_classes_kernel deals the 256 rows of a tile to four work items BY WIDTH.
//     How far to the right a row can find a partner is (1 - t) x its width, a wave evaluates column blocks up to the largest
//     reach of its rows, and with consecutive ranks every wave holds a box of nearly the largest width (reach ~207 px at config
//     2; by quartile 58 / 109 / 160 / 210 px: ~23 % fewer pair tests);
//   * work item = ONE WAVE = (row tile, width quartile): it walks its column blocks left to right and stops at the first block
//     beyond its reach (no futile work items; with the four quartiles of a tile in one 256-thread block the narrow ones idled
//     until the widest was done: - 7 % instead of - 23 %; with one block per (tile, quartile, column tile) two thirds of 246 000
//     blocks per launch only found out that they had nothing to do: + 20 %).  The next block's columns are requested while the
//     current block is evaluated and staged in the other half of a double LDS buffer;
//   * every XCD takes a contiguous eighth of the work items (whole frames): the partially written lines of a slot stay in ONE
//     L2 until the frame is done (spread over all XCDs: 4.3 x write amplification, PMC WRITE_SIZE).
// adj_finish_kernel (adjrows_kernels.hpp) pads the lists and writes the walk's records.  A row with more neighbours than a slot
// holds latches over_bits (its surplus entries are dropped): the host rebuilds the graph through the bit matrix.  Entry order
// inside a list depends on the order of the atomics; lists are sets to every consumer.  Predicate: pred_margins, the same two
// signed margins as iou_bits_sym_kernel (utils/nms.pyx:57-65 without the division; the half-ulp band redone with the IEEE quotient).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"

namespace vdet {

// (direct lists) slot of (tile 0, quartile 0) of group g in the per-quartile reach table: four entries per 256-row tile
__device__ __forceinline__ int qreach_slot(const GroupDesc &gd, int g) { return ((gd.box_off >> 8) + g) * 4; }

// Which rank sits in slot s of its 256-row tile -- the tile's rows in ascending WIDTH (ties by rank) -- and, per 64 slots (one
// quartile = one work item of graph_lists_kernel), how far to the right a partner can start (reach_table_kernel's formula).
// grid = (groups, tiles per group), block = 256.
__global__ __launch_bounds__(256) void row_classes_kernel(const float4 *__restrict__ xbox, const GroupDesc *__restrict__ groups,
                                                          const uint32_t *__restrict__ group_flags, float one_minus_t,
                                                          uint16_t *__restrict__ rowperm, float *__restrict__ qreach)
{
    __shared__ float sw[256];
    __shared__ float sreach[256];
    const int g = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    if (!(group_flags[g] & kFlagRegular)) return;
    const GroupDesc gd = groups[g];
    if (t * 256 >= gd.nbox) return;
    const int s = t * 256 + tid;
    const bool valid = s < gd.nbox;
    float wd = 3.0e38f, reach = -3.0e38f;
    if (valid) { const float4 b = xbox[gd.box_off + s]; wd = b.z - b.x; reach = b.x + one_minus_t * ((b.z - b.x) + 1.0f) * 1.001f + 1.0f; }
    sw[tid] = wd;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < 256; ++j) { const float o = sw[j]; rank += (o < wd || (o == wd && j < tid)) ? 1 : 0; }
    if (valid) rowperm[gd.box_off + t * 256 + rank] = (uint16_t)s;
    sreach[rank] = reach;            // (a permutation of 0 .. 255: the invalid slots rank last)
    __syncthreads();
    float rc = sreach[tid];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) rc = fmaxf(rc, __shfl_xor(rc, d, 64));
    if ((tid & 63) == 0) qreach[qreach_slot(gd, g) + 4 * t + (tid >> 6)] = rc;
}

struct ListItem { int32_t group; int32_t tq; };      // tq = 4 * row tile + width quartile

struct GraphListsParams {
    const float4 *xbox;              // boxes in x1 order (FrameIndex::xbox)
    const uint16_t *xord;            // their box indices
    const GroupDesc *groups;
    const uint32_t *group_flags;
    const ListItem *items;
    int nitems;
    float t32, one_minus_t;
    uint32_t *row_deg;               // per rank-row: degree = cursor of its slot (zero when the kernel starts)
    const float2 *reach_table;       // .y = x1 of the first box of every 64-column block
    const uint16_t *rowperm;
    const float *qreach;
    uint16_t *adj;
    uint32_t slot_cap;
    int *status;
    int over_bits;
    uint32_t trash_off;              // byte offset of >= 256 unused bytes of the pool (where the entries of an overflowing list go)
};

// One 32-bit half of a row word -> BOTH entries of every edge: the column's box goes to the row's list (byte offset boff of the
// pool, consecutive places), the row's own box (myid) to the column's list -- at the byte offset kept in lpos[column], which a
// returning LDS add hands out and advances.  The columns' entries need no second pass over the transposed word that way (the
// two enumerations cost ~55 issue slots per half each; the extra add + store per edge ~4).
__device__ __forceinline__ void emit_half(uint16_t *__restrict__ adj, uint32_t h, uint32_t boff, const uint16_t *ids, uint32_t *lpos,
                                          const uint16_t myid)
{
    constexpr int kSlots = 4;
    if (__ballot(h != 0u) == 0ull) return;                   // (wave-uniform)
    uint32_t hk[kSlots];
    uint16_t ek[kSlots];
    int bk[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) { hk[k] = h; bk[k] = __builtin_ctz(h) & 31; h &= h - 1u; }      // (an empty slot: bit 31, unused)
#pragma unroll
    for (int k = 0; k < kSlots; ++k) ek[k] = ids[bk[k]];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        if (hk[k] != 0u) {
            *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(adj) + (boff + 2u * k)) = ek[k];
            const uint32_t o = __hip_atomic_fetch_add(&lpos[bk[k]], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(adj) + o) = myid;
        }
    }
    boff += 2u * kSlots;
    while (__ballot(h != 0u) != 0ull) {
        if (h != 0u) {
            const int b = __builtin_ctz(h);
            *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(adj) + boff) = ids[b];
            const uint32_t o = __hip_atomic_fetch_add(&lpos[b], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            *reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(adj) + o) = myid;
            boff += 2u;
            h &= h - 1u;
        }
    }
}

// The entries of one evaluated block (m = my row's word over the block's 64 columns, tcnt = entries of column `lane`, pat / ptat =
// the places the two cursors returned).  A list that would outgrow its slot latches over_bits; its entries
// land in the pool's spare slot: the host rebuilds the graph through the bit matrix.
__device__ __forceinline__ void emit_block(const uint32_t slot_cap, uint16_t *__restrict__ adj, int *__restrict__ status, const int over_bits,
                                           const uint32_t trash_off, unsigned long long m, const uint32_t tcnt, const uint32_t pat,
                                           const uint32_t ptat, const uint32_t row_slot, const uint32_t col_slot, const uint16_t *cids,
                                           const uint16_t myid, uint32_t *lpos, const int lane)
{
    // (an overflowing row still hands its box to its columns' lists -- those rows are not to blame and their lists must come out
    //  whole: the walk of an asynchronous step reads them before the host sees the flag)
    const uint32_t cnt = (uint32_t)__popcll(m);
    const bool rover = pat + cnt > slot_cap, cover = ptat + tcnt > slot_cap;
    if ((rover && cnt) || (cover && tcnt)) atomicOr(status, over_bits);
    lpos[lane] = cover ? trash_off : (col_slot * slot_cap + ptat) * 2u;
    const uint32_t boff = rover ? trash_off : (row_slot * slot_cap + pat) * 2u;
    emit_half(adj, (uint32_t)m, boff, cids, lpos, myid);
    emit_half(adj, (uint32_t)(m >> 32), boff + 2u * (uint32_t)__popc((uint32_t)m), cids + 32, lpos + 32, myid);
}

// the 64 margins pairs of one block: complement of the sign words = predicate bits, anyb = some pair in the half-ulp band
template <bool XS, bool INTS>
__device__ __forceinline__ void margin_block(const float4 brx, const float rarea, const float4 *sb, const float *sa, float t32, float t_lo,
                                             uint32_t &lo, uint32_t &hi, bool &anyb)
{
    uint32_t nr0 = 0, nq0 = 0, nr1 = 0, nq1 = 0;
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 31 - 8 * g - j;
            float mr, mq;
            pred_margins<XS, INTS>(brx, rarea, sb[k], sa[k], t32, t_lo, mr, mq);
            shl1_or_sign(nr0, mr); shl1_or_sign(nq0, mq);
        }
    }
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 31 - 8 * g - j;
            float mr, mq;
            pred_margins<XS, INTS>(brx, rarea, sb[32 + k], sa[32 + k], t32, t_lo, mr, mq);
            shl1_or_sign(nr1, mr); shl1_or_sign(nq1, mq);
        }
    }
    lo = ~nr0; hi = ~nr1;
    anyb = (nr0 != nq0) | (nr1 != nq1);
}

__global__ __launch_bounds__(64, 6) void graph_lists_kernel(const GraphListsParams prm)
{
    __shared__ float4 sbox[2][64];
    __shared__ float sarea[2][64];
    __shared__ uint16_t scord[2][64];
    __shared__ uint32_t lpos[64];
    // block b runs on XCD b % 8: a contiguous eighth of the items (whole frames) per XCD
    const int per = gridDim.x >> 3;               // (the grid is the item count rounded up to a multiple of 8)
    const int idx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (idx >= prm.nitems) return;
    const ListItem it = prm.items[idx];
    const uint32_t gf = prm.group_flags[it.group];
    if (!(gf & kFlagRegular)) return;
    const GroupDesc gd = prm.groups[it.group];
    const int B = gd.nbox, W = (B + 63) >> 6;
    const int lane = threadIdx.x;
    const int mt = it.tq >> 2, qd = it.tq & 3;
    const int s = mt * 256 + qd * 64 + lane;
    const int v = s < B ? (int)prm.rowperm[gd.box_off + s] : B;             // my row (x1 rank)
    const unsigned long long rowvalid = __ballot(v < B);
    if (rowvalid == 0ull) return;
    const float my_reach = prm.qreach[qreach_slot(gd, it.group) + it.tq];
    const float2 *rtab = prm.reach_table + reach_slot(gd, it.group);
    const float4 *xb = prm.xbox + gd.box_off;
    const uint16_t *xo = prm.xord + gd.box_off;
    const float t32 = prm.t32;
    const float t_lo = t32 * (1.0f - 4.76837158203125e-7f);
    const TransposeConsts tcs = transpose_consts(lane);
    const bool ints = (gf & kFlagU16) != 0u;      // integer pixel coordinates: x2 + 1 / y2 + 1 formed once per box
    const uint32_t slot_cap = prm.slot_cap;
    uint16_t *adj = prm.adj;

    float4 br = make_float4(0.f, 0.f, 0.f, 0.f);
    uint16_t myid = 0;
    if (v < B) { br = xb[v]; myid = xo[v]; }
    const float rarea = box_area(br);
    float4 brx = br;
    if (ints) { brx.z += 1.0f; brx.w += 1.0f; }

    // the first column block (my tile's first) goes to LDS buffer 0
    int c = mt * 4;
    {
        const int u = c * 64 + lane;
        float4 bc = make_float4(0.f, 0.f, 0.f, 0.f);
        uint16_t oc = 0;
        if (u < B) { bc = xb[u]; oc = xo[u]; }
        sarea[0][lane] = box_area(bc);
        if (ints) { bc.z += 1.0f; bc.w += 1.0f; }
        sbox[0][lane] = bc; scord[0][lane] = oc;
    }
    // the block whose atomics are in flight (row word / transposed word, the returned cursors, its column block and buffer)
    unsigned long long pm = 0ull;
    uint32_t pat = 0u, ptat = 0u, ptcnt = 0u;
    int pc = -1;
    // (where the next block starts is fetched ONE BLOCK AHEAD: a load that is looked at right away is waited for with vmcnt(0),
    //  i.e. together with the atomics issued just before it, whose round trip the block's pair tests are meant to hide)
    float fnext = (c + 1 < W) ? rtab[c + 1].y : 3.0e38f;
    for (int buf = 0; c < W; ++c, buf ^= 1) {
        const bool dtile = (c >> 2) == mt;        // my tile's own columns: masked to the columns of higher rank
        // the next block: is there one, and does it start within my reach? (blocks are sorted: the first one beyond ends the item)
        const int cn = c + 1;
        bool more = cn < W;
        float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
        uint16_t on = 0;
        const float fn = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(fnext)));
        fnext = (c + 2 < W) ? rtab[c + 2].y : 3.0e38f;
        if (more) {
            more = ((cn >> 2) == mt) || fn <= my_reach;
            const int u = cn * 64 + lane;
            if (more && u < B) { bn = xb[u]; on = xo[u]; }
        }
        const int cols_left = B - c * 64;
        const unsigned long long colvalid = cols_left >= 64 ? ~0ull : ((1ull << cols_left) - 1ull);
        uint32_t lo, hi;
        bool anyb;
        if (ints) { if (!dtile) margin_block<true, true>(brx, rarea, sbox[buf], sarea[buf], t32, t_lo, lo, hi, anyb);
                    else margin_block<false, true>(brx, rarea, sbox[buf], sarea[buf], t32, t_lo, lo, hi, anyb); }
        else { if (!dtile) margin_block<true, false>(brx, rarea, sbox[buf], sarea[buf], t32, t_lo, lo, hi, anyb);
               else margin_block<false, false>(brx, rarea, sbox[buf], sarea[buf], t32, t_lo, lo, hi, anyb); }
        if (__builtin_expect(__ballot(anyb) != 0ull, 0)) {
            // rare (~1e-6 of the pairs sit in the half-ulp band, e.g. IoU exactly 3/10): redo the block with the IEEE quotient
            lo = hi = 0u;
            for (int k = 0; k < 64; ++k) {
                float4 bk = sbox[buf][k];
                if (ints) { bk.z -= 1.0f; bk.w -= 1.0f; }        // (exact: integers)
                const bool p = pair_pred_exact_slow(br, rarea, bk, sarea[buf][k], t32);
                if (k < 32) lo |= p ? (1u << k) : 0u; else hi |= p ? (1u << (k - 32)) : 0u;
            }
        }
        unsigned long long m = (((unsigned long long)hi << 32) | lo) & colvalid;
        if (dtile) {     // column k of block c is rank c * 64 + k; mine are the ones of higher rank (every pair once, no self edge)
            const int d = v - c * 64;
            m &= d < 0 ? ~0ull : (d >= 63 ? 0ull : ~((2ull << d) - 1ull));
        }
        if (v >= B) m = 0ull;
        uint32_t tlo = (uint32_t)m, thi = (uint32_t)(m >> 32);
        wave_transpose64(tlo, thi, tcs);          // (of the MASKED words)
        // the previous block's entries: its cursors have had this block's time to arrive
        if (pc >= 0)
            emit_block(slot_cap, adj, prm.status, prm.over_bits, prm.trash_off, pm, ptcnt, pat, ptat, (uint32_t)(gd.box_off + v),
                       (uint32_t)(gd.box_off + pc * 64 + lane), scord[buf ^ 1], myid, lpos, lane);
        // this block's reservations
        {
            unsigned long long tm = 0ull;
            const int u = c * 64 + lane;
            if (u < B) tm = (((unsigned long long)thi << 32) | tlo) & rowvalid;
            const uint32_t cnt = (uint32_t)__popcll(m), tcnt = (uint32_t)__popcll(tm);
            pat = 0u; ptat = 0u;
            if (cnt) pat = atomicAdd(&prm.row_deg[gd.box_off + v], cnt);
            if (tcnt) ptat = atomicAdd(&prm.row_deg[gd.box_off + u], tcnt);
            pm = m; ptcnt = tcnt; pc = c;
        }
        if (!more) break;
        // stage the next block (the previous block's translations in that buffer have just been used)
        sarea[buf ^ 1][lane] = box_area(bn);
        if (ints) { bn.z += 1.0f; bn.w += 1.0f; }
        sbox[buf ^ 1][lane] = bn; scord[buf ^ 1][lane] = on;
    }
    // the last block's entries (its translations sit in the buffer it was evaluated from)
    if (pc >= 0)
        emit_block(slot_cap, adj, prm.status, prm.over_bits, prm.trash_off, pm, ptcnt, pat, ptat, (uint32_t)(gd.box_off + v),
                   (uint32_t)(gd.box_off + pc * 64 + lane), scord[(pc - mt * 4) & 1], myid, lpos, lane);
}

}  // namespace vdet
