// adjrows_kernels.hpp -- K2r: bit rows -> adjacency lists of REGULAR frames, one workgroup per 64-row strip (round 6).
//
// adj_build_kernel (nms_kernels.hpp) gives every row of the bit matrix to one LANE: the lane takes its ~42 window words apart
// bit by bit, 98 dependent extraction steps on average, into a 32 KB LDS stage that the block then copies out -- 8 waves per CU
// whose lanes run serial chains of very different lengths (1.4 ms per config-2 video for 2.3 GB of traffic: neither bandwidth
// nor issue, latency at 2 waves per SIMD).  Here the work is turned by 90 degrees:
//   * a block owns the 64 rows of one word-row `wr`.  Which words of those rows EXIST is a property of the strip, not of the
//     row: block (wr, c) of the predicate matrix was evaluated -- and written -- by iou_bits_sym_kernel iff its reach test passed
//     (reach_table_kernel), the same test for all 64 rows.  The existing column words are listed once (~56 of 157 at config 2);
//   * 32 of them at a time are staged in LDS (lane = row: one coalesced 512-byte load per word, all of a wave's loads in flight
//     together) next to the x1-rank -> box-index translation of their 64 columns;
//   * extraction: a wave owns 16 of the strip's rows and takes a pass of 16 words in four steps of FOUR NEIGHBOURING words x 16 rows,
//     lane = (word, row).  A lane reserves its word's entries in the row's list with one returning LDS atomic on the row's cursor;
//   * the strip's 64 lists are built in a 16 KB LDS stage at their offsets inside the strip's slab and leave in one coalesced copy
//     (a slab that does not fit -- a frame several times denser than config 2 -- is written entry by entry).
// Measured on the way (config-2 video, ms per video; adj_build_kernel 1.38): half a wave per row, lane = word, places from a DPP
// prefix sum, entries stored straight to the pool 1.17; words of similar density per step, still straight to the pool 1.24 (both
// bound by ~1 000 partial-line write requests per wave and pass); + the LDS stage 0.97; + the block's start-up loads in one go,
// two entries per turn 0.91; + a strip's words in one round trip 0.93; + slab offsets from a pre-pass instead of an atomic per
// strip 0.86 (+ 0.09 for the pre-pass).  One block per 256-row TILE, strips one after the other with the next strip's words in
// flight: 1.48 (a quarter of the blocks, the same number of atomics).  Where the 0.86 go (timing experiments: an extra launch
// with a piece left out in front of the real one): start-up 0.10, word loads + staging + barriers 0.37 (2.1 GB of existing words
// at 5.5 TB/s: the HBM floor of reading the bit matrix), the bit loops 0.29, the copy-out 0.06.
// 32 KB of LDS per block: 5 blocks (20 waves) per CU.  List order inside a row differs from adj_build_kernel's (lists are sets: the
// walk ORs them into a mask, the re-scoring scans them with an index tie-break); offsets, degrees, padding (multiples of 8
// entries, copies of an entry of the list) and the walk's records are the same.  Row degrees come from iou_bits_sym_kernel's
// counters as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nms_kernels.hpp"
#include "binsort_kernels.hpp"      // wave_incl_scan_u32

namespace vdet {

constexpr int kRowsChunk = 16;      // existing words staged per pass
constexpr int kRowsStage = 8192;    // u16 entries of the strip's slab staged in LDS (16 KB; 64 lists of ~104 entries at config 2)
constexpr int kRowsStride = 72;     // u64 words between two staged words: 64 rows + padding (lane (a, b) of a step reads bank 16 a + 2 b)

// Where every strip's slab starts, WITHOUT an atomic per strip: with one returning atomicAdd on the pool counter per block, the
// ~1 300 resident blocks queue on ONE address of the L2's atomic unit (~13 ns each, measured: a strip's block sat 16 of its 22 us
// in that queue -- 0.61 of the kernel's 0.88 ms per video were its start-up).  strip_totals_kernel sums the padded list lengths
// of every strip of the launch (degrees are final once iou_bits_sym_kernel is done), strip_scan_kernel -- one block -- turns them
// into offsets behind ONE reservation, and the lists of a video land at reproducible places.
__global__ __launch_bounds__(256) void strip_totals_kernel(const GroupDesc *__restrict__ groups, const TileDesc *__restrict__ tiles,
                                                           int nstrips, const uint32_t *__restrict__ row_deg,
                                                           const uint32_t *__restrict__ group_flags, uint32_t *__restrict__ totals)
{
    const int lane = threadIdx.x & 63;
    const int sidx = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (sidx >= nstrips) return;
    const TileDesc td = tiles[sidx >> 2];
    const GroupDesc gd = groups[td.group];
    const int v = (td.row_tile * (kRowsPerTile / 64) + (sidx & 3)) * 64 + lane;
    uint32_t t = 0u;
    if ((group_flags[td.group] & kFlagRegular) && v < gd.nbox) t = (row_deg[gd.box_off + v] + 7u) & ~7u;
    t = wave_incl_scan_u32(t);
    if (lane == 63) totals[sidx] = t;
}

__global__ __launch_bounds__(1024) void strip_scan_kernel(const uint32_t *__restrict__ totals, int n, unsigned long long *__restrict__ offsets,
                                                          unsigned long long *__restrict__ pool_used)
{
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long sbase;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (n + 1023) / 1024;
    const int i0 = tid * per, i1 = min(n, i0 + per);
    unsigned long long sum = 0ull;
    for (int i = i0; i < i1; ++i) sum += totals[i];
    unsigned long long incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0ull;
        for (int k = 0; k < 16; ++k) { const unsigned long long x = wsum[k]; wsum[k] = run; run += x; }
        sbase = atomicAdd(pool_used, run);
    }
    __syncthreads();
    unsigned long long run = sbase + wsum[w] + incl - sum;
    for (int i = i0; i < i1; ++i) { offsets[i] = run; run += totals[i]; }
}

__global__ __launch_bounds__(256) void adj_rows_kernel(const GroupDesc *__restrict__ groups, const TileDesc *__restrict__ tiles,
                                                       const uint64_t *__restrict__ bits, const uint32_t *__restrict__ row_deg,
                                                       uint2 *__restrict__ row_meta, uint16_t *__restrict__ adj,
                                                       const unsigned long long *__restrict__ slab_off, unsigned long long pool_cap,
                                                       int *__restrict__ status, const uint32_t *__restrict__ group_flags,
                                                       const float4 *__restrict__ xbox_all, const uint16_t *__restrict__ xord_all,
                                                       int pool_bits, WalkMeta *__restrict__ wmeta, const float2 *__restrict__ reach_table,
                                                       uint4 *__restrict__ wmeta16)
{
    __shared__ float2 srt[kMaxWordRows];
    __shared__ uint16_t scol[kMaxWordRows];
    __shared__ unsigned long long sb[kRowsChunk * kRowsStride];       // word e of row r at e * kRowsStride + r
    __shared__ uint16_t sx[kRowsChunk * 64];                 // box index of column k of word e at e * 64 + k
    __shared__ uint32_t srow_off[64], srow_deg[64], scur[64], spad[64];
    __shared__ __attribute__((aligned(16))) uint16_t sstage[kRowsStage];
    __shared__ int s_ne, s_over2;
    __shared__ uint32_t s_total, s_base;

    const TileDesc td = tiles[blockIdx.x >> 2];
    if (!(group_flags[td.group] & kFlagRegular)) return;     // irregular frames: adj_build_kernel (zero-union tags, no x-index)
    const GroupDesc gd = groups[td.group];
    const int B = gd.nbox, W = (B + 63) >> 6;
    const int wr = td.row_tile * (kRowsPerTile / 64) + (int)(blockIdx.x & 3);
    if (wr >= W) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint16_t *tr = xord_all + gd.box_off;
    const uint64_t *gbits = bits + gd.bits_off;
    const int v = wr * 64 + lane;                            // the row of lane `lane` (staging) / of thread `lane` of wave 1 (allocation)

    // everything that only needs the group's descriptor is requested up front, in one go: the reach table (all threads) and,
    // by wave 1, the rows' degrees, box numbers and boxes -- the block's start is a chain of memory round trips otherwise
    uint32_t deg = 0u;
    int vo = 0;
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned long long sbase0 = 0ull;
    if (w == 1) sbase0 = slab_off[blockIdx.x];
    if (w == 1 && v < B) {
        deg = row_deg[gd.box_off + v];
        vo = (int)tr[v];
        bx = xbox_all[gd.box_off + v];
    }
    {
        const float2 *rt = reach_table + reach_slot(gd, td.group);
        for (int i = tid; i < W; i += 256) srt[i] = rt[i];
    }
    if (w == 1) {
        // one slab for the strip's 64 lists (padded to multiples of 8 entries), one atomic; the rows' records
        const uint32_t tot_al = (deg + 7u) & ~7u;
        const uint32_t incl = wave_incl_scan_u32(tot_al);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const unsigned long long base = sbase0;                 // strip_scan_kernel: no atomic here
        const bool over = base + total > pool_cap || base + total > 0xFFFFFFFFull;
        if (over && lane == 0) atomicOr(status, pool_bits);
        const uint32_t p = over ? 0u : (uint32_t)base + incl - tot_al;
        // (the lists are built at their offset INSIDE the slab: in the LDS stage when the slab fits, else in the pool itself)
        srow_off[lane] = incl - tot_al; srow_deg[lane] = over ? 0u : deg; scur[lane] = 0u; spad[lane] = 0u;
        if (lane == 0) { s_total = total; s_base = (uint32_t)base; s_over2 = over ? 1 : 0; }
        if (v < B) {
            row_meta[gd.box_off + vo] = over ? make_uint2(0u, 0u) : make_uint2(p, deg);
            if (wmeta) {
                wmeta[gd.box_off + vo].box = over ? make_float4(0.f, 0.f, 0.f, 0.f) : bx;
                wmeta[gd.box_off + vo].row = over ? make_uint4(0u, 0u, 0u, 0u) : make_uint4(p, deg, 0u, 0u);
            }
            // frames of integer coordinates in [0, 65535]: the same record in 16 bytes (the walk's one load per candidate)
            if (wmeta16 && (group_flags[td.group] & kFlagU16))
                wmeta16[gd.box_off + vo] = over ? make_uint4(0u, 0u, 0u, 0u)
                                                : make_uint4((uint32_t)bx.x | ((uint32_t)bx.y << 16), (uint32_t)bx.z | ((uint32_t)bx.w << 16), p, deg);
        }
    }
    __syncthreads();
    if (s_over2) return;
    {
        // the strip's existing column words, ascending: block (min, max) of the upper triangle was in reach.  Every wave makes the
        // list (the same values to the same places): no second barrier, a wave reads what it wrote itself
        const float2 me = srt[wr];
        int n = 0;
        for (int c0 = 0; c0 < W; c0 += 64) {
            const int c = c0 + lane;
            const float2 sc = srt[min(c, W - 1)];
            const bool ex = c < W && (c == wr || (c > wr ? sc.y <= me.x : me.y <= sc.x));
            const unsigned long long em = __ballot(ex);
            if (ex) scol[n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u))] = (uint16_t)c;
            n += __popcll(em);
        }
        s_ne = n;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const int ne = __builtin_amdgcn_readfirstlane(s_ne);
    constexpr int WPW = kRowsChunk / 4;        // words a wave stages per pass
    // extraction: the wave owns rows 16 w .. 16 w + 15 of the strip; lane = (word offset a, row b)
    const int xa = lane >> 4, xrow = 16 * w + (lane & 15);

    const uint32_t my_off = srow_off[xrow];
    const uint32_t total = s_total;
    const bool staged = total <= (uint32_t)kRowsStage;               // (block-uniform; a denser strip writes its entries one by one)
    uint16_t *slab = adj + s_base;
    // The words of up to kRowsSuper = 64 existing columns are requested TOGETHER, 16 per wave into registers (unconditional, clamped
    // addresses): ONE memory round trip per strip at config 2 (~56 existing words).  With a round trip per 16-word pass the
    // block spent most of its 22 us waiting at barriers for 8 loads per wave (measured: 0.91 ms per video, VALU issue 13 %).
    // The passes then only talk through LDS (lds_only_barrier: nothing in flight is drained).
    constexpr int kRowsSuper = 4 * kRowsChunk;
    for (int s0 = 0; s0 < ne; s0 += kRowsSuper) {
        unsigned long long mw[4 * WPW];
        uint16_t xw[4 * WPW];
#pragma unroll
        for (int j = 0; j < 4 * WPW; ++j) {
            const int q = j / WPW, i = j % WPW;
            const int c = (int)scol[min(s0 + kRowsChunk * q + w + 4 * i, ne - 1)];
            mw[j] = gbits[bit_word(B, min(v, B - 1), c)];
            xw[j] = tr[min(c * 64 + lane, B - 1)];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e0 = s0 + kRowsChunk * q;
            if (e0 >= ne) break;                                      // (block-uniform)
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                const int el = w + 4 * i;
                sb[el * kRowsStride + lane] = (e0 + el < ne && v < B) ? mw[q * WPW + i] : 0ull;
                sx[el * 64 + lane] = xw[q * WPW + i];
            }
            lds_only_barrier();
            // Steps of FOUR NEIGHBOURING words x 16 rows.  A row's neighbours crowd the words around its own x1 rank and the 64
            // rows of a strip share that centre, so the density of a word is a property of its position in the pass: the 64 lanes
            // of a step hold words of similar density and the bit loop below runs about as long for all of them.  A lane's entries
            // go to consecutive places of its row's list, reserved with one returning LDS atomic on the row's cursor.
#pragma unroll 1
            for (int i = 0; i < kRowsChunk / 4; ++i) {
                const int el = 4 * i + xa;
                unsigned long long m = sb[el * kRowsStride + xrow];
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (__ballot(cnt != 0u) == 0ull) continue;              // (wave-uniform)
                const uint16_t *sxw = sx + el * 64;
                uint32_t pos = 0u;
                if (cnt != 0u) {
                    const uint32_t at = atomicAdd(&scur[xrow], cnt);
                    if (at == 0u) spad[xrow] = sxw[__ffsll(m) - 1];      // (the list's first entry: what its padding repeats)
                    pos = my_off + at;
                }
                if (staged) {
                    // two entries per turn: their translations are read together (a turn is an LDS round trip, not its instructions)
                    while (__ballot(m != 0ull) != 0ull) {
                        if (m != 0ull) {
                            const unsigned long long m1 = m & (m - 1ull);
                            const uint16_t v0 = sxw[__ffsll(m) - 1];
                            const uint16_t v1 = sxw[m1 ? __ffsll(m1) - 1 : 0];
                            sstage[pos] = v0;
                            if (m1) sstage[pos + 1] = v1;
                            pos += m1 ? 2u : 1u;
                            m = m1 & (m1 - 1ull);
                        }
                    }
                } else {
                    while (__ballot(m != 0ull) != 0ull) {
                        if (m != 0ull) {
                            slab[pos++] = sxw[__ffsll(m) - 1];
                            m &= m - 1ull;
                        }
                    }
                }
            }
            lds_only_barrier();
        }
    }
    __syncthreads();
    // every list is padded to a multiple of 8 entries with copies of one of its entries (the packed walk applies whole pieces;
    // a second OR of the same bit is harmless)
    if (tid < 64) {
        const uint32_t deg = srow_deg[tid];
        const uint32_t p = srow_off[tid] + deg, padv = spad[tid];
        const uint32_t npad = ((deg + 7u) & ~7u) - deg;
        if (deg != 0u)
            for (uint32_t k = 0; k < npad; ++k) { if (staged) sstage[p + k] = (uint16_t)padv; else slab[p + k] = (uint16_t)padv; }
    }
    if (staged) {   // the strip's slab leaves in one coalesced copy (lists are multiples of 8 entries: whole 16-byte groups)
        __syncthreads();
        const uint4 *s4 = reinterpret_cast<const uint4 *>(sstage);
        uint4 *d4 = reinterpret_cast<uint4 *>(slab);
        for (uint32_t i = tid; i < (total >> 3); i += 256) d4[i] = s4[i];
    }
}

// The direct lists' second half (iou_bits_sym_kernel<true, true> wrote the entries into fixed slots and left every row's degree in
// its cursor): pad each list to a multiple of 8 entries (with the row's own box) and write the rows' records (row_meta,
// the walk's 32-byte and 16-byte records) -- exactly what adj_rows_kernel's wave 1 writes.  One thread per row, grid = the
// batch's tiles.  A degree above the slot's capacity latches `over_bits` (the host rebuilds the graph through the bit matrix).
__global__ __launch_bounds__(256) void adj_finish_kernel(const GroupDesc *__restrict__ groups, const TileDesc *__restrict__ tiles,
                                                         const uint32_t *__restrict__ row_deg, uint2 *__restrict__ row_meta,
                                                         uint16_t *__restrict__ adj, uint32_t slot_cap, int *__restrict__ status,
                                                         const uint32_t *__restrict__ group_flags, const float4 *__restrict__ xbox_all,
                                                         const uint16_t *__restrict__ xord_all, int over_bits,
                                                         WalkMeta *__restrict__ wmeta, uint4 *__restrict__ wmeta16)
{
    const TileDesc td = tiles[blockIdx.x];
    const uint32_t gf = group_flags[td.group];
    if (!(gf & kFlagRegular)) return;
    const GroupDesc gd = groups[td.group];
    const int v = td.row_tile * kRowsPerTile + (int)threadIdx.x;
    if (v >= gd.nbox) return;
    const uint32_t row = (uint32_t)(gd.box_off + v);
    const uint32_t deg = row_deg[row];
    const int vo = (int)xord_all[row];
    const float4 bx = xbox_all[row];
    const bool over = deg > slot_cap;
    if (over) atomicOr(status, over_bits);
    const uint32_t p = over ? 0u : row * slot_cap;
    const uint32_t d = over ? 0u : deg;
    // padding = the row's OWN box: the walk applies whole 16-byte pieces, and a survivor that marks itself dead after it has been
    // kept changes nothing (every candidate is looked at once); every other reader of a list stops at its degree
    if (d & 7u) {
        uint16_t *l = adj + (uint64_t)p;
        for (uint32_t k = d; k < ((d + 7u) & ~7u); ++k) l[k] = (uint16_t)vo;
    }
    row_meta[gd.box_off + vo] = make_uint2(p, d);
    const bool rec16 = wmeta16 && (gf & kFlagU16);       // (the walk takes the 16-byte record of such a frame: no 32-byte one)
    if (wmeta && !rec16) {
        wmeta[gd.box_off + vo].box = over ? make_float4(0.f, 0.f, 0.f, 0.f) : bx;
        wmeta[gd.box_off + vo].row = make_uint4(p, d, 0u, 0u);
    }
    if (rec16)
        wmeta16[gd.box_off + vo] = over ? make_uint4(0u, 0u, 0u, 0u)
                                        : make_uint4((uint32_t)bx.x | ((uint32_t)bx.y << 16), (uint32_t)bx.z | ((uint32_t)bx.w << 16), p, d);
}

// (direct lists, failure path only) the pool the bit-matrix path needs for the same graph: the padded degrees, summed
__global__ __launch_bounds__(256) void deg_sum_kernel(const uint32_t *__restrict__ row_deg, int64_t n, unsigned long long *__restrict__ out)
{
    unsigned long long s = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (row_deg[i] + 7u) & ~7u;
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// (direct lists) the dynamic part of the pool -- the lists of irregular frames, adj_build_kernel -- starts behind the fixed slots
__global__ void pool_start_kernel(unsigned long long *pool_used, unsigned long long first) { *pool_used = first; }

}  // namespace vdet
