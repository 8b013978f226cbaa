// valu_bench.hip -- VALU issue-rate microbenchmark for MI355X (gfx950): settles the ceiling the VALU-bound
// kernels of libvdet_hip (K1s iou_bits_sym, the walk, the sort) are priced against.
//   hipcc --offload-arch=gfx950 -O3 -o devtools/valu_bench devtools/valu_bench.hip && devtools/valu_bench
// Every variant runs ITER iterations of 16 INDEPENDENT instructions per lane (8 accumulators x 2), at 8 and at
// 4 waves per SIMD on all CUs, and reports lane-operations per second (one wave64 instruction = 64 lane-ops;
// a packed f32 instruction counts 128) next to the guide's figure 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 7.86e13.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

// mode: 0 v_fma_f32  1 v_pk_fma_f32  2 v_max_f32  3 v_add_f32  4 v_pk_add_f32  5 v_pk_mul_f32
//       6 v_cmp_ge_f32 + v_addc_co_u32 (K1s's bit accumulator)  7 v_add_u32  8 v_and_b32/v_lshlrev mix  9 v_min_f32+v_max_f32
//       10 v_max_i32  11 v_min_u32  12 v_cmp_ge_f32 into 4 SGPR pairs  13 v_mul_f32  14 v_max3_f32  15 v_cndmask_b32  16 v_min_f32
//       17 v_med3_f32  18 v_sub_f32
template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float *out, float seed)
{
    float a[8];
    float2 p[8];
    unsigned u[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = make_float2(a[i], a[i] * 0.5f); u[i] = (unsigned)threadIdx.x * 2654435761u + i; }
    const float b = 1.0000001f, c = 1e-9f;
    const float2 b2 = make_float2(b, b), c2 = make_float2(c, c);
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 2) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 3) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 4) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 5) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 6) {
#define OP(i) asm volatile("v_cmp_ge_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(c) : "vcc");
            REP8(OP)
#undef OP
        } else if (MODE == 7) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 8) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 5, %0\n v_and_b32 %0, 0x3ff, %0" : "+v"(u[i]));
            REP8(OP)
#undef OP
        } else if (MODE == 9) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1\n v_min_f32 %0, %0, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP8(OP)
#undef OP
        } else if (MODE == 10) {
#define OP(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 11) {
#define OP(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 12) {
            unsigned long long m0, m1, m2, m3;
            asm volatile("v_cmp_ge_f32 %0, %4, %8\n v_cmp_ge_f32 %1, %5, %8\n v_cmp_ge_f32 %2, %6, %8\n v_cmp_ge_f32 %3, %7, %8\n"
                         "v_cmp_ge_f32 %0, %5, %8\n v_cmp_ge_f32 %1, %6, %8\n v_cmp_ge_f32 %2, %7, %8\n v_cmp_ge_f32 %3, %4, %8\n"
                         "v_cmp_ge_f32 %0, %6, %8\n v_cmp_ge_f32 %1, %7, %8\n v_cmp_ge_f32 %2, %4, %8\n v_cmp_ge_f32 %3, %5, %8\n"
                         "v_cmp_ge_f32 %0, %7, %8\n v_cmp_ge_f32 %1, %4, %8\n v_cmp_ge_f32 %2, %5, %8\n v_cmp_ge_f32 %3, %6, %8\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(c));
            u[0] += (unsigned)(m0 ^ m1 ^ m2 ^ m3);
        } else if (MODE == 13) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 14) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 15) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 16) {
#define OP(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 17) {
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 19) {
#define OP(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 20) {
#define OP(i) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 21) {
#define OP(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 22) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 23) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 24) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 25) {
#define OP(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 26) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 27) {
#define OP(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 29) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 30) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "s"(c));
            REP8(OP) REP8(OP)
#undef OP
        } else if (MODE == 28) {
#define OP(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            REP8(OP) REP8(OP)
#undef OP
        } else {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            REP8(OP) REP8(OP)
#undef OP
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
double run(int blocks_per_cu, int ncu, float *d_out, int lane_ops_per_instr)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grid = blocks_per_cu * ncu;
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = 16.0 * ITER * (double)grid * 4 /*waves*/ * 5;
    return instr * lane_ops_per_instr / (ms * 1e-3);
}

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    float *d_out;
    CHK(hipMalloc(&d_out, 4096));
    printf("# %s, %d CUs, clock %d MHz; guide figure 256*4*32*2.4e9 = 7.86e13 lane-ops/s\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    printf("variant,waves_per_simd,lane_ops_per_s,frac_of_7.86e13\n");
    for (int bpc : {8, 2}) {
        const int wps = bpc;    // 256-thread blocks: bpc blocks/CU = bpc waves per SIMD
#define ROW(NAME, MODE, LO) { const double r = run<MODE>(bpc, ncu, d_out, LO); printf("%s,%d,%.4g,%.3f\n", NAME, wps, r, r / 7.86e13); }
        ROW("v_fma_f32", 0, 64)
        ROW("v_pk_fma_f32(x2 lanes-ops)", 1, 128)
        ROW("v_max_f32", 2, 64)
        ROW("v_add_f32", 3, 64)
        ROW("v_pk_add_f32(x2)", 4, 128)
        ROW("v_pk_mul_f32(x2)", 5, 128)
        ROW("v_cmp_ge_f32+v_addc_co_u32", 6, 64)
        ROW("v_add_u32", 7, 64)
        ROW("v_lshrrev_b32+v_and_b32", 8, 64)
        ROW("v_max_f32+v_min_f32", 9, 64)
        ROW("v_max_i32", 10, 64)
        ROW("v_min_u32", 11, 64)
        ROW("v_cmp_ge_f32->sgpr", 12, 64)
        ROW("v_mul_f32", 13, 64)
        ROW("v_max3_f32", 14, 64)
        ROW("v_cndmask_b32", 15, 64)
        ROW("v_min_f32", 16, 64)
        ROW("v_med3_f32", 17, 64)
        ROW("v_sub_f32", 18, 64)
        ROW("v_pk_max_i16(x2)", 19, 128)
        ROW("v_pk_min_i16(x2)", 20, 128)
        ROW("v_pk_sub_i16(x2)", 21, 128)
        ROW("v_mul_u32_u24", 22, 64)
        ROW("v_mad_u32_u24", 23, 64)
        ROW("v_cvt_f32_u32", 24, 64)
        ROW("v_pk_mad_u16(x2)", 25, 128)
        ROW("v_mul_lo_u32", 26, 64)
        ROW("v_sub_u32", 27, 64)
        ROW("v_pk_max_u16(x2)", 28, 128)
        ROW("v_alignbit_b32", 29, 64)
        ROW("v_max_f32(sgpr operand)", 30, 64)
#undef ROW
    }
    CHK(hipFree(d_out));
    return 0;
}
