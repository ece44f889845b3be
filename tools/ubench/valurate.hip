// Issue rate of a few VALU instructions on gfx950 under full occupancy (tuning aid; not part of the product).
// build: hipcc --offload-arch=gfx950 -O2 -o valurate valurate.hip
// 8192 waves (8 per SIMD), each issues REPS x 16 instructions of one kind on 8 independent accumulators; the kernel time gives
// cycles per wave-instruction per SIMD at an assumed 2.4 GHz -- the RATIOS between the kinds are what counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t *sink, int reps)
{
    uint32_t a[8], b = threadIdx.x * 2654435761u | 1u, c = threadIdx.x + 12345u;
    double d[8], e = 1.0 + threadIdx.x * 1e-9, f = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x + i; d[i] = (double)i; }
    const double se = __builtin_amdgcn_readfirstlane(reps) * 1.0;
    const uint32_t sb = __builtin_amdgcn_readfirstlane(reps * 3);
    for (int r = 0; r < reps; r++) {
        if (KIND == 0) {
#define S(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 1) {
#define S(i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 2) {
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(e), "v"(f));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 3) {
#define S(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 4) {
#define S(i) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 5) {
#define S(i) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 6) {
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(b), "v"(c) : "vcc");
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 7) {
#define S(i) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 8) {
#define S(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 10) {
#define S(i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(d[i]) : "v"(e), "v"(f));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 11) {
#define S(i) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 12) {
#define S(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 13) {
#define S(i) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 14) {
#define S(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "s"(se), "v"(f));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 15) {
#define S(i) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "s"(sb), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 16) {
#define S(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(e));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 17) {
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 20) {
#define S(i) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 21) {
#define S(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 22) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 23) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 24) {
#define S(i) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 25) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 26) {
#define S(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 27) {
#define S(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 28) {
#define S(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 29) {
#define S(i) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 30) {
#define S(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 31) {
#define S(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 32) {
#define S(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 33) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 34) {
#define S(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 35) {
#define S(i) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 36) {
#define S(i) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 37) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d[i]) : "v"(e));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 38) {
#define S(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 39) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 40) {
#define S(i) asm volatile("v_lshl_or_b32 %0, %1, 16, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 41) {
#define S(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 42) {
#define S(i) asm volatile("v_med3_i32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 43) {
#define S(i) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 44) {
#define S(i) asm volatile("v_xad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 45) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 46) {
#define S(i) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(e));
            REP8(S) REP8(S)
#undef S
        }
        else if (KIND == 9) {
#define S(i) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            REP8(S) REP8(S)
#undef S
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x ^= a[i] ^ (uint32_t)__double2loint(d[i]);
    sink[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int KIND> static void run(const char *name, uint32_t *sink)
{
    const int reps = 16384, grid = 2048;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, sink, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, sink, reps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: 8 waves x reps x 16 instructions
    const double cyc = ms * 1e-3 * 2.4e9 / (8.0 * reps * 16.0);
    printf("%-18s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc);
}

int main()
{
    uint32_t *sink;
    CK(hipMalloc(&sink, 2048 * 256 * 4));
    for (int w = 0; w < 3; w++) run<3>("v_add_u32 (warm-up)", sink);
    run<3>("v_add_u32", sink);
    run<13>("v_mul_i32_i24 e32", sink);
    run<12>("v_fma_f32", sink);
    run<17>("v_fmac_f32 e32", sink);
    run<15>("v_mad_i32_i24 s,v,v", sink);
    run<11>("v_dot2c_i32_i16", sink);
    run<10>("v_fmac_f64 e32", sink);
    run<14>("v_fma_f64 s,v,v", sink);
    run<16>("v_add_f64", sink);
    run<0>("v_mad_i32_i24", sink);
    run<1>("v_dot2_i32_i16", sink);
    run<9>("v_dot4_i32_i8", sink);
    run<4>("v_perm_b32", sink);
    run<5>("v_sad_u32", sink);
    run<2>("v_fma_f64", sink);
    run<7>("v_floor_f64", sink);
    run<8>("v_cvt_f64_i32", sink);
    run<6>("v_mad_u64_u32", sink);
    run<20>("v_alignbit_b32", sink);
    run<21>("v_add3_u32", sink);
    run<22>("v_lshl_add_u32", sink);
    run<23>("v_bfe_u32", sink);
    run<24>("v_and_or_b32", sink);
    run<25>("v_cndmask_b32", sink);
    run<26>("v_lshrrev_b32", sink);
    run<27>("v_ffbl_b32", sink);
    run<28>("v_bfrev_b32", sink);
    run<29>("v_max_u32", sink);
    run<30>("v_mov_b32 dpp shr1", sink);
    run<31>("v_cvt_f32_i32", sink);
    run<32>("v_mul_f32", sink);
    run<33>("v_cvt_f64_f32", sink);
    run<34>("v_xor_b32", sink);
    run<35>("v_ashrrev_i32 v,v", sink);
    run<36>("v_add_u32 sdwa", sink);
    run<37>("v_pk_fma_f32", sink);
    run<38>("v_mul_lo_u32", sink);
    run<39>("v_mad_u32_u24", sink);
    run<40>("v_lshl_or_b32", sink);
    run<41>("v_sub_u32", sink);
    run<42>("v_med3_i32", sink);
    run<43>("v_pk_mad_i16", sink);
    run<44>("v_xad_u32", sink);
    run<45>("v_mov_b32", sink);
    run<46>("v_mov_b64", sink);
    return 0;
}
