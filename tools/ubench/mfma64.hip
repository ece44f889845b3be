// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout, rounding order, and cost of a dependent chain (tuning aid; not part of
// the product).  build: hipcc --offload-arch=gfx950 -O2 -o mfma64 mfma64.hip
//  1. layout: 64 launches in one kernel -- A one-hot in lane la, B[lane] = lane + 1, C = 0: the output lanes that light up and
//     their values tell which (A lane, B lane) pairs meet in which output lane.
//  2. order: random operands with wide exponent spread; the host composes the candidates (fma chain over k ascending /
//     descending, pairwise, unfused) from the layout of step 1 and counts bit-exact matches.
//  3. cost: cycles per MFMA in a dependent chain, alone and with two LDS reads per MFMA.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_layout(double *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; la++) {
        const double a = lane == la ? 1.0 : 0.0, b = (double)(lane + 1);
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[la * 64 + lane] = d;
    }
}

__global__ void __launch_bounds__(64) k_values(const double *A, const double *B, const double *C, double *D, int n)
{
    const int lane = threadIdx.x;
    for (int i = 0; i < n; i++)
        D[i * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[i * 64 + lane], B[i * 64 + lane], C[i * 64 + lane], 0, 0, 0);
}

__global__ void __launch_bounds__(64) k_chain(uint64_t *cyc, double *sink, int reps)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9, acc = 0.0;
    const uint64_t t0 = clock64();
    for (int i = 0; i < reps; i++) {
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc, 0, 0, 0);
    }
    const uint64_t t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc;
}

__global__ void __launch_bounds__(256) k_chain_lds(uint64_t *cyc, double *sink, int reps)
{
    __shared__ double buf[4][1024 + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 1024 + 64; i += 64) buf[wv][i] = 1.0 + i * 1e-9;
    __syncthreads();
    double acc = 0.0;
    const uint64_t t0 = clock64();
    for (int r = 0; r < reps; r++) {
        const double *p = &buf[wv][(lane & 15) + (lane >> 4)];
        for (int i = 0; i < 1024; i += 16) {
            const double a0 = p[i], b0 = p[i + 3], a1 = p[i + 4], b1 = p[i + 7], a2 = p[i + 8], b2 = p[i + 11], a3 = p[i + 12], b3 = p[i + 15];
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b3, acc, 0, 0, 0);
        }
    }
    const uint64_t t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
    sink[(blockIdx.x * 4 + wv) * 64 + lane] = acc;
}

template <int NV> __global__ void __launch_bounds__(64) k_chain_valu(uint64_t *cyc, double *sink, int reps)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-9, b = 1.0 - lane * 1e-9, acc = 0.0;
    uint32_t x = lane, y = lane * 3;
    const uint64_t t0 = clock64();
    for (int i = 0; i < reps * 4; i++) {
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; v++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
    const uint64_t t1 = clock64();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + lane] = acc + x;
}

static double rnd_wide()
{
    const double m = 1.0 + (double)(rand() & 0xFFFFFF) / 16777216.0 + (double)(rand() & 0xFFFFFF) / 16777216.0 / 16777216.0;
    const int e = (rand() % 61) - 30;
    return (rand() & 1 ? -1.0 : 1.0) * ldexp(m, e);
}

int main()
{
    double *d_out;
    CK(hipMalloc(&d_out, 64 * 64 * 8));
    k_layout<<<1, 64>>>(d_out);
    static double lay[64][64];
    CK(hipMemcpy(lay, d_out, sizeof lay, hipMemcpyDeviceToHost));
    // pairs[o][k] = (A lane, B lane) that meet in output lane o
    static int pa[64][8], pb[64][8], np[64];
    memset(np, 0, sizeof np);
    for (int la = 0; la < 64; la++)
        for (int o = 0; o < 64; o++)
            if (lay[la][o] != 0.0) { const int lb = (int)lay[la][o] - 1; if (np[o] < 8) { pa[o][np[o]] = la; pb[o][np[o]] = lb; } np[o]++; }
    printf("layout: output lane <- (A lane, B lane) pairs in A-lane order\n");
    for (int o = 0; o < 64; o++) {
        printf("  D[%2d] <-", o);
        for (int k = 0; k < np[o] && k < 8; k++) printf(" (%2d,%2d)", pa[o][k], pb[o][k]);
        printf("\n");
    }
    // rounding order
    const int N = 4096;
    double *hA = (double *)malloc(N * 64 * 8), *hB = (double *)malloc(N * 64 * 8), *hC = (double *)malloc(N * 64 * 8), *hD = (double *)malloc(N * 64 * 8);
    srand(12345);
    for (int i = 0; i < N * 64; i++) { hA[i] = rnd_wide(); hB[i] = rnd_wide(); hC[i] = rnd_wide(); }
    double *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, N * 64 * 8)); CK(hipMalloc(&dB, N * 64 * 8)); CK(hipMalloc(&dC, N * 64 * 8)); CK(hipMalloc(&dD, N * 64 * 8));
    CK(hipMemcpy(dA, hA, N * 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, N * 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, hC, N * 64 * 8, hipMemcpyHostToDevice));
    k_values<<<1, 64>>>(dA, dB, dC, dD, N);
    CK(hipMemcpy(hD, dD, N * 64 * 8, hipMemcpyDeviceToHost));
    long asc = 0, desc = 0, unf = 0, pairw = 0, tot = 0;
    for (int i = 0; i < N; i++)
        for (int o = 0; o < 64; o++) {
            if (np[o] != 4) continue;
            double p[4], q[4];
            for (int k = 0; k < 4; k++) { p[k] = hA[i * 64 + pa[o][k]]; q[k] = hB[i * 64 + pb[o][k]]; }
            const double c = hC[i * 64 + o], d = hD[i * 64 + o];
            double x = c; for (int k = 0; k < 4; k++) x = fma(p[k], q[k], x);
            double y = c; for (int k = 3; k >= 0; k--) y = fma(p[k], q[k], y);
            volatile double z = c; for (int k = 0; k < 4; k++) { volatile double t = p[k] * q[k]; z = z + t; }
            const double w = fma(p[0], q[0], fma(p[1], q[1], 0.0)) + fma(p[2], q[2], fma(p[3], q[3], 0.0)) + c;
            tot++;
            asc += memcmp(&x, &d, 8) == 0; desc += memcmp(&y, &d, 8) == 0; unf += memcmp((const void *)&z, &d, 8) == 0; pairw += memcmp(&w, &d, 8) == 0;
        }
    printf("rounding: %ld outputs; equal to fma chain k ascending (A-lane order) %ld, descending %ld, unfused ascending %ld, pairwise %ld\n", tot, asc, desc, unf, pairw);
    // cost
    uint64_t *d_cyc; double *d_sink;
    CK(hipMalloc(&d_cyc, 16384 * 8)); CK(hipMalloc(&d_sink, 16384 * 64 * 8));
    uint64_t cyc[4096];
    for (int blocks : {1, 256 * 4, 256 * 8, 256 * 16, 256 * 28}) {
        k_chain<<<blocks, 64>>>(d_cyc, d_sink, 1024);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); k_chain<<<blocks, 64>>>(d_cyc, d_sink, 1024); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("chain: %5d waves x 4096 dependent MFMAs: %.1f ticks per MFMA in wave 0, kernel %.3f ms\n", blocks, (double)cyc[0] / 4096.0, ms);
    }
    {
        auto run = [&](auto kern, int nv) {
            for (int blocks : {1, 256 * 28}) {
                kern<<<blocks, 64>>>(d_cyc, d_sink, 1024);
                CK(hipDeviceSynchronize());
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                CK(hipEventRecord(e0)); kern<<<blocks, 64>>>(d_cyc, d_sink, 1024); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost));
                printf("chain + %d VALU per MFMA: %5d waves x 4096 MFMAs: %.1f ticks per MFMA in wave 0, kernel %.3f ms\n", nv, blocks, (double)cyc[0] / 4096.0, ms);
            }
        };
        run(k_chain_valu<0>, 0); run(k_chain_valu<2>, 2); run(k_chain_valu<4>, 4); run(k_chain_valu<8>, 8); run(k_chain_valu<16>, 16);
    }
    for (int blocks : {1}) {
        k_chain_lds<<<blocks, 256>>>(d_cyc, d_sink, 4);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0)); k_chain_lds<<<blocks, 256>>>(d_cyc, d_sink, 4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        printf("chain + 2 LDS reads: %5d waves x 1024 MFMAs: %.1f ticks per MFMA in wave 0, kernel %.3f ms\n", blocks * 4, (double)cyc[0] / 1024.0, ms);
    }
    return 0;
}
