// v_mad_i32_i16 with op_sel: D = sext16(S0.half) * sext16(S1.half) + S2 -- check against plain C on random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void k(const int32_t *q, const uint32_t *p, int32_t *lo, int32_t *hi)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t a = 5, b = -7;
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,0,0,0]" : "+v"(a) : "v"(q[i]), "v"(p[i]));
    asm volatile("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[0,1,0,0]" : "+v"(b) : "v"(q[i]), "v"(p[i]));
    lo[i] = a; hi[i] = b;
}
int main()
{
    const int N = 1 << 16;
    int32_t *q, *lo, *hi; uint32_t *p;
    hipMallocManaged(&q, N * 4); hipMallocManaged(&p, N * 4); hipMallocManaged(&lo, N * 4); hipMallocManaged(&hi, N * 4);
    srand(1);
    for (int i = 0; i < N; i++) {
        q[i] = (rand() % 32768) - 16384;
        const int32_t x = (int32_t)((((int64_t)rand() << 16) ^ rand()) % (1 << 25)) - (1 << 24);     // 25-bit sample
        p[i] = ((uint32_t)(x >> 12) << 16) | ((uint32_t)x & 0xFFFu);
    }
    k<<<N / 256, 256>>>(q, p, lo, hi);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    int bad = 0;
    for (int i = 0; i < N; i++) {
        const int32_t xl = (int32_t)(int16_t)(p[i] & 0xFFFF), xh = (int32_t)(int16_t)(p[i] >> 16);
        if (lo[i] != 5 + q[i] * xl || hi[i] != -7 + q[i] * xh) bad++;
    }
    printf("v_mad_i32_i16 op_sel: %d mismatches of %d\n", bad, N);
    return bad != 0;
}
