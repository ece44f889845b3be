// gfx950emu (test infrastructure): probe kernels for tests/test_emu_isa.py -- each runs a handful of instructions whose results the
// test derives independently (from the ISA's definitions, in Python), so that the interpreter's reading of the less common
// instruction forms the library's kernels rely on is pinned by something other than the library itself.
// One wave of 64 lanes per kernel; out[lane * 16 + k] receives result k of the lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void probe_int(const uint32_t *in, uint32_t *out)
{
    const int l = threadIdx.x;
    const uint32_t a = in[l], b = in[64 + l], c = in[128 + l];
    uint32_t r[16];
    asm volatile("v_sad_u32 %0, %1, %2, %3" : "=v"(r[0]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_sub_u32_e64 %0, %1, %2 clamp" : "=v"(r[1]) : "v"(a), "v"(b));
    asm volatile("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r[2]) : "v"(a), "v"(b));
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[3]) : "v"(a), "v"(b), "v"(c & 0x0F0F0F0Fu));
    asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(r[4]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_bfe_i32 %0, %1, %2, %3" : "=v"(r[5]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r[6]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r[7]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x78" : "=v"(r[8]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xd2" : "=v"(r[9]) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_ffbh_u32 %0, %1" : "=v"(r[10]) : "v"(a & (0xFFFFFFFFu >> (b & 31))));
    asm volatile("v_ffbl_b32 %0, %1" : "=v"(r[11]) : "v"(a << (b & 31)));
    asm volatile("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r[12]) : "v"(a), "v"(b & 31), "v"(c));
    asm volatile("v_add_u32_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r[13]) : "v"(a), "v"(b));
    asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r[14]) : "v"(a), "v"(b));
    asm volatile("v_med3_u32 %0, %1, %2, %3" : "=v"(r[15]) : "v"(a), "v"(b), "v"(c));
    for (int k = 0; k < 16; k++) out[l * 16 + k] = r[k];
}

extern "C" __global__ void probe_lanes(const uint32_t *in, uint32_t *out)
{
    const int l = threadIdx.x;
    const uint32_t a = in[l];
    uint32_t r[16];
    for (int k = 0; k < 16; k++) r[k] = 0xEEEEEEEEu;
    // (old value 0xEEEEEEEE stays where DPP does not write: an invalid source without bound_ctrl, a masked row or bank)
    asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[0]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 row_shr:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[1]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[2]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(r[3]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(r[4]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[5]) : "v"(a));
    asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0x5" : "+v"(r[6]) : "v"(a));
    asm volatile("v_add_u32_dpp %0, %1, %2 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r[7]) : "v"(a), "v"(a));
    r[8] = (uint32_t)__shfl_xor((int)a, 5);
    r[9] = (uint32_t)__shfl((int)a, (l * 7 + 3) & 63);
    r[10] = (uint32_t)__builtin_amdgcn_readlane((int)a, 17);
    r[11] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    { const unsigned long long m = __ballot(a & 1); r[12] = (uint32_t)m; r[13] = (uint32_t)(m >> 32); }
    r[14] = __builtin_amdgcn_mbcnt_hi(0xF0F0F0F0u, __builtin_amdgcn_mbcnt_lo(0x0F0F0F0Fu, 0));
    if (a & 2) r[15] = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);       // first ACTIVE lane
    for (int k = 0; k < 16; k++) out[l * 16 + k] = r[k];
}

extern "C" __global__ void probe_f64(const double *in, double *out)
{
    const int l = threadIdx.x;
    const double a = in[l], b = in[64 + l], c = in[128 + l];
    double r[8];
    r[0] = __builtin_fma(a, b, c);
    r[1] = a / b;                          // the v_div_scale / v_rcp / v_fma / v_div_fmas / v_div_fixup expansion
    r[2] = __builtin_floor(a * 1e-3);
    r[3] = (double)(int)(c * 1e-6);
    r[4] = __builtin_fabs(a) + __builtin_fmax(b, c);
    r[5] = __builtin_ldexp(a, (int)(l % 40) - 20);
    {
        // the matrix-core chain: lane k * 16 + blk * 4 + x holds A[blk][x][k] and B[blk][k][x]; D[blk][i][j] lands in lane i * 16 + blk * 4 + j
        double acc = c;
        asm volatile("s_nop 1\n\tv_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
        r[6] = acc;
    }
    r[7] = (double)(long long)a;
    for (int k = 0; k < 8; k++) out[l * 8 + k] = r[k];
}

extern "C" __global__ void probe_lds(const uint32_t *in, uint32_t *out)
{
    __shared__ uint32_t s[256];
    const int l = threadIdx.x;
    s[l] = in[l]; s[64 + l] = in[64 + l]; s[128 + l] = 0; s[192 + l] = 0;
    __syncthreads();
    uint32_t r[8];
    r[0] = s[(l * 5 + 1) & 127];
    atomicOr(&s[128 + (l & 7)], 1u << l % 32);
    atomicAdd(&s[192 + (l >> 4)], in[l] & 0xFF);
    atomicMax(&s[136], in[64 + l]);
    __syncthreads();
    r[1] = s[128 + (l & 7)]; r[2] = s[192 + (l >> 4)]; r[3] = s[136];
    r[4] = ((const uint16_t *)s)[l * 3 + 1];
    r[5] = (uint32_t)(int32_t)((const int16_t *)s)[l + 7];
    r[6] = ((const uint8_t *)s)[l * 2 + 1];
    { const uint2 v = *(const uint2 *)&s[(l & 31) * 2]; r[7] = v.x ^ (v.y << 1); }
    for (int k = 0; k < 8; k++) out[l * 8 + k] = r[k];
}

extern "C" int probe_launch(int which, const void *in, void *out)
{
    switch (which) {
    case 0: hipLaunchKernelGGL(probe_int, dim3(1), dim3(64), 0, 0, (const uint32_t *)in, (uint32_t *)out); break;
    case 1: hipLaunchKernelGGL(probe_lanes, dim3(1), dim3(64), 0, 0, (const uint32_t *)in, (uint32_t *)out); break;
    case 2: hipLaunchKernelGGL(probe_f64, dim3(1), dim3(64), 0, 0, (const double *)in, (double *)out); break;
    default: hipLaunchKernelGGL(probe_lds, dim3(1), dim3(64), 0, 0, (const uint32_t *)in, (uint32_t *)out); break;
    }
    return (int)hipDeviceSynchronize();
}
