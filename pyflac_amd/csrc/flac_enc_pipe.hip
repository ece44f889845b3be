// flac_enc_pipe.hip -- dispatcher over the de-fused encoder pipeline (flac_enc_pipe_impl.h, pipe_*.hip) and the launchers of
// its shape-independent kernel (chunk assembly + CRC-16; the frame sizes come out of the scan, flac_enc_kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string.h>
#include "flac_enc_pipe_impl.h"

namespace {
// Start-up self-check of the matrix-core chain.  The autocorrelation's bit-exactness rests on a property
// of v_mfma_f64_4x4x4_4b_f64 no manual states: its sum over k is a chain of fused multiply-adds in ascending k, each rounded like
// v_fma_f64 (tools/ubench/mfma64.hip found it).  This kernel checks that on the device at hand: random operands of the kind the
// kernel uses -- floats widened to double, exponents spread over 40 binades, signs mixed --, 64 dependent instructions a wave (the
// accumulator carries on, as in the chains), every result compared bit for bit with the v_fma_f64 chain over the same operands in
// the layout fg_pipe_autoc_kernel relies on (A[i][k] in lane 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane
// 16 i + 4 b + j).  A few microseconds at flacgpu_ctx_create.
__global__ void __launch_bounds__(64) fg_mfma_selfcheck_kernel(unsigned int *bad, unsigned int rounds)
{
    __shared__ double sa[64], sbv[64];
    const int lane = threadIdx.x;
    unsigned int x = 0x9E3779B9u * (blockIdx.x * 64u + (unsigned)lane + 1u);
    auto rnd = [&]() -> double {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const float m = 1.0f + (float)(x & 0x7FFFFF) * (1.0f / 8388608.0f);
        const int e = (int)((x >> 23) % 41u) - 20;
        const float v = ldexpf((x >> 31) ? -m : m, e);
        return (double)v;
    };
    double acc_m = 0.0, acc_v = 0.0;
    unsigned int nbad = 0;
    const int i_ = lane >> 4, b_ = (lane >> 2) & 3, j_ = lane & 3;
    for (unsigned int r = 0; r < rounds; r++) {
        const double a = rnd(), b = rnd();
        sa[lane] = a; sbv[lane] = b;
        __syncthreads();
        acc_m = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc_m, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; k++) acc_v = __builtin_fma(sa[16 * k + 4 * b_ + i_], sbv[16 * k + 4 * b_ + j_], acc_v);
        if (__double_as_longlong(acc_m) != __double_as_longlong(acc_v)) nbad++;
        acc_v = acc_m;                      // (after a difference the chains would part for good: count instructions, not their wake)
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}
}  // namespace

extern "C" {

// returns the number of matrix-core results that differ from the v_fma_f64 chain (0 = the encoder's autocorrelation is bit-exact
// on this device), negative on a launch failure
int fg_mfma_selfcheck(hipStream_t stream)
{
    unsigned int *d_bad = nullptr, h_bad = 0;
    if (hipMalloc(&d_bad, 4) != hipSuccess) return -1;
    int rc = -1;
    if (hipMemsetAsync(d_bad, 0, 4, stream) == hipSuccess) {
        hipLaunchKernelGGL(fg_mfma_selfcheck_kernel, dim3(16), dim3(64), 0, stream, d_bad, 64u);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipStreamSynchronize(stream) == hipSuccess)
            rc = (int)(h_bad > 0x7FFFFFFFu ? 0x7FFFFFFFu : h_bad);
    }
    (void)hipFree(d_bad);
    return rc;
}

#define FG_DECLP(name) int fg_pipe_launch_##name(const FgPipeLaunch *L);
FG_DECLP(ms_o8) FG_DECLP(ms_o12) FG_DECLP(st_o8) FG_DECLP(st_o12) FG_DECLP(mono_o8) FG_DECLP(mono_o12)

// Does the pipeline cover this configuration at all?  (Per block it further needs n % (64 * ws) == 0 and at least 16
// samples per packing lane; partitions no finer than a lane.)
int fg_pipe_supported(const FgEncParams *P)
{
    if (P->channels < 1 || P->channels > 2 || P->max_lpc_order > 12 || (P->bps > 24 && P->bps != 32) || P->sig_stride == 0) return 0;
    return 1;
}

// Packing waves per subframe for a block of n samples: two when a lane then still walks >= 32 samples (every lane spans
// at least one output word, which the register-assembled emission needs), one for everything else (shorter blocks, and the
// ragged geometry of blocks whose length is no multiple of 64 -- flac_enc_pipe_impl.h PipeGeo).
uint32_t fg_pipe_block_ws(uint32_t n)
{
    if (n % 128 == 0 && n / 128 >= 32) return 2;
    return 1;
}

// Can the pipeline take a block of n samples?  Every working lane needs 16 samples (the predictor history of a lane then lies in
// one other lane, the warm-up samples in lane 0): n / 2^pm >= 16 with pm = the finest partition order the block allows, and no
// more than 64 finest partitions.
int fg_pipe_block_ok(uint32_t n, uint32_t max_po)
{
    uint32_t pm = 0, b = n;
    if (n < 32) return 0;
    while (!(b & 1)) { pm++; b >>= 1; }
    if (pm > max_po) pm = max_po;
    if (pm > 6) return 0;
    return (n >> pm) >= 16 ? 1 : 0;
}

// bytes of scratch the pipeline needs for `nblocks` blocks, and the carve of it
size_t fg_pipe_scratch_bytes(const FgEncParams *P, uint32_t nblocks)
{
    const size_t NC = 4, MAXO = 12;
    size_t b = 0;
    auto add = [&](size_t x) { b += (x + 255) & ~(size_t)255; };
    add((size_t)nblocks * NC * P->nvec * (MAXO + 1) * 8);
    add((size_t)nblocks * NC * 4);
    add((size_t)nblocks * 4);
    add((size_t)nblocks * NC * P->nvec * MAXO * 4);
    add((size_t)nblocks * NC * P->nvec * 4);
    add((size_t)nblocks * NC * sizeof(FgPipeDec));
    add((size_t)nblocks * 4 * 4);
    add(64);
    return b;
}

void fg_pipe_carve(const FgEncParams *P, uint32_t nblocks, void *base, FgPipeBufs *B)
{
    const size_t NC = 4, MAXO = 12;
    unsigned char *p = (unsigned char *)base;
    auto take = [&](size_t x) { unsigned char *r = p; p += (x + 255) & ~(size_t)255; return r; };
    B->autoc = (double *)take((size_t)nblocks * NC * P->nvec * (MAXO + 1) * 8);
    B->wasted = (uint32_t *)take((size_t)nblocks * NC * 4);
    B->nv = (uint32_t *)take((size_t)nblocks * 4);
    B->qres = (int32_t *)take((size_t)nblocks * NC * P->nvec * MAXO * 4);
    B->lres = (uint32_t *)take((size_t)nblocks * NC * P->nvec * 4);
    B->dec = (FgPipeDec *)take((size_t)nblocks * NC * sizeof(FgPipeDec));
    B->chunk_bits = (uint32_t *)take((size_t)nblocks * 4 * 4);
    B->guard = (unsigned long long *)take(64);
}

// Returns 0 on success, -1 when no specialisation covers the configuration.
int fg_launch_encode_pipe(const FgPipeLaunch *L)
{
    if (L->nblocks == 0) return 0;
    const FgEncParams *P = &L->P;
    if (!fg_pipe_supported(P)) return -1;
    const int nch = (int)P->channels, ms = P->do_mid_side ? 1 : 0;
    const int maxo = P->max_lpc_order <= 8 ? 8 : 12;
#define FG_CALLP(name) return fg_pipe_launch_##name(L)
    if (nch == 2 && ms) { if (maxo == 8) FG_CALLP(ms_o8); else FG_CALLP(ms_o12); }
    else if (nch == 2) { if (maxo == 8) FG_CALLP(st_o8); else FG_CALLP(st_o12); }
    else { if (maxo == 8) FG_CALLP(mono_o8); else FG_CALLP(mono_o12); }
}

// `direct` set (behind a direct launch, FgPackDirect): only the blocks [first, nblocks) of the list, placed through the look-back words;
// the kernel runs even without such blocks -- its first wave hands the error flags to the host's words.
int fg_launch_pipe_assemble(const FgBlockDesc *d_descs, uint32_t nblocks, const uint8_t *d_slots, uint32_t slot_bytes,
                            uint32_t chunk_cap_words, uint32_t nw, const uint32_t *d_chunk_bits, FgBlockResult *d_results,
                            unsigned long long *d_offsets, uint8_t *d_dst, uint64_t dst_cap, const uint16_t *d_crctab,
                            unsigned long long *d_user_offsets, const unsigned long long *d_guard, hipStream_t stream, uint32_t first,
                            const FgPackDirect *direct)
{
    if (nblocks == 0) return 0;
    constexpr int WPB = 4;
    FgPackDirect D;
    memset(&D, 0, sizeof D);
    if (direct) D = *direct; else first = 0;
    const uint32_t cnt = nblocks - first;
    hipLaunchKernelGGL((fg_pipe_assemble_kernel<WPB>), dim3(cnt ? (cnt + WPB - 1) / WPB : 1), dim3(WPB * 64), 0, stream, d_descs, nblocks, d_slots,
                       slot_bytes, chunk_cap_words, nw, d_chunk_bits, d_results, (u64 *)d_offsets, d_dst, (u64)dst_cap, d_crctab,
                       (u64 *)d_user_offsets, d_guard, first, D);
    return (int)hipGetLastError();
}

// sizes of the blocks descs[first, first + count) into the look-back words of a direct launch (frames of the generic kernel)
int fg_launch_pipe_publish(const FgBlockDesc *d_descs, uint32_t first, uint32_t count, const FgBlockResult *d_results, const uint32_t *d_chunk_bits,
                           const FgPackDirect *direct, hipStream_t stream)
{
    if (count == 0) return 0;
    hipLaunchKernelGGL(fg_pipe_publish_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_descs, first, count, d_results, d_chunk_bits, *direct);
    return (int)hipGetLastError();
}

}  // extern "C"
