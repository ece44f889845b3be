// flac_enc_pipe.hip -- dispatcher over the de-fused encoder pipeline (flac_enc_pipe_impl.h, pipe_*.hip) and the launchers of
// its shape-independent kernel (chunk assembly + CRC-16; the frame sizes come out of the scan, flac_enc_kernels.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "flac_enc_pipe_impl.h"

extern "C" {

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

int fg_launch_pipe_assemble(const FgBlockDesc *d_descs, uint32_t nblocks, const uint8_t *d_slots, uint32_t slot_bytes,
                            uint32_t chunk_cap_words, uint32_t nw, const uint32_t *d_chunk_bits, FgBlockResult *d_results,
                            unsigned long long *d_offsets, uint8_t *d_dst, uint64_t dst_cap, const uint16_t *d_crctab,
                            unsigned long long *d_user_offsets, const unsigned long long *d_guard, hipStream_t stream)
{
    if (nblocks == 0) return 0;
    constexpr int WPB = 4;
    hipLaunchKernelGGL((fg_pipe_assemble_kernel<WPB>), dim3((nblocks + WPB - 1) / WPB), dim3(WPB * 64), 0, stream, d_descs, nblocks, d_slots,
                       slot_bytes, chunk_cap_words, nw, d_chunk_bits, d_results, (u64 *)d_offsets, d_dst, (u64)dst_cap, d_crctab,
                       (u64 *)d_user_offsets, d_guard);
    return (int)hipGetLastError();
}

}  // extern "C"
