// flac_enc_fast.hip -- dispatcher over the specialised encoder kernels (flac_enc_fast_impl.h, fast_*.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fg_types.h"

#define FGS_FBW 1024
#define FGS_CSTR 136

extern "C" {

#define FG_DECL(name)                                                                                                      \
    int fg_fast_launch_##name(const void *, const FgBlockDesc *, const float *, const FgEncParams *, uint32_t, uint8_t *,  \
                              FgBlockResult *, FgDebugRec *, const uint16_t *, size_t, int, hipStream_t);
FG_DECL(ms_o8) FG_DECL(ms_o12) FG_DECL(st_o8) FG_DECL(st_o12) FG_DECL(mono_o8) FG_DECL(mono_o12)
size_t fg_fast_lds_bytes(const FgEncParams *P, int nch, int ms, int maxo)
{
    const size_t sb = P->bps > 16 ? 4 : 2;   // staged sample size
    const int NC = ms ? 4 : nch;
    const int MQ = maxo > 0 ? maxo : 1;
    size_t off = 0;
    auto add = [&](size_t b) { off += (b + 15) & ~(size_t)15; };
    add((size_t)(P->sig_stride + 128) * sb);
    add(nch == 2 ? (size_t)(P->sig_stride + 128) * sb : 16);
    size_t ubytes = (size_t)NC * FGS_CSTR * 8;
    if (ubytes < (FGS_FBW + 2) * 4) ubytes = (FGS_FBW + 2) * 4;
    add(ubytes);
    add((size_t)NC * P->nvec * (maxo + 1) * 8);
    add((size_t)NC * P->nvec * MQ * 4);
    add((size_t)NC * P->nvec * 4);
    add((size_t)NC * MQ * 4);
    add(128 * 4);
    return off;
}

// Returns 0 on success, -1 when no specialisation covers the configuration.
int fg_launch_encode_fast(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                          uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                          const uint16_t *d_crctab, hipStream_t stream)
{
    if (nblocks == 0) return 0;
    const int nch = (int)P->channels, ms = P->do_mid_side ? 1 : 0;
    if (nch < 1 || nch > 2 || P->max_lpc_order > 12 || P->bps > 24) return -1;
    const int maxo = P->max_lpc_order <= 8 ? 8 : 12;
    const int acc64 = P->bps > 16 ? 1 : 0;
    size_t lds = fg_fast_lds_bytes(P, nch, ms, maxo);
    if (fg_tune("FLACGPU_LDS_PAD")) lds += (size_t)atoi(fg_tune("FLACGPU_LDS_PAD"));   // occupancy experiments
    if (lds > 160 * 1024) return -1;
#define FG_CALL(name) return fg_fast_launch_##name(d_pcm, d_descs, d_windows, P, nblocks, d_slots, d_results, d_dbg, d_crctab, lds, acc64, stream)
    if (nch == 2 && ms) { if (maxo == 8) FG_CALL(ms_o8); else FG_CALL(ms_o12); }
    else if (nch == 2) { if (maxo == 8) FG_CALL(st_o8); else FG_CALL(st_o12); }
    else { if (maxo == 8) FG_CALL(mono_o8); else FG_CALL(mono_o12); }
}

}  // extern "C"
