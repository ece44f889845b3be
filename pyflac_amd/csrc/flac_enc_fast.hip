// flac_enc_fast.hip -- dispatcher over the specialised encoder kernels (flac_enc_fast_impl.h, fast_*.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "fg_types.h"

#define FGS_FBW 1024
#define FGS_DSTR 168
#define FGS_CSTR 136

extern "C" {

#define FG_DECL(name)                                                                                                      \
    int fg_fast_launch_##name(const void *, const FgBlockDesc *, const float *, const FgEncParams *, uint32_t, uint8_t *,  \
                              FgBlockResult *, FgDebugRec *, const uint16_t *, size_t, int, hipStream_t);
FG_DECL(ms_o8) FG_DECL(ms_o12) FG_DECL(st_o8) FG_DECL(st_o12) FG_DECL(mono_o8) FG_DECL(mono_o12)
#define FG_DECLW(name)                                                                                                     \
    int fg_wave_launch_##name(const void *, const FgBlockDesc *, const float *, const FgEncParams *, uint32_t, uint8_t *,  \
                              FgBlockResult *, FgDebugRec *, const uint16_t *, size_t, int, hipStream_t);
FG_DECLW(ms_o8) FG_DECLW(ms_o12) FG_DECLW(st_o8) FG_DECLW(st_o12)

// LDS of the wave-per-candidate kernel (flac_enc_wave_impl.h); mirrors its carve
size_t fg_wave_lds_bytes(const FgEncParams *P, int nch, int ms, int maxo)
{
    const size_t sb = P->bps > 16 ? 4 : 2;
    const size_t NC = ms ? 4 : (size_t)nch;
    const size_t MQ = maxo > 0 ? (size_t)maxo : 1;
    size_t off = 0;
    auto add = [&](size_t b) { off += (b + 15) & ~(size_t)15; };
    add((size_t)(P->sig_stride + 256) * sb);
    add(nch == 2 ? (size_t)(P->sig_stride + 256) * sb : 16);
    const size_t lev = ((size_t)P->nvec * (P->max_lpc_order ? P->max_lpc_order : 1) * 12 + 64 + 15) & ~(size_t)15;
    const size_t wbytes = lev > (size_t)FGS_DSTR * 8 ? lev : (size_t)FGS_DSTR * 8;
    size_t ubytes = NC * wbytes;
    if (ubytes < (size_t)nch * (FGS_FBW + 2) * 4) ubytes = (size_t)nch * (FGS_FBW + 2) * 4;
    add(ubytes);
    add(NC * P->nvec * (maxo + 1) * 8);
    add(NC * P->nvec * MQ * 4);
    add(NC * P->nvec * 4);
    add(NC * MQ * 4);
    add(NC * 64 * 4);
    add(NC * 32);
    add(1536 * 2);
    add(64 * 4);
    add(64 * NC * 4);
    add(72 * 4);
    add(32);
    add(16);
    return off;
}

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
    // FLACGPU_WAVE=1: the variant with one wavefront per predictor candidate (flac_enc_wave_impl.h).  It has the lower
    // latency for a handful of blocks but repeats the serial stages (autocorrelation chain, Levinson-Durbin, Rice search)
    // in every wave, so the one-block-per-wavefront kernel wins on throughput (1.23 ms against 1.42 ms for 7032 blocks)
    if (nch == 2 && !P->limit_min_bitrate && getenv("FLACGPU_WAVE") && atoi(getenv("FLACGPU_WAVE")) == 1) {
        const size_t wl = fg_wave_lds_bytes(P, nch, ms, maxo);
        if (wl <= 160 * 1024) {
#define FG_CALLW(name) return fg_wave_launch_##name(d_pcm, d_descs, d_windows, P, nblocks, d_slots, d_results, d_dbg, d_crctab, wl, acc64, stream)
            if (ms) { if (maxo == 8) FG_CALLW(ms_o8); else FG_CALLW(ms_o12); }
            else { if (maxo == 8) FG_CALLW(st_o8); else FG_CALLW(st_o12); }
        }
    }
    size_t lds = fg_fast_lds_bytes(P, nch, ms, maxo);
    if (getenv("FLACGPU_LDS_PAD")) lds += (size_t)atoi(getenv("FLACGPU_LDS_PAD"));   // occupancy experiments
    if (lds > 160 * 1024) return -1;
#define FG_CALL(name) return fg_fast_launch_##name(d_pcm, d_descs, d_windows, P, nblocks, d_slots, d_results, d_dbg, d_crctab, lds, acc64, stream)
    if (nch == 2 && ms) { if (maxo == 8) FG_CALL(ms_o8); else FG_CALL(ms_o12); }
    else if (nch == 2) { if (maxo == 8) FG_CALL(st_o8); else FG_CALL(st_o12); }
    else { if (maxo == 8) FG_CALL(mono_o8); else FG_CALL(mono_o12); }
}

}  // extern "C"
