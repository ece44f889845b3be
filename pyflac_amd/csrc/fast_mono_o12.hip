// Instantiation of the specialised encoder kernel for one (mid-side, channels, max order) shape; see
// flac_enc_fast_impl.h.  Split across translation units so the variants compile in parallel.
#include "flac_enc_fast_impl.h"

extern "C" int fg_func_set_lds(const void *fn, size_t bytes);   // fg_ctx.cpp: per device, thread-safe
extern "C" int fg_fast_launch_mono_o12(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                                     uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                                     const uint16_t *d_crctab, size_t lds, int acc64, hipStream_t stream)
{
    const void *fn = acc64 ? (const void *)fg_encode_fast_kernel<false, 1, 12, true>
                           : (const void *)fg_encode_fast_kernel<false, 1, 12, false>;
    { const int e_ = fg_func_set_lds(fn, lds); if (e_ != 0) return e_; }
    if (acc64)
        hipLaunchKernelGGL((fg_encode_fast_kernel<false, 1, 12, true>), dim3(nblocks), dim3(64), lds, stream, d_pcm, d_descs,
                           d_windows, *P, d_slots, d_results, d_dbg, d_crctab);
    else
        hipLaunchKernelGGL((fg_encode_fast_kernel<false, 1, 12, false>), dim3(nblocks), dim3(64), lds, stream, d_pcm, d_descs,
                           d_windows, *P, d_slots, d_results, d_dbg, d_crctab);
    return (int)hipGetLastError();
}
