// Instantiation of the wave-per-candidate encoder kernel for one (mid-side, channels, max order) shape; see
// flac_enc_wave_impl.h.  Split across translation units so the variants compile in parallel.
#include "flac_enc_wave_impl.h"

extern "C" int fg_wave_launch_ms_o12(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                                     uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                                     const uint16_t *d_crctab, size_t lds, int acc64, hipStream_t stream)
{
    static size_t configured[2] = {0, 0};
    constexpr int NW = 1 ? 4 : 2;
    const void *fn = acc64 ? (const void *)fg_encode_wave_kernel<true, 2, 12, true>
                           : (const void *)fg_encode_wave_kernel<true, 2, 12, false>;
    if (lds > configured[acc64 ? 1 : 0]) {
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return (int)e_;
        configured[acc64 ? 1 : 0] = lds;
    }
    if (acc64)
        hipLaunchKernelGGL((fg_encode_wave_kernel<true, 2, 12, true>), dim3(nblocks), dim3(64 * NW), lds, stream, d_pcm, d_descs,
                           d_windows, *P, d_slots, d_results, d_dbg, d_crctab);
    else
        hipLaunchKernelGGL((fg_encode_wave_kernel<true, 2, 12, false>), dim3(nblocks), dim3(64 * NW), lds, stream, d_pcm, d_descs,
                           d_windows, *P, d_slots, d_results, d_dbg, d_crctab);
    return (int)hipGetLastError();
}
