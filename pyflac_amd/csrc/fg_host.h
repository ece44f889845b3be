// Host-side internals of libflacgpu (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "fg_types.h"
#include "flacgpu.h"

// ---- kernel launchers (flac_enc_kernels.hip / flac_dec_kernels.hip)
extern "C" {
size_t fg_enc_lds_bytes(const FgEncParams *P);
#define FG_CRC_TABLE_WORDS (2048 + 2 * 5632)
const uint16_t *fg_crc_tables_host();       // fg_ctx.cpp: the kernels' CRC-16 tables, built once per process
int fg_launch_widen16(const int16_t *d_src, int32_t *d_dst, unsigned long long n, hipStream_t stream);
int fg_launch_encode(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                     uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                     const uint16_t *d_crctab, hipStream_t stream);
size_t fg_fast_lds_bytes(const FgEncParams *P, int nch, int ms, int maxo);
int fg_launch_encode_fast(const void *d_pcm, const FgBlockDesc *d_descs, const float *d_windows, const FgEncParams *P,
                          uint32_t nblocks, uint8_t *d_slots, FgBlockResult *d_results, FgDebugRec *d_dbg,
                          const uint16_t *d_crctab, hipStream_t stream);
// de-fused pipeline (flac_enc_pipe.hip)
int fg_pipe_supported(const FgEncParams *P);
uint32_t fg_pipe_block_ws(uint32_t n);
int fg_pipe_block_ok(uint32_t n, uint32_t max_po);
size_t fg_pipe_scratch_bytes(const FgEncParams *P, uint32_t nblocks);
void fg_pipe_carve(const FgEncParams *P, uint32_t nblocks, void *base, FgPipeBufs *B);
int fg_launch_encode_pipe(const FgPipeLaunch *L);
int fg_launch_merge(const FgBlockResult *d_res, const uint32_t *d_chunk_bits, const unsigned long long *d_moffs, const uint8_t *d_mtmp,
                    const uint32_t *d_fbase, const uint32_t *d_fstr, uint32_t channels, uint32_t nframes, uint32_t *d_sizes, uint32_t *d_errs,
                    unsigned long long *d_offsets, uint8_t *d_dst, unsigned long long dst_cap, FgBlockResult *d_fres,
                    unsigned long long *d_user_offsets, hipStream_t stream);
int fg_mfma_selfcheck(hipStream_t stream);
int fg_launch_pipe_assemble(const FgBlockDesc *d_descs, uint32_t nblocks, const uint8_t *d_slots, uint32_t slot_bytes,
                            uint32_t chunk_cap_words, uint32_t nw, const uint32_t *d_chunk_bits, FgBlockResult *d_results,
                            unsigned long long *d_offsets, uint8_t *d_dst, uint64_t dst_cap, const uint16_t *d_crctab,
                            unsigned long long *d_user_offsets, const unsigned long long *d_guard, hipStream_t stream, uint32_t first,
                            const FgPackDirect *direct);
int fg_launch_pipe_publish(const FgBlockDesc *d_descs, uint32_t first, uint32_t count, const FgBlockResult *d_results, const uint32_t *d_chunk_bits,
                           const FgPackDirect *direct, hipStream_t stream);
int fg_launch_dec_index_init(unsigned long long *d_offsets, unsigned long long *d_alt, unsigned long long *d_info, uint32_t nframes,
                             unsigned long long *d_stamp, hipStream_t stream);
// end-of-call hand-over through pinned memory (flac_enc_kernels.hip)
int fg_launch_stamp(unsigned long long *d_stamp, hipStream_t stream);
int fg_launch_signal(const unsigned long long *src0, uint32_t n0, const unsigned long long *src1, uint32_t n1,
                     const unsigned long long *d_stamp, unsigned long long *h_sig, unsigned long long seq, hipStream_t stream,
                     unsigned long long *d_reset = nullptr);
int fg_launch_md5_streams(const void *d_pcm, uint32_t pcm_i16, uint32_t channels, uint32_t bps, const void *d_jobs, uint32_t njobs,
                          uint32_t *d_out, hipStream_t stream);
int fg_launch_signal_direct(const unsigned long long *d_total, const unsigned long long *d_guard, const unsigned long long *d_stamp,
                            unsigned long long *h_sig, unsigned long long seq, hipStream_t stream, unsigned long long *d_reset);
size_t fg_scan_words(uint32_t nblocks);
int fg_launch_scan(FgBlockResult *d_results, const uint32_t *d_chunk_bits, uint32_t nblocks, unsigned long long *d_offsets, int all_pipe,
                   const unsigned long long *d_errs, hipStream_t stream);
int fg_launch_copy(const uint8_t *d_slots, uint32_t slot_bytes, const FgBlockResult *d_results, uint32_t nblocks,
                   const unsigned long long *d_offsets, uint8_t *d_dst, hipStream_t stream, uint64_t dst_cap);
size_t fg_dec_scan_words(uint32_t nframes);
int fg_launch_dec_headers(const uint8_t *d_stream, unsigned long long stream_len, const unsigned long long *d_offsets, uint32_t nframes,
                          uint32_t si_channels, uint32_t si_bps, FgDecFrame *d_frames, FgDecResult *d_results,
                          unsigned long long *d_totals, unsigned long long cap_samples, hipStream_t stream, int write_err);
int fg_launch_dec_index(const uint8_t *d_stream, unsigned long long len, uint32_t channels, uint32_t bps, unsigned long long first_number,
                        uint32_t nframes, unsigned long long *d_offsets, unsigned long long *d_info, unsigned long long *d_alt,
                        const FgDecRange *d_ranges, uint32_t nranges, hipStream_t stream, uint32_t *d_hdrrec, unsigned long long *d_stamp,
                        unsigned long long *d_gate = nullptr, unsigned long long epoch = 0);
int fg_launch_decode_slow(const uint8_t *d_stream, const FgDecFrame *d_frames, const uint32_t *d_frame_list, uint32_t nlist,
                          int32_t *d_pcm, FgDecResult *d_results, const uint16_t *d_crctab, int32_t *d_scratch, uint32_t interleave,
                          hipStream_t stream);
int fg_launch_decode_wparse(const uint8_t *d_stream, uint64_t stream_len, const FgDecFrame *d_frames, uint32_t nframes,
                            int32_t *d_scratch, FgDecSub *d_subs, FgDecResult *d_results, int wide, uint16_t *d_rparams,
                            unsigned long long *d_counters, hipStream_t stream, int plane16, const FgDecSelf *self);
int fg_launch_decode_wrestore(const FgDecFrame *d_frames, uint32_t nframes, uint32_t channels, const int32_t *d_scratch,
                              const FgDecSub *d_subs, int32_t *d_pcm, FgDecResult *d_results, uint32_t interleave, int wide,
                              hipStream_t stream, int plane16, FgDecResult *h_rows, const unsigned long long *d_planeoff,
                              unsigned long long *d_join = nullptr, unsigned long long epoch = 0);
int fg_launch_dec_raise(unsigned long long *d_word, unsigned long long epoch, hipStream_t stream);
int fg_launch_dec_spin(unsigned long long ticks, hipStream_t stream);
int fg_launch_decode_warmup(const FgDecFrame *d_frames, uint32_t nframes, uint32_t channels, const FgDecSub *d_subs,
                            const int32_t *d_scratch, int32_t *d_warm, hipStream_t stream);
int fg_launch_narrow16(const int32_t *d_in, int16_t *d_out, uint64_t n, hipStream_t stream);
int fg_launch_compare(const int32_t *d_a, const int32_t *d_b, uint64_t n, unsigned long long *d_first, hipStream_t stream);
int fg_launch_decode_crc(const uint8_t *d_stream, const FgDecFrame *d_frames, uint32_t nframes, FgDecResult *d_results,
                         const uint16_t *d_crctab, hipStream_t stream, const unsigned long long *d_offsets, unsigned long long stream_len);
}

void fg_set_error(const std::string &msg);

// Grow-only device buffer.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes);
    void release();
};

struct WindowEntry {
    uint32_t n, parts;
    uint32_t off;   // floats
};

struct flacgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // the decoder's index tables as the last call left them: emptied for `idx_clean_n` frames behind its end-of-call signal (0: not)
    uint32_t idx_clean_n = 0;
    void *idx_clean_off = nullptr, *idx_clean_info = nullptr;
    unsigned long long gate_epoch = 0;      // the decode launch's fork / join words (flacgpu_dec_api.cpp decode_frames_impl): one epoch a call
    bool gate_off = false;                  // a wait on them timed out once: events from then on
    void *gate_words_of = nullptr;          // the dec_info buffer whose header (counters, join word) has been zeroed
    uint32_t dec_p16_hold = 0;       // decode calls that still take 32-bit residual planes (a stream showed values beyond 16 bits)
    hipStream_t stream2 = nullptr;   // tail blocks (generic kernel) run beside the specialised kernel
    hipStream_t stream3 = nullptr;   // short blocks of the pipeline's packing stage
    hipEvent_t evp[2] = {nullptr, nullptr};
    hipStream_t gstream[3] = {nullptr, nullptr, nullptr};      // the encoder pipeline's groups 1..3 (FgPipeLaunch.ngroups)
    hipEvent_t gev_fork = nullptr, gev_join[3] = {nullptr, nullptr, nullptr};
    hipEvent_t gev_eval[3] = {nullptr, nullptr, nullptr};      // behind the evaluation of groups 0..2 (direct packing: FgPipeLaunch.gev_eval)
    // direct packing path (round 5, FgPackDirect): look-back words of the frames of a call, the epoch that tells one call's words from
    // another's, and the switch (flacgpu_set_direct: 0 = chunks through HBM, scan and assembly kernel as in rounds 2-4)
    hipStream_t md5_stream = nullptr;      // flacgpu_md5_streams: a stream, two events and a lock of its own (it runs beside encode calls)
    hipEvent_t md5_ev[2] = {nullptr, nullptr};
    std::mutex md5_mu;
    DevBuf lb;
    uint32_t lb_epoch = 0;
    int direct = 2;            // (2: with the evaluation inside the packing kernel -- flacgpu_set_direct)
    uint32_t desc_side_first = 0, desc_slow_first = 0;         // (cached with the block list: a block that keeps the chunk form / takes the generic kernel lies in front of a direct block)
    hipEvent_t evx[3] = {nullptr, nullptr, nullptr};       // [2]: frame table ready (header pass + scan on the side stream)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evs[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stage timing
    unsigned long long *guard_clean_ptr = nullptr;      // the pipeline guard counters the last call's signal kernel reset (FgPipeLaunch.guard_clean)
    int stage_timing = 0;            // 0: no events (end of call through the pinned signal area, GPU time from wall-clock stamps),
                                     // 1: HIP events around the call and its kernel groups, 2: also between the encoder's stages
    unsigned long long *h_sig = nullptr;   // pinned: [0] sequence number, [2..10) payload, [10] start stamp, [11] end stamp
    unsigned long long sig_seq = 0;
    double wall_khz = 100000.0;      // rate of the device's constant wall clock
    DevBuf stamp;
    bool wait_signal(unsigned long long seq);   // poll h_sig[0] (bounded), then hipStreamSynchronize; false on a device error
    double log_guard_thr = 1e-6;
    std::recursive_mutex mu;    // (the batch entry points nest: streams of more than two channels run the one-channel encode inside)
    DevBuf descs, slots, results, dbg, crctab, windows, offsets, scratch_pcm, scratch_out, dec_frames, dec_results,
        dec_scratch, dec_subs, dec_poff, dec_hrec, dec_prof, dec_redo, dec_info, dec_off, dec_rparams, dec_warm, dec_ranges, pipe,
        mc_tmp, mc_offs, mc_map, mc_sizes, mc_res, mc_foffs,   // streams of more than two channels (fg_ctx.cpp encode_multichannel)
        md5_jobs;
    const uint32_t *last_chunk_bits = nullptr;   // the pipeline's chunk bit counts of the last encode call (device), or null
    std::vector<unsigned char> desc_key;   // settings + stream list the block list in `dev_descs` was built for
    uint32_t desc_nfast = 0, desc_nws2 = 0, desc_nrag = 0;
    std::vector<FgBlockDesc> dev_descs;   // copy of the block list currently in `descs`
    const void *dev_descs_ptr = nullptr;
    std::vector<float> h_windows;
    std::vector<WindowEntry> win_index;
    bool windows_dirty = false;
    std::string window_note;      // set when the window self-check replaced a table (see window_offset)
    int mfma_bad = 0;             // results of the matrix-core self-check that differed from the v_fma_f64 chain (fg_mfma_selfcheck)
    std::string selfcheck_note;
    bool debug = false;
    uint32_t last_nblocks = 0;
    // pinned staging for the stream (callback) API
    void *h_pin = nullptr;
    size_t h_pin_cap = 0;
    bool ensure_pinned(size_t bytes);
    // pinned landing area for the small device-to-host reads of the batch calls (totals, per-frame status)
    void *h_res = nullptr;
    size_t h_res_cap = 0;
    bool ensure_pinned_res(size_t bytes);
    uint32_t window_offset(uint32_t n, uint32_t parts);   // returns float offset, computing the table if new
    bool sync_windows();
};

// The calling thread queued the upload of its next encode call's samples on the context's main stream and did not wait: that call
// orders every other stream it uses behind the main one (no FgPipeLaunch.no_fork).  Per thread: encoders on several threads share
// the default context.
void fg_set_input_on_stream(bool on);
flacgpu_ctx *fg_default_ctx();   // lazily created context on the current/default device

// End-of-call wait.  The batch calls last one to a few milliseconds; a blocking hipStreamSynchronize adds its wake-up
// latency to each of them, so the stream is polled for a bounded time first (FLACGPU_SPIN_US, default 3000; 0 = never).
hipError_t fg_stream_wait(hipStream_t stream);

int fg_resolve_settings(flacgpu_settings *s);
// settings helpers shared by the libFLAC-style encoder and the batch API
void fg_fill_params(const flacgpu_settings &s, uint32_t max_n, bool pcm_i16, bool debug, FgEncParams *P);
uint32_t fg_slot_bytes(const flacgpu_settings &s, uint32_t max_n);
void fg_tukey_window(float *w, int32_t L, float p);

// MD5 (RFC 1321) used for STREAMINFO
struct FgMd5 {
    uint32_t a, b, c, d;
    uint64_t len;
    uint8_t buf[64];
    uint32_t fill;
    void init();
    void update(const uint8_t *p, size_t n);
    void final(uint8_t out[16]);
    void update_pcm(const int32_t *interleaved, uint64_t nvalues, uint32_t bps);
};

uint8_t fg_crc8(const uint8_t *p, size_t n);
uint16_t fg_crc16(const uint8_t *p, size_t n);
