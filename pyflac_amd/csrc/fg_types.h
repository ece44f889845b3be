// Structures shared by the HIP kernels and the host side of libflacgpu.
#pragma once
#include <stdint.h>

// ---- environment switches --------------------------------------------------------------------------------------------------------
// The release library (`make`: libflacgpu.so) reads ONE environment variable: FLACGPU_DEVICE, which GPU the default context uses.
//   * fg_sel(): the kernel SELECTORS that choose among implementations with identical results and the two TEST HOOKS that alter
//     results on purpose (FLACGPU_NO_FAST, FLACGPU_WS, FLACGPU_MC, FLACGPU_GROUPS, FLACGPU_KEEP, FLACGPU_QUICK_START, FLACGPU_FBW, FLACGPU_DIRECT24, FLACGPU_AUTOC1, FLACGPU_DEC_SELF,
//     FLACGPU_DEC_GATE, FLACGPU_DEC_P16, FLACGPU_DEC_WAVE; FLACGPU_WINDOW_SELFTEST, FLACGPU_VERIFY_SELFTEST) are read by the test-hooks library only
//     (libflacgpu_testhooks.so: the same kernel objects, the three host files compiled with -DFG_TESTHOOKS; the cross-check tests
//     load it explicitly, pyflac_amd/_lib.py testhooks_lib()).
//   * fg_tune(): everything that skips work, reorders it for an experiment or prints diagnostics (FLACGPU_DEC_SKIP, FLACGPU_STOP,
//     FLACGPU_DEC_CRC_LATE, FLACGPU_LDS_PAD, FLACGPU_FCAP, FLACGPU_ALIAS, FLACGPU_DIRECT_X, FLACGPU_SPIN_US,
//     FLACGPU_DEC_PROF, FLACGPU_API_PROF, FG_REFWALK_DEBUG, FLACGPU_DEC_WPS) is read by `make TUNING=1` builds only.
// flacgpu_build_flags() says what a library is: bit 0 tuning, bit 2 test hooks.
#include <stdlib.h>
static inline const char *fg_tune(const char *name)
{
#ifdef FG_TUNING
    return getenv(name);
#else
    (void)name;
    return (const char *)0;
#endif
}
static inline const char *fg_sel(const char *name)
{
#if defined(FG_TESTHOOKS) || defined(FG_TUNING)
    return getenv(name);
#else
    (void)name;
    return (const char *)0;
#endif
}


#define FG_MAX_ORDER 32
#define FG_MAX_VEC 16    // autocorrelation vectors per candidate; subdivide_tukey(3) needs 9
#define FG_MAX_CAND 4    // L, R, M, S
#define FG_MAX_PARTS 256 // 2^8: FLAC__SUBSET_MAX_RICE_PARTITION_ORDER (format.h:151)
#define FG_WINW 128      // bit-packer LDS window, 32-bit words
#define FG_DH 32         // autocorrelation history entries kept in front of each chunk
#define FG_DK 96          // autocorrelation chunk length (doubles per candidate)

// Error bits reported per block.
#define FG_ERR_RANGE 1u        // input sample outside the bits-per-sample range
#define FG_ERR_SLOT 2u         // frame did not fit its output slot
#define FG_ERR_SIDE33 4u       // (unused since the generic kernel handles the 33-bit side channel; value kept reserved)
#define FG_ERR_INTERNAL 8u
#define FG_ERR_REDO 16u         // not an error: the specialised kernel hands this block to the generic kernel
#define FG_ERR_CHAIN 32u        // not an error: a frame of the direct packing path could not be placed (a block handed back, a frame beyond
                                // the LDS frame buffer, a look-back that ran out of time): the call repeats packing, scan and assembly the
                                // round-2 way (chunks through HBM)

struct FgEncParams {
    uint32_t channels, bps, sample_rate, blocksize;
    uint32_t do_mid_side;
    uint32_t max_lpc_order, qlp_precision, min_po, max_po;
    uint32_t apod_parts;   // 0/1: one tukey window; >= 2: subdivide_tukey(parts)
    uint32_t rice_limit;   // 15 (bps <= 16) or 31
    uint32_t slot_bytes;   // output slot stride, multiple of 4
    uint32_t sig_stride;   // int32 elements per staged channel in LDS
    uint32_t nvec;         // autocorrelation vectors per candidate
    uint32_t pcm_i16;      // input is interleaved int16 instead of int32
    uint32_t debug;
    uint32_t lds_dbuf_bytes;
    uint32_t limit_min_bitrate;   // a frame may not consist of CONSTANT subframes only (see the kernels' baseline stage)
};

struct FgBlockDesc {
    uint64_t pcm_off;      // first inter-channel sample of the block in the PCM buffer
    uint32_t n;            // samples per channel in this block
    uint32_t frame_number;
    uint32_t win_off;      // offset (floats) of this block length's window table
    uint32_t forced_ca;    // 0xFF = choose; else 0 or 3 (loose mid-side: the frames between two decisions copy the last one and
                           // evaluate only what they use; 3 | 0x80 = the decision frame itself when it chose mid/side: libFLAC
                           // evaluated all four candidates there, so limit_min_bitrate applies to mid and side as in any
                           // full frame -- a follower's mid/side are never limited)
    uint32_t out_slot;     // index of the output slot / result / debug record of this block
    uint32_t reserved;     // 0: the block's samples lie back to back (all channels interleaved); C > 0: a ONE-channel view of a
                           // stream of C interleaved channels -- pcm_off is then the ELEMENT offset of the view's first sample,
                           // the next one lies C elements on (fg_ctx.cpp: streams of more than two channels)
};

struct FgBlockResult {
    uint32_t bytes;
    uint32_t ca;
    uint32_t err;
    uint32_t best_bits[4];  // (one-channel blocks of the generic kernel: [3] = the frame's bits in front of its padding)
    uint32_t reserved;
};

struct FgDebugCand {
    uint32_t wasted, sbps;
    uint64_t fixed_tot[5];
    uint32_t fixed_guess, fixed_bits, nvec, pad0;
    uint32_t lpc_guess[FG_MAX_VEC], lpc_bits[FG_MAX_VEC];
    double autoc[FG_MAX_VEC][FG_MAX_ORDER + 1];
    uint32_t type, order, precision;
    int32_t shift;
    int32_t qlp[FG_MAX_ORDER];
    uint32_t rice_method, porder, bits, pad1;
    uint32_t rice_params[FG_MAX_PARTS];
};

struct FgDebugRec {
    FgDebugCand cand[FG_MAX_CAND];
    uint64_t t[16];    // clock64() stamps at stage boundaries
};


// ---- de-fused encode pipeline (flac_enc_pipe_impl.h): records handed from kernel to kernel through HBM
struct FgPipeDec {         // decision of one candidate of one block (K4 -> K5)
    uint32_t bits;         // size of the best subframe found for this candidate
    uint32_t type;         // 0 constant, 1 verbatim, 2 fixed, 3 lpc
    uint32_t order, prec;
    int32_t shift;
    uint32_t porder, method, wasted;
    int32_t q[12];
    uint8_t k[64];         // Rice parameter per partition
};

struct FgPipeBufs {
    double *autoc;         // [block][cand][nvec][MAXO + 1]                      K2 -> K3
    uint32_t *wasted;      // [block][cand]                                      K2 -> K3, K4
    uint32_t *nv;          // [block] autocorrelation vectors actually computed  K2 -> K3, K4
    int32_t *qres;         // [block][cand][nvec][MAXO]                          K3 -> K4
    uint32_t *lres;        // [block][cand][nvec]                                K3 -> K4
    FgPipeDec *dec;        // [block][cand]                                      K4 -> K5
    uint32_t *chunk_bits;  // [slot][4]                                          K5 -> size scan, K6
    unsigned long long *guard;   // [0] order guesses re-done with the exact log, [1] smallest margin seen (double bits)
    unsigned long long *stamp;   // when set: K2 leaves the start-of-call wall-clock stamp here (fg_signal_kernel reads it)
};

// Round 5, the direct packing path (fg_pipe_pack_kernel<..., DIRECT>): a workgroup assembles its whole frame in LDS, takes the CRC-16
// there and writes the bytes at the frame's final place in the output stream; the place comes from a decoupled look-back over the
// frame sizes (`lb`: one 64-bit word per output slot -- epoch of the call, state, value; a workgroup publishes its size as soon
// as it knows it, then the sum of everything up to and including itself once it has looked back).  Sizes scan, chunk round trip
// through HBM and the assembly kernel disappear from a launch.
struct FgPackDirect {
    unsigned long long *lb;             // [nblocks] look-back words
    unsigned long long *offsets;        // the context's offsets array: [slot] frame offset, [nblocks] total
    unsigned long long *user_offsets;   // the caller's frame index, or null
    uint8_t *dst;
    unsigned long long dst_cap;
    const uint16_t *crcx;               // CRC-16 constants for a pass of NT threads over 16-byte granules (fg_crc_tables_kernel)
    uint32_t epoch;                     // 20 bits, never 0 ... (a word of another call counts as empty)
    uint32_t nblocks;                   // blocks of the call (all kinds)
    uint32_t fcap_words;                // LDS frame buffer
    uint32_t reserved;
};
#define FG_LB_AGG 1ull                  // value = size of this frame
#define FG_LB_PFX 2ull                  // value = sum of the sizes up to and including this frame
#define FG_LB_POISON 3ull               // the chain is broken here (FG_ERR_CHAIN)
#define FG_LB_VBITS 42

struct FgPipeLaunch {
    const void *pcm;
    const FgBlockDesc *descs;
    const float *windows;
    FgEncParams P;
    FgPipeBufs B;
    uint32_t nblocks;
    uint8_t *slots;
    FgBlockResult *results;
    FgDebugRec *dbg;
    uint32_t chunk_cap_words;   // capacity of one wave's chunk inside the block's slot
    uint32_t fbw_words;         // LDS frame-bit window of one packing wave
    uint32_t nblocks_ws2;       // the first nblocks_ws2 blocks are packed by two waves per subframe, the rest by one
    uint32_t nblocks_rag;       // the LAST nblocks_rag blocks have the ragged lane geometry (flac_enc_pipe_impl.h PipeGeo)
    uint32_t acc64;             // > 16 bit samples
    uint32_t stages;            // bit 0: analysis (K2-K4), bit 1: pack (K5)
    double guard_thr;           // order guesses closer than this many bits are re-done with the correctly rounded log
    void *stream;               // hipStream_t
    void *stream2;              // side stream for the (few) blocks packed by one wave per subframe; events to fork and join
    void *ev_fork, *ev_join;
    // Round 4: the blocks of a launch are cut into `ngroups` contiguous groups, each with its own chain autocorrelation ->
    // Levinson-Durbin -> evaluation -> packing on its own stream (group 0 on `stream`): the kernels of different stages then
    // run beside each other where their ends and starts meet, and nobody's last, half-empty round of workgroups leaves the chip idle
    // (two groups: 0.485 -> 0.474 ms on the headline stream; more groups lose to their synchronisation, pipe_shape.inc).
    uint32_t no_keep;           // packing: the two-walk form for every block (FLACGPU_KEEP=0: cross-check of the kept-residual form)
    uint32_t no_autoc1;         // autocorrelation: 1 = fg_pipe_autoc_kernel also for launches of a few blocks, 2 = fg_pipe_autoc1_kernel for every launch, 3 = its windows never side by side (FLACGPU_AUTOC1=0 / 2 / 3: cross-checks)
    uint32_t ngroups;           // 0, 1: one chain on `stream`
    void *gstream[3];           // streams of groups 1..3
    void *gev_fork, *gev_join[3];
    // The start of a launch of several groups.  guard_clean: the counters the kernels add to were reset by the signal kernel of the
    // previous call (fg_signal_kernel `reset`), so no fg_pipe_begin_kernel runs in front of the groups -- group 0's
    // autocorrelation kernel takes the call's stamp --; no_fork: nothing this call queued on `stream` concerns the other groups
    // (same block list as the last call, no debug records to clear), so their streams do not wait for an event on it.  Kernel,
    // event record and waits were 14 us of idle GPU in front of the first autocorrelation kernel.
    uint32_t guard_clean, no_fork;
    // direct packing path: blocks [0, nblocks_direct) of the list (all packed by two waves per subframe) write their frames
    // themselves; the others (short blocks, ragged geometry) keep the chunk form, publish their sizes (fg_pipe_publish_kernel) and
    // are assembled by fg_pipe_assemble_kernel once everything is placed.  side_first: one of those others lies in front of a
    // direct block in the output, so their whole chain runs on `side_stream` beside the analysis of the direct blocks and the
    // direct packing kernels wait for `side_ev`; otherwise they ride at the end of the last group as before.
    uint32_t nblocks_direct;
    uint32_t side_first;
    uint32_t fused;             // the direct blocks are evaluated inside their packing kernel (fg_pipe_pack_kernel<FUSED>; flacgpu_set_direct(ctx, 2))
    uint32_t reserved5;
    void *side_stream, *side_ev;
    void *gev_eval[3];          // recorded behind the direct packing kernel of groups 0..2: the one of the next group waits for it
    FgPackDirect D;
};

// ---- decoder ----
struct FgDecFrame {
    uint64_t byte_off;     // frame start in the stream buffer
    uint64_t out_off;      // first inter-channel sample of this frame in the output
    uint32_t bytes;        // frame length including CRC-16
    uint32_t n;            // block size
    uint32_t hdr_bytes;    // header length including CRC-8
    uint32_t channels, ca, bps;
};

// One stream of a multi-stream decode call (flacgpu_decode_streams_dev): where its bytes start, how its frames are numbered and
// where its frames go in the frame table.
struct FgDecRange {
    uint64_t byte_start;
    uint64_t first_number;
    uint32_t slot_base;
    uint32_t nframes;
};

// Per-subframe predictor description handed from the parse kernel to the restore kernel.
struct FgDecSub {
    uint32_t order;        // samples stored verbatim at the start of the subframe (0 for CONSTANT / VERBATIM)
    int32_t shift;
    uint32_t wasted;
    uint32_t flags;        // bits 0-1 type (0 constant, 1 verbatim, 2 fixed, 3 lpc), 2-6 coefficient precision, 7-10 partition
                           // order, 11 five-bit Rice parameters, 12 record valid (frames of the generic decoder leave it 0)
    int32_t q[12];         // (constant subframe: q[0] = the value)         // FIR coefficients (quantised LPC, or the binomial coefficients of a fixed predictor)
};

// What lets the wave parser start from the frame OFFSETS alone (flac_dec_wave.hip): the index pass's resolve kernel leaves a packed
// header record per frame (fg_dec_hdr.h), the parser takes its frame's fields from it, applies the length rules, and places the
// frame's part of the residual plane at frame number x stride (stride = the largest block size among the first 64 records: a
// stream of fixed block size; a frame that does not fit takes the generic decoder) -- `planeoff[f]` tells the restore kernel
// where.  Header pass and scan of the block sizes then run beside the parser instead of in front of it.
// offsets == nullptr: the parser starts from the frame table (header pass and scan done), the plane lies in output order.
struct FgDecSelf {
    const unsigned long long *offsets;     // nframes + 1 frame positions
    const uint32_t *hdrrec;                // nframes packed header records (0: none)
    unsigned long long *planeoff;          // nframes: byte offset of the frame's part of the plane
    unsigned long long plane_cap_bytes;
    uint32_t si_bps;                       // STREAMINFO's, or 0
    uint32_t reserved;
};

// how long a wave waits for a word another kernel raises before it gives up and flags the call (wall clock, 100 MHz): 20 ms.
// (The side streams' work ends before the parser does; where kernels are serialised across streams -- a profiler collecting
// counters does that -- the restore kernel is started in front of the kernel that raises its word, waits this long once, and the
// context goes back to events.)
#define FG_GATE_TICKS 2000000ull

struct FgDecResult {
    uint32_t err;          // 0 ok, 1 malformed (bad header / reserved or inconsistent fields), 2 crc16 mismatch,
                           // 4 contents parse but do not end at the frame boundary (damaged residual or truncation),
                           // 5 contents parse but the padding bits before the CRC-16 are not zero
    uint32_t crc;
};
