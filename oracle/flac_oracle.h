/*
 * flac_oracle.h -- CPU restatement of the FLAC hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference (sonos/pyFLAC) delegates every hot-path operation to the
 * third-party libFLAC 1.4.3 (pin: /root/reference/.github/workflows/build.yml:9,
 * /root/reference/CHANGELOG.rst:14; call sites pyflac/encoder.py:115,132,319,401 and
 * pyflac/decoder.py:170,196,271,294,372,388).  libFLAC's source is NOT in
 * /root/reference, so this file restates its published algorithm (SURVEY.md
 * Appendix A = encoder, Appendix B = decoder; field widths from
 * pyflac/include/FLAC/format.h:191-484,536-557; level presets from
 * pyflac/include/FLAC/stream_encoder.h:845-853).
 *
 * Pinning: tests/test_oracle_vs_reference.py drives the reference's bundled
 * binary (pyflac/libraries/linux-x86_64/libFLAC-12.1.0.so) through
 * oracle/libflac_ref.py in the build container and requires byte-identical
 * output; tests/golden/ holds vectors generated from that binary
 * (oracle/gen_golden.py) so the pin travels to the GPU box.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * use this code.  The product (pyflac_amd/) never links or calls it.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLO_MAX_CHANNELS 8
#define FLO_MAX_LPC_ORDER 32
#define FLO_MAX_APOD_VECTORS 16
#define FLO_INFO_MAX_PARTS 256

typedef struct {
    uint32_t channels, bps, sample_rate, blocksize;
    uint32_t do_mid_side, loose_mid_side;
    uint32_t max_lpc_order, qlp_coeff_precision;
    uint32_t min_partition_order, max_partition_order;
    uint32_t apod_type;  /* 0 = tukey(p), 1 = subdivide_tukey(parts) with p/parts */
    float apod_p;
    uint32_t apod_parts;
    uint32_t streamable_subset;
    uint32_t do_md5;
    uint32_t limit_min_bitrate;   /* FLAC__stream_encoder_set_limit_min_bitrate: no frame of constant subframes only */
} flo_config;

/* One analysed candidate subframe (L, R, M or S). */
typedef struct {
    uint32_t wasted, sbps;
    uint64_t fixed_tot[5];
    uint32_t fixed_guess;
    uint32_t n_vectors;                                  /* autocorrelation vectors evaluated */
    double autoc[FLO_MAX_APOD_VECTORS][FLO_MAX_LPC_ORDER + 1];
    uint32_t lpc_guess[FLO_MAX_APOD_VECTORS];            /* chosen order per vector, 0 = skipped */
    uint32_t lpc_bits[FLO_MAX_APOD_VECTORS];             /* estimate per vector, 0 = skipped */
    uint32_t fixed_bits;                                 /* 0 = fixed not evaluated */
    /* winner */
    uint32_t type;                                       /* 0 CONSTANT 1 VERBATIM 2 FIXED 3 LPC */
    uint32_t order, precision;
    int32_t shift;
    int32_t qlp[FLO_MAX_LPC_ORDER];
    uint32_t rice_method, porder;
    uint32_t rice_params[FLO_INFO_MAX_PARTS];
    uint32_t bits;                                       /* estimated bits of the winner */
} flo_subframe_info;

typedef struct {
    uint32_t blocksize, channel_assignment;              /* 0 indep, 1 left/side, 2 right/side, 3 mid/side */
    uint32_t n_candidates;                               /* channels, or 4 for stereo with mid/side */
    flo_subframe_info cand[FLO_MAX_CHANNELS];            /* stereo+M/S: L, R, M, S */
    uint32_t frame_bytes;
} flo_frame_info;

typedef struct {
    uint32_t count;          /* loose mid-side frame counter */
    uint32_t last_ca;
} flo_loose_state;

/* Resolve a compression level the way FLAC__stream_encoder_set_compression_level +
 * init_stream do.  Returns a FLAC__StreamEncoderInitStatus code (0 = OK). */
int flo_config_from_level(flo_config *cfg, uint32_t level, uint32_t channels, uint32_t bps,
                          uint32_t sample_rate, uint32_t blocksize, uint32_t streamable_subset);

/* Window table for block length n: the first window of the configured apodization. */
void flo_window(const flo_config *cfg, uint32_t n, float *w);

/* Encode one frame of n inter-channel samples (interleaved int32).  Returns bytes written. */
size_t flo_encode_frame(const flo_config *cfg, const int32_t *interleaved, uint32_t n,
                        uint32_t frame_number, flo_loose_state *loose, uint8_t *out,
                        flo_frame_info *info);

/* 4 + 38 + 44 = 86 byte stream header (fLaC, STREAMINFO, VORBIS_COMMENT).
 * min/max framesize, total_samples, md5 may be 0 / NULL (stream mode). */
size_t flo_stream_header(const flo_config *cfg, uint32_t min_framesize, uint32_t max_framesize,
                         uint64_t total_samples, const uint8_t md5[16], uint8_t *out);

/* Whole stream: header + all frames (final short frame included).  If finalize != 0 the
 * STREAMINFO carries the final statistics (file mode), else zeros (stream mode).
 * frame_sizes (optional) receives each frame's byte count. Returns total bytes, 0 on overflow. */
size_t flo_encode_stream(const flo_config *cfg, const int32_t *interleaved, uint64_t nsamples,
                         int finalize, uint8_t *out, size_t cap, uint32_t *frame_sizes,
                         uint32_t *n_frames);

void flo_md5_pcm(const int32_t *interleaved, uint64_t nsamples, uint32_t channels, uint32_t bps,
                 uint8_t digest[16]);

/* ---- decoder ---- */
typedef struct {
    uint32_t min_blocksize, max_blocksize, min_framesize, max_framesize;
    uint32_t sample_rate, channels, bps;
    uint64_t total_samples;
    uint8_t md5[16];
    /* results */
    uint64_t decoded_samples;     /* inter-channel samples written */
    uint32_t n_frames;
    uint32_t n_errors;
    uint32_t errors[64];          /* FLAC__StreamDecoderErrorStatus codes, first 64 */
} flo_decode_result;

/* Decode a whole FLAC stream into interleaved int32.  Returns 0 on success (possibly with
 * recoverable errors listed in res), <0 on fatal error.  out may be NULL to count only. */
int flo_decode_stream(const uint8_t *data, size_t len, int32_t *out, uint64_t out_cap_samples,
                      flo_decode_result *res, uint32_t *frame_offsets, uint32_t frame_offsets_cap);

uint8_t flo_crc8(const uint8_t *p, size_t n);
uint16_t flo_crc16(const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
