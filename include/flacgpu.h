/*
 * flacgpu.h -- C ABI of libflacgpu.so, the MI355X-native FLAC core behind pyFLAC's API.
 *
 * Part 1 is the libFLAC entry-point subset that pyFLAC binds through cffi
 * (reference: pyflac/builder/encoder.py:266-322 and pyflac/builder/decoder.py:387-475).  Names,
 * signatures, enum values and callback contracts are those of libFLAC 1.4.3
 * (pyflac/include/FLAC/stream_encoder.h, stream_decoder.h, format.h), so the reference can link
 * against this library by changing only pyflac/builder/build_args.py:49-51.
 *
 * Part 2 is the batch extension the single-stream callback API cannot express: many blocks /
 * many streams per launch with PCM and output resident in HBM ("Extensions").
 *
 * Plain C, no torch / HIP types in any signature; device buffers are passed as void* addresses.
 */
#ifndef FLACGPU_H
#define FLACGPU_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ===================================================================== Part 1: libFLAC-compatible ABI */
typedef int FLAC__bool;
typedef uint8_t FLAC__byte;
typedef int32_t FLAC__int32;
typedef int64_t FLAC__int64;
typedef uint32_t FLAC__uint32;
typedef uint64_t FLAC__uint64;
typedef uint16_t FLAC__uint16;
typedef uint8_t FLAC__uint8;

/* ---- encoder enums (pyflac/builder/encoder.py:51-106) */
typedef enum {
    FLAC__STREAM_ENCODER_OK = 0,
    FLAC__STREAM_ENCODER_UNINITIALIZED,
    FLAC__STREAM_ENCODER_OGG_ERROR,
    FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR,
    FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA,
    FLAC__STREAM_ENCODER_CLIENT_ERROR,
    FLAC__STREAM_ENCODER_IO_ERROR,
    FLAC__STREAM_ENCODER_FRAMING_ERROR,
    FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR
} FLAC__StreamEncoderState;
extern const char *const FLAC__StreamEncoderStateString[];

typedef enum {
    FLAC__STREAM_ENCODER_INIT_STATUS_OK = 0,
    FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR,
    FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION,
    FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER,
    FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE,
    FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA,
    FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamEncoderInitStatus;
extern const char *const FLAC__StreamEncoderInitStatusString[];

typedef enum {
    FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE,
    FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM,
    FLAC__STREAM_ENCODER_READ_STATUS_ABORT,
    FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED
} FLAC__StreamEncoderReadStatus;
typedef enum {
    FLAC__STREAM_ENCODER_SEEK_STATUS_OK,
    FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR,
    FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED
} FLAC__StreamEncoderSeekStatus;
typedef enum {
    FLAC__STREAM_ENCODER_TELL_STATUS_OK,
    FLAC__STREAM_ENCODER_TELL_STATUS_ERROR,
    FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED
} FLAC__StreamEncoderTellStatus;
typedef enum {
    FLAC__STREAM_ENCODER_WRITE_STATUS_OK = 0,
    FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR
} FLAC__StreamEncoderWriteStatus;

/* Opaque two-pointer handles (pyflac/builder/encoder.py:109-114, decoder.py:139-144). */
typedef struct {
    struct FLAC__StreamEncoderProtected *protected_;
    struct FLAC__StreamEncoderPrivate *private_;
} FLAC__StreamEncoder;
typedef struct {
    struct FLAC__StreamDecoderProtected *protected_;
    struct FLAC__StreamDecoderPrivate *private_;
} FLAC__StreamDecoder;

/* ---- metadata (pyflac/builder/encoder.py:117-248, format.h:506-880).  The encoder produces STREAMINFO (and the vendor
 * VORBIS_COMMENT in the stream header); the decoder parses every block type and hands the ones the client asked for
 * (FLAC__stream_decoder_set_metadata_respond*) to the metadata callback.  Field order and types are the ABI. */
typedef enum {
    FLAC__METADATA_TYPE_STREAMINFO = 0,
    FLAC__METADATA_TYPE_PADDING = 1,
    FLAC__METADATA_TYPE_APPLICATION = 2,
    FLAC__METADATA_TYPE_SEEKTABLE = 3,
    FLAC__METADATA_TYPE_VORBIS_COMMENT = 4,
    FLAC__METADATA_TYPE_CUESHEET = 5,
    FLAC__METADATA_TYPE_PICTURE = 6,
    FLAC__METADATA_TYPE_UNDEFINED = 7,
    FLAC__MAX_METADATA_TYPE = 126
} FLAC__MetadataType;

typedef struct {
    uint32_t min_blocksize, max_blocksize;
    uint32_t min_framesize, max_framesize;
    uint32_t sample_rate;
    uint32_t channels;
    uint32_t bits_per_sample;
    FLAC__uint64 total_samples;
    FLAC__byte md5sum[16];
} FLAC__StreamMetadata_StreamInfo;

typedef struct { int dummy; } FLAC__StreamMetadata_Padding;

typedef struct {
    FLAC__byte id[4];
    FLAC__byte *data;                  /* length - 4 bytes */
} FLAC__StreamMetadata_Application;

typedef struct {
    FLAC__uint64 sample_number;
    FLAC__uint64 stream_offset;
    uint32_t frame_samples;
} FLAC__StreamMetadata_SeekPoint;

typedef struct {
    uint32_t num_points;
    FLAC__StreamMetadata_SeekPoint *points;
} FLAC__StreamMetadata_SeekTable;

typedef struct {
    FLAC__uint32 length;
    FLAC__byte *entry;                 /* NUL-terminated copy of the length bytes */
} FLAC__StreamMetadata_VorbisComment_Entry;

typedef struct {
    FLAC__StreamMetadata_VorbisComment_Entry vendor_string;
    FLAC__uint32 num_comments;
    FLAC__StreamMetadata_VorbisComment_Entry *comments;
} FLAC__StreamMetadata_VorbisComment;

typedef struct {
    FLAC__uint64 offset;
    FLAC__byte number;
} FLAC__StreamMetadata_CueSheet_Index;

typedef struct {
    FLAC__uint64 offset;
    FLAC__byte number;
    char isrc[13];
    uint32_t type : 1;
    uint32_t pre_emphasis : 1;
    FLAC__byte num_indices;
    FLAC__StreamMetadata_CueSheet_Index *indices;
} FLAC__StreamMetadata_CueSheet_Track;

typedef struct {
    char media_catalog_number[129];
    FLAC__uint64 lead_in;
    FLAC__bool is_cd;
    uint32_t num_tracks;
    FLAC__StreamMetadata_CueSheet_Track *tracks;
} FLAC__StreamMetadata_CueSheet;

typedef int FLAC__StreamMetadata_Picture_Type;   /* format.h:734-758: 0 other ... 3 front cover ... 20 publisher logotype */

typedef struct {
    FLAC__StreamMetadata_Picture_Type type;
    char *mime_type;
    FLAC__byte *description;
    FLAC__uint32 width;
    FLAC__uint32 height;
    FLAC__uint32 depth;
    FLAC__uint32 colors;
    FLAC__uint32 data_length;
    FLAC__byte *data;
} FLAC__StreamMetadata_Picture;

typedef struct { FLAC__byte *data; } FLAC__StreamMetadata_Unknown;

typedef struct {
    FLAC__MetadataType type;
    FLAC__bool is_last;
    uint32_t length;
    union {
        FLAC__StreamMetadata_StreamInfo stream_info;
        FLAC__StreamMetadata_Padding padding;
        FLAC__StreamMetadata_Application application;
        FLAC__StreamMetadata_SeekTable seek_table;
        FLAC__StreamMetadata_VorbisComment vorbis_comment;
        FLAC__StreamMetadata_CueSheet cue_sheet;
        FLAC__StreamMetadata_Picture picture;
        FLAC__StreamMetadata_Unknown unknown;
    } data;
} FLAC__StreamMetadata;

/* ---- encoder callbacks (pyflac/builder/encoder.py:251-256) */
typedef FLAC__StreamEncoderReadStatus (*FLAC__StreamEncoderReadCallback)(const FLAC__StreamEncoder *encoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
typedef FLAC__StreamEncoderWriteStatus (*FLAC__StreamEncoderWriteCallback)(const FLAC__StreamEncoder *encoder, const FLAC__byte buffer[], size_t bytes, uint32_t samples, uint32_t current_frame, void *client_data);
typedef FLAC__StreamEncoderSeekStatus (*FLAC__StreamEncoderSeekCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamEncoderTellStatus (*FLAC__StreamEncoderTellCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef void (*FLAC__StreamEncoderMetadataCallback)(const FLAC__StreamEncoder *encoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamEncoderProgressCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 bytes_written, FLAC__uint64 samples_written, uint32_t frames_written, uint32_t total_frames_estimate, void *client_data);

/* ---- encoder functions (pyflac/builder/encoder.py:266-322) */
FLAC__StreamEncoder *FLAC__stream_encoder_new(void);
void FLAC__stream_encoder_delete(FLAC__StreamEncoder *encoder);

/* Not part of pyFLAC's cdef (pyflac/builder/encoder.py:266-322) but of libFLAC's encoder interface (stream_encoder.h:1214): the
 * metadata blocks written behind STREAMINFO, in the order given (native FLAC keeps the caller's order; only Ogg FLAC moves the
 * VORBIS_COMMENT to the front).  PADDING, APPLICATION, SEEKTABLE (verbatim), VORBIS_COMMENT (takes the place of the default one,
 * vendor string replaced by libFLAC's), CUESHEET, PICTURE and unknown types.  Lists libFLAC refuses at init -- a STREAMINFO
 * block, two SEEKTABLEs or VORBIS_COMMENTs, an unsorted SEEKTABLE, a CUESHEET or PICTURE that fails
 * FLAC__format_cuesheet_is_legal / FLAC__format_picture_is_legal -- give FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA. */
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *encoder, FLAC__StreamMetadata **metadata, uint32_t num_blocks);
FLAC__bool FLAC__stream_encoder_set_verify(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_channels(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_bits_per_sample(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_sample_rate(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_blocksize(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_do_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_loose_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *encoder, const char *specification);
FLAC__bool FLAC__stream_encoder_set_max_lpc_order(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_qlp_coeff_precision(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_do_qlp_coeff_prec_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_exhaustive_model_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_min_residual_partition_order(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_max_residual_partition_order(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_rice_parameter_search_dist(FLAC__StreamEncoder *encoder, uint32_t value);
FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *encoder, FLAC__uint64 value);
FLAC__bool FLAC__stream_encoder_set_streamable_subset(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_limit_min_bitrate(FLAC__StreamEncoder *encoder, FLAC__bool value);
/* exported by libFLAC 1.4.3 (stream_encoder.h) though not in pyFLAC's cdef; used by bench.py */
FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);

FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *encoder);
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *encoder);
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_sample, uint32_t *frame_number, uint32_t *channel, uint32_t *sample, FLAC__int32 *expected, FLAC__int32 *got);
FLAC__bool FLAC__stream_encoder_get_verify(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_streamable_subset(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_channels(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_bits_per_sample(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_sample_rate(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_mid_side_stereo(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_loose_mid_side_stereo(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_max_lpc_order(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_qlp_coeff_precision(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_qlp_coeff_prec_search(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_escape_coding(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_exhaustive_model_search(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_min_residual_partition_order(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_max_residual_partition_order(const FLAC__StreamEncoder *encoder);
uint32_t FLAC__stream_encoder_get_rice_parameter_search_dist(const FLAC__StreamEncoder *encoder);
FLAC__uint64 FLAC__stream_encoder_get_total_samples_estimate(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_limit_min_bitrate(const FLAC__StreamEncoder *encoder);

FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderReadCallback read_callback, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *encoder, const FLAC__int32 *const buffer[], uint32_t samples);
FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *encoder, const FLAC__int32 buffer[], uint32_t samples);

/* ---- decoder enums (pyflac/builder/decoder.py:49-136) */
typedef enum {
    FLAC__STREAM_DECODER_SEARCH_FOR_METADATA = 0,
    FLAC__STREAM_DECODER_READ_METADATA,
    FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC,
    FLAC__STREAM_DECODER_READ_FRAME,
    FLAC__STREAM_DECODER_END_OF_STREAM,
    FLAC__STREAM_DECODER_OGG_ERROR,
    FLAC__STREAM_DECODER_SEEK_ERROR,
    FLAC__STREAM_DECODER_ABORTED,
    FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR,
    FLAC__STREAM_DECODER_UNINITIALIZED
} FLAC__StreamDecoderState;
extern const char *const FLAC__StreamDecoderStateString[];

typedef enum {
    FLAC__STREAM_DECODER_INIT_STATUS_OK = 0,
    FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER,
    FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS,
    FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR,
    FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE,
    FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamDecoderInitStatus;
extern const char *const FLAC__StreamDecoderInitStatusString[];

typedef enum {
    FLAC__STREAM_DECODER_READ_STATUS_CONTINUE,
    FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM,
    FLAC__STREAM_DECODER_READ_STATUS_ABORT
} FLAC__StreamDecoderReadStatus;
typedef enum {
    FLAC__STREAM_DECODER_SEEK_STATUS_OK,
    FLAC__STREAM_DECODER_SEEK_STATUS_ERROR,
    FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED
} FLAC__StreamDecoderSeekStatus;
typedef enum {
    FLAC__STREAM_DECODER_TELL_STATUS_OK,
    FLAC__STREAM_DECODER_TELL_STATUS_ERROR,
    FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED
} FLAC__StreamDecoderTellStatus;
typedef enum {
    FLAC__STREAM_DECODER_LENGTH_STATUS_OK,
    FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR,
    FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED
} FLAC__StreamDecoderLengthStatus;
typedef enum {
    FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE,
    FLAC__STREAM_DECODER_WRITE_STATUS_ABORT
} FLAC__StreamDecoderWriteStatus;
typedef enum {
    FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC,
    FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER,
    FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH,
    FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM,
    FLAC__STREAM_DECODER_ERROR_STATUS_BAD_METADATA
} FLAC__StreamDecoderErrorStatus;
extern const char *const FLAC__StreamDecoderErrorStatusString[];

typedef enum {
    FLAC__FRAME_NUMBER_TYPE_FRAME_NUMBER,
    FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER
} FLAC__FrameNumberType;
typedef enum {
    FLAC__CHANNEL_ASSIGNMENT_INDEPENDENT = 0,
    FLAC__CHANNEL_ASSIGNMENT_LEFT_SIDE = 1,
    FLAC__CHANNEL_ASSIGNMENT_RIGHT_SIDE = 2,
    FLAC__CHANNEL_ASSIGNMENT_MID_SIDE = 3
} FLAC__ChannelAssignment;
typedef enum {
    FLAC__SUBFRAME_TYPE_CONSTANT = 0,
    FLAC__SUBFRAME_TYPE_VERBATIM = 1,
    FLAC__SUBFRAME_TYPE_FIXED = 2,
    FLAC__SUBFRAME_TYPE_LPC = 3
} FLAC__SubframeType;
typedef enum {
    FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE = 0,
    FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 = 1
} FLAC__EntropyCodingMethodType;

/* ---- FLAC__Frame (pyflac/builder/decoder.py:146-231).  pyFLAC reads only
 * header.{blocksize,sample_rate,channels,bits_per_sample} (pyflac/decoder.py:500-524); the subframe
 * records are filled with type / order / wasted bits, residual pointers are NULL. */
typedef struct {
    uint32_t blocksize;
    uint32_t sample_rate;
    uint32_t channels;
    FLAC__ChannelAssignment channel_assignment;
    uint32_t bits_per_sample;
    FLAC__FrameNumberType number_type;
    union {
        FLAC__uint32 frame_number;
        FLAC__uint64 sample_number;
    } number;
    FLAC__uint8 crc;
} FLAC__FrameHeader;

typedef struct {
    uint32_t *parameters;
    uint32_t *raw_bits;
    uint32_t capacity_by_order;
} FLAC__EntropyCodingMethod_PartitionedRiceContents;
typedef struct {
    uint32_t order;
    const FLAC__EntropyCodingMethod_PartitionedRiceContents *contents;
} FLAC__EntropyCodingMethod_PartitionedRice;
typedef struct {
    FLAC__EntropyCodingMethodType type;
    union {
        FLAC__EntropyCodingMethod_PartitionedRice partitioned_rice;
    } data;
} FLAC__EntropyCodingMethod;
typedef struct {
    FLAC__int64 value;
} FLAC__Subframe_Constant;
typedef enum {
    FLAC__VERBATIM_SUBFRAME_DATA_TYPE_INT32 = 0,
    FLAC__VERBATIM_SUBFRAME_DATA_TYPE_INT64 = 1
} FLAC__VerbatimSubframeDataType;
typedef struct {
    union {
        const FLAC__int32 *int32;
        const FLAC__int64 *int64;
    } data;
    FLAC__VerbatimSubframeDataType data_type;
} FLAC__Subframe_Verbatim;
typedef struct {
    FLAC__EntropyCodingMethod entropy_coding_method;
    uint32_t order;
    FLAC__int64 warmup[4];
    const FLAC__int32 *residual;
} FLAC__Subframe_Fixed;
typedef struct {
    FLAC__EntropyCodingMethod entropy_coding_method;
    uint32_t order;
    uint32_t qlp_coeff_precision;
    int quantization_level;
    FLAC__int32 qlp_coeff[32];
    FLAC__int64 warmup[32];
    const FLAC__int32 *residual;
} FLAC__Subframe_LPC;
typedef struct {
    FLAC__SubframeType type;
    union {
        FLAC__Subframe_Constant constant;
        FLAC__Subframe_Fixed fixed;
        FLAC__Subframe_LPC lpc;
        FLAC__Subframe_Verbatim verbatim;
    } data;
    uint32_t wasted_bits;
} FLAC__Subframe;
typedef struct {
    FLAC__uint16 crc;
} FLAC__FrameFooter;
typedef struct {
    FLAC__FrameHeader header;
    FLAC__Subframe subframes[8];
    FLAC__FrameFooter footer;
} FLAC__Frame;

/* ---- decoder callbacks (pyflac/builder/decoder.py:368-375) */
typedef FLAC__StreamDecoderReadStatus (*FLAC__StreamDecoderReadCallback)(const FLAC__StreamDecoder *decoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
typedef FLAC__StreamDecoderSeekStatus (*FLAC__StreamDecoderSeekCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderTellStatus (*FLAC__StreamDecoderTellCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderLengthStatus (*FLAC__StreamDecoderLengthCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *stream_length, void *client_data);
typedef FLAC__bool (*FLAC__StreamDecoderEofCallback)(const FLAC__StreamDecoder *decoder, void *client_data);
typedef FLAC__StreamDecoderWriteStatus (*FLAC__StreamDecoderWriteCallback)(const FLAC__StreamDecoder *decoder, const FLAC__Frame *frame, const FLAC__int32 *const buffer[], void *client_data);
typedef void (*FLAC__StreamDecoderMetadataCallback)(const FLAC__StreamDecoder *decoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamDecoderErrorCallback)(const FLAC__StreamDecoder *decoder, FLAC__StreamDecoderErrorStatus status, void *client_data);

/* ---- decoder functions (pyflac/builder/decoder.py:387-475) */
FLAC__StreamDecoder *FLAC__stream_decoder_new(void);
void FLAC__stream_decoder_delete(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_set_md5_checking(FLAC__StreamDecoder *decoder, FLAC__bool value);
FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *decoder, FLAC__MetadataType type);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *decoder, FLAC__MetadataType type);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *decoder);
FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *decoder);
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *decoder);
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *decoder);
uint32_t FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *decoder);
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *decoder);
uint32_t FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *decoder);
uint32_t FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *decoder);
uint32_t FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *decoder, FLAC__uint64 *position);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *decoder, FLAC__uint64 sample);

extern const char *FLAC__VERSION_STRING;
extern const char *FLAC__VENDOR_STRING;

/* ===================================================================== Part 2: batch extension */
typedef struct flacgpu_ctx flacgpu_ctx;

/* Encoder settings after level/blocksize resolution (what FLAC__stream_encoder_set_compression_level
 * + init_stream establish; stream_encoder.h:845-853). */
typedef struct {
    uint32_t channels, bits_per_sample, sample_rate, blocksize;
    uint32_t do_mid_side, loose_mid_side;
    uint32_t max_lpc_order, qlp_coeff_precision;
    uint32_t min_partition_order, max_partition_order;
    uint32_t apod_parts;          /* 0 = tukey(0.5); n >= 2 = subdivide_tukey(n) */
    uint32_t streamable_subset;
    uint32_t limit_min_bitrate;   /* FLAC__stream_encoder_set_limit_min_bitrate (stream_encoder.h): no frame of CONSTANT subframes only */
} flacgpu_settings;

/* Returns a FLAC__StreamEncoderInitStatus value (0 = OK). */
int flacgpu_settings_from_level(flacgpu_settings *s, uint32_t level, uint32_t channels, uint32_t bits_per_sample,
                                uint32_t sample_rate, uint32_t blocksize, int streamable_subset);

int flacgpu_device_count(void);
flacgpu_ctx *flacgpu_ctx_create(int device);           /* NULL on failure; see flacgpu_last_error() */
void flacgpu_ctx_destroy(flacgpu_ctx *ctx);
const char *flacgpu_last_error(void);

/* One stream of a batch: `nsamples` inter-channel samples starting at sample index `pcm_offset` of the
 * device PCM buffer; frames are numbered from first_frame.  A stream may be handed over in several calls
 * (first_frame = frames already encoded): with loose mid-side (levels 1 and 4) libFLAC decides the channel
 * assignment on every `period`-th frame of the STREAM and the frames in between copy it
 * (stream_encoder.h:826-838), so a continuation call passes the previous call's
 * flacgpu_encode_stats.last_channel_assignment in prev_channel_assignment (0 at the start of a stream). */
typedef struct {
    uint64_t pcm_offset;
    uint64_t nsamples;
    uint32_t first_frame;
    uint32_t prev_channel_assignment;
} flacgpu_stream_desc;

typedef struct {
    uint32_t nblocks;             /* frames produced */
    uint32_t error_flags;         /* OR of per-block FG_ERR_* bits; 0 = ok */
    uint64_t total_bytes;         /* bytes written to d_out */
    float encode_kernel_ms;       /* HIP-event time of the frame-encode kernels (analysis + packing); timing level >= 1 */
    float total_gpu_ms;           /* GPU time of the whole enqueue (encode + frame assembly): wall-clock stamps, or HIP events at level >= 1 */
    uint32_t last_channel_assignment;  /* loose mid-side: assignment (0 independent / 3 mid-side) of the last frame of the last stream */
    uint32_t redo_blocks;         /* blocks the specialised kernels handed to the generic kernel */
    float stage_ms[8];            /* with flacgpu_set_stage_timing(ctx, 2): analysis, packing, sizes + scan, assembly + CRC-16
                                     (HIP events between the kernel groups); else zeros */
    uint32_t log_guard_subframes; /* LPC order guesses that were within the guard threshold of a tie and were re-done with the
                                     correctly rounded logarithm (see flacgpu_set_log_guard) */
    uint32_t direct_path;         /* 1: the packing kernel wrote the frames at their final place (round 5; flacgpu_set_direct); 2: it
                                     tried, a frame could not be placed, and the call repeated packing, scan and assembly in the
                                     chunk form; 0: chunk form */
    double lpc_order_min_margin;  /* smallest distance (bits) between the best and the second-best order estimate in this call */
} flacgpu_encode_stats;

/* Encode every block of every stream.  d_pcm: device address of interleaved PCM (int32, or int16 when
 * pcm_is_i16).  d_out receives the frames back to back in (stream, block) order; d_frame_offsets
 * (device, nblocks+1 uint64) their byte offsets.  Blocks are settings->blocksize long, the last block of a
 * stream may be shorter.  Synchronous: returns when the GPU work is complete.  0 on success. */
int flacgpu_encode_streams(flacgpu_ctx *ctx, const flacgpu_settings *settings, const void *d_pcm, int pcm_is_i16,
                           const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_out, uint64_t out_capacity,
                           void *d_frame_offsets, flacgpu_encode_stats *stats);

/* STREAMINFO's md5sum (format.h:543) for device-resident streams: MD5 over the samples, little-endian at
 * (bits_per_sample + 7) / 8 bytes each, channels interleaved -- what libFLAC hashes and what FLAC__stream_encoder_finish writes into
 * STREAMINFO.  flacgpu_encode_streams does not compute it (the hash is ONE serial chain per stream: the GPU gives it a thread per
 * stream, about 60 MB/s each, so 128 streams of 60 s take 0.2 s where their encode launch takes 5 ms); this call does, on a
 * stream of its own, for callers that finalise STREAMINFO themselves.  d_md5: 16 bytes per stream (device).  *gpu_ms: kernel time.
 * May be called from another thread while an encode call on the same context runs.  0 on success. */
int flacgpu_md5_streams(flacgpu_ctx *ctx, const void *d_pcm, int pcm_is_i16, uint32_t channels, uint32_t bits_per_sample,
                        const flacgpu_stream_desc *streams, uint32_t nstreams, void *d_md5, float *gpu_ms);

/* Upper bound of the encoded size of the given batch (for sizing d_out). */
uint64_t flacgpu_encode_bound(const flacgpu_settings *settings, const flacgpu_stream_desc *streams, uint32_t nstreams,
                              uint32_t *nblocks);

/* Debug: per-block analysis records of the last flacgpu_encode_streams call made with
 * flacgpu_set_debug(ctx, 1).  Layout: FgDebugRec (pyflac_amd/csrc/fg_types.h). */
void flacgpu_set_debug(flacgpu_ctx *ctx, int on);
/* libFLAC picks the LPC order by comparing estimates that contain log(); the device's log and glibc's may differ in the last
 * bit, which can only matter when two orders are within ~1e-10 bits of each other.  Estimates closer than `threshold_bits`
 * (default 1e-6) are re-done with a correctly rounded logarithm and counted in flacgpu_encode_stats.log_guard_subframes. */
void flacgpu_set_log_guard(flacgpu_ctx *ctx, double threshold_bits);
/* Direct packing (default on): blocks packed by two waves per subframe assemble their frame in LDS and store it at its final place in
 * d_out (sizes by decoupled look-back); 0 = every block hands chunks through HBM to the scan and assembly kernels, as in rounds 2-4.
 * Same bytes either way (tests cross-check the two). */
void flacgpu_set_direct(flacgpu_ctx *ctx, int on);
/* (on = 2: direct packing with the evaluation of the candidates inside the same kernel -- blocks of 4096 samples, <= 16 bit: the PCM
 * is staged and read once for both.  1: direct packing behind a separate evaluation kernel.) */
/* Window self-check: empty string, or a note that this host's cosf produced a tukey taper different from the committed one
 * (the committed one is then used; the note is also left in flacgpu_last_error() when it happens). */
const char *flacgpu_window_note(flacgpu_ctx *ctx);
/* How the library was built: bit 0 = `make TUNING=1` (experiment and diagnostic environment switches are read), bit 1 = `make LEGACY=1`
 * (the superseded kernels of rounds 1 and 2 are present and selectable: FLACGPU_PIPE=0, FLACGPU_DEC_WAVE=0, FLACGPU_DEC_FUSED=0). */
unsigned int flacgpu_build_flags(void);
/* Sixteen hex digits each: hashes over the sources this library was built from -- all of them (build id), the kernel files and the
 * headers they include (kernel id), the host files and theirs (host id).  The committed counter passes (profiles: the _pmc json
 * files) name the build they were measured on; bench.py quotes their HBM traffic and instruction counts only when the KERNEL id
 * matches: an edit of a host file changes the build id and the host id and leaves the counter passes standing. */
const char *flacgpu_build_id(void);
const char *flacgpu_kernel_id(void);
const char *flacgpu_host_id(void);
/* Start-up self-check of the encoder's matrix-core autocorrelation (run by flacgpu_ctx_create): the number of
 * v_mfma_f64_4x4x4_4b_f64 results that differed from the chain of v_fma_f64 the bit-exactness of stage L6 rests on; 0 on
 * a device that behaves like the MI355X this was written on.  Non-zero: every block is encoded by the generic kernel (same bytes,
 * slower), *note says so.  flacgpu_force_selfcheck_result overrides the outcome (tests of that fallback). */
int flacgpu_selfcheck(flacgpu_ctx *ctx, const char **note);
void flacgpu_force_selfcheck_result(flacgpu_ctx *ctx, int mfma_bad);
/* What the batch calls time, and how they end.  level 0 (default): no HIP events (each record idles the GPU for a few
 * microseconds between two kernels); the call ends with a kernel that writes totals and wall-clock stamps into pinned memory,
 * which the host polls; total_gpu_ms comes from the stamps, the *_kernel_ms / index_ms / stage_ms fields stay 0.
 * level 1: HIP events around the call and around its kernel groups (all *_ms fields but stage_ms).  level 2: also between the
 * encoder's stages (stage_ms).  level 3: as level 0, plus one event in front of the call's first kernel and one behind its last:
 * total_gpu_ms is the HIP-event time of exactly the kernels, in exactly the order, the default call runs. */
void flacgpu_set_stage_timing(flacgpu_ctx *ctx, int level);
int flacgpu_copy_debug(flacgpu_ctx *ctx, void *host_dst, uint32_t first_block, uint32_t nblocks);
/* The kernels' CRC-16 tables as the library builds them on the host (13 312 uint16 entries; layout: fg_ctx.cpp fg_crc_tables_host):
 * needs no GPU, so that a CPU test can hold them against the bit-by-bit definition of FLAC__crc16 (format.h:447). */
void flacgpu_debug_crc_tables(uint16_t *out);
int flacgpu_copy_block_results(flacgpu_ctx *ctx, void *host_dst, uint32_t nblocks);

typedef struct {
    uint32_t nframes;
    uint32_t error_frames;        /* frames with CRC-16 mismatch or malformed contents */
    uint64_t total_samples;       /* inter-channel samples written */
    uint32_t channels, bits_per_sample, sample_rate, max_blocksize;
    float decode_kernel_ms;
    float total_gpu_ms;
    float index_ms;               /* flacgpu_decode_stream_dev: HIP-event time of the frame index pass (part of total_gpu_ms) */
    uint32_t plane_bits;          /* width of the residual plane between the parse and the restore kernel: 16 for streams of up to 16 bits
                                   * (0 where no plane is used); 32 after a frame showed a value beyond 16 bits and the call was repeated */
    uint32_t generic_frames;      /* frames the wave parser handed to the generic (one lane a frame) decoder: predictor orders above 12,
                                   * 33-bit subframes, codes of hundreds of bytes, a frame whose part of the residual plane did not fit.
                                   * Same samples, much slower: a large count on ordinary material is a defect to report */
    uint32_t join_late_workgroups;/* workgroups of the restore kernel whose first look at the join word -- the word the kernel behind header pass,
                                   * scan and CRC pass raises -- found it not yet raised, i.e. that waited for those kernels (0 in a call
                                   * that joins through events; normally 0 anyway: they end before the parser does) */
} flacgpu_decode_stats;

/* Decode the audio frames of one FLAC stream held in device memory.  d_stream/len: the frame data (device);
 * the frame index (h_frame_offsets: nframes+1 byte offsets into d_stream) comes from flacgpu_index_frames.
 * d_pcm receives interleaved int32 (nsamples_total * channels).  0 on success. */
int flacgpu_decode_frames(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, const uint64_t *h_frame_offsets,
                          uint32_t nframes, uint32_t channels_hint, uint32_t bps_hint, void *d_pcm,
                          uint64_t pcm_capacity_samples, void *h_frame_status, flacgpu_decode_stats *stats);

/* The same with the frame index already in device memory (d_frame_offsets: nframes+1 uint64 byte offsets), e.g. the
 * offsets flacgpu_encode_streams wrote: no host round trip of the index. */
int flacgpu_decode_frames_dev(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, const uint64_t *d_frame_offsets,
                              uint32_t nframes, uint32_t channels_hint, uint32_t bps_hint, void *d_pcm,
                              uint64_t pcm_capacity_samples, void *h_frame_status, flacgpu_decode_stats *stats);

/* Decode from the bytes alone: the frame index is made on the device (one pass over the stream: sync code, header
 * fields, header CRC-8; every header found is filed under its frame number -- format.h:418-475), then the frames are decoded
 * as above and their CRC-16 checked.  d_stream/len: the audio frames of ONE fixed-block-size stream (what follows the
 * metadata blocks).  nframes_hint: the number of frames when known (STREAMINFO: ceil(total_samples / blocksize)), 0 to have
 * them counted (one more pass and a host round trip).  first_frame_number: number of the first frame (0 for a whole
 * stream).  h_frame_status (optional) receives at most status_capacity rows of {status, crc} -- stats->nframes tells how
 * many frames were found.  d_frame_offsets_out (optional, device, nframes+1 uint64) receives the index.  Streams this does not cover
 * (variable block size, two headers claiming one frame number) fail with a message; flacgpu_index_frames handles them. */
int flacgpu_decode_stream_dev(flacgpu_ctx *ctx, const void *d_stream, uint64_t len, uint32_t nframes_hint, uint64_t first_frame_number,
                              uint32_t channels, uint32_t bits_per_sample, void *d_pcm, uint64_t pcm_capacity_samples,
                              void *h_frame_status, uint64_t status_capacity, void *d_frame_offsets_out, flacgpu_decode_stats *stats);

/* The same for several streams laid back to back in one device buffer (BASELINE config 5: a batch of independent streams,
 * SURVEY.md section 8e): one pass over all bytes finds every stream's frames, every stream numbers its frames from its own
 * first_frame_number, and the PCM of the streams comes out back to back in stream order.  The ranges must cover the buffer
 * without gaps, in order; nframes of every stream is required (STREAMINFO).  h_frame_status (optional) receives at most
 * status_capacity rows of {status, crc}; stats->nframes tells how many there are. */
typedef struct {
    uint64_t byte_offset, byte_length;     /* the audio frames of the stream inside the buffer */
    uint64_t first_frame_number;
    uint32_t nframes;
    uint32_t reserved;
} flacgpu_stream_range;
int flacgpu_decode_streams_dev(flacgpu_ctx *ctx, const void *d_bytes, uint64_t len, const flacgpu_stream_range *ranges, uint32_t nranges,
                               uint32_t channels, uint32_t bits_per_sample, void *d_pcm, uint64_t pcm_capacity_samples,
                               void *h_frame_status, uint64_t status_capacity, void *d_frame_offsets_out, flacgpu_decode_stats *stats);

/* How much of FLAC__Frame.subframes[] the decoder's write callback sees (format.h:285-396).  0: nothing, 1 (default): type,
 * wasted bits, order, precision, shift, coefficients, warm-up samples, partition order, Rice parameters (the pointers stay
 * valid until the next frame is delivered), 2: also the `residual` / verbatim `data` arrays (costs one more copy of the size
 * of the PCM from the device).  Frames the generic decode kernel handles (predictor order > 12, 33-bit side channels) and
 * subframes with a residual partition order above 8 (non-subset streams; the Rice parameters of 256 partitions are kept) carry
 * no subframe details. */
void flacgpu_stream_decoder_set_subframe_detail(FLAC__StreamDecoder *decoder, int level);

/* Extension: block delivery.  With a block callback set (before FLAC__stream_decoder_init_*), the decoder hands over every
 * round of decoded frames in ONE call instead of one write callback per frame (stream_decoder.h:463-500): `pcm` holds the
 * frames back to back, channels interleaved, as int16 when no frame of the round has more than 16 bits per sample
 * (bytes_per_sample 2) and int32 otherwise (4); blocks[i] describes frame i -- `offset` is where it starts in `pcm`, in
 * inter-channel samples, or FLACGPU_BLOCK_SILENCE for a frame of silence the decoder fills a gap with (stream damage).  The
 * order of blocks and of error callbacks is the order of the write and error callbacks without it.  A caller that turns the
 * blocks into arrays of its own (pyflac_amd.StreamDecoder) saves the per-frame crossing into its language and the widening
 * to FLAC__int32.  `pcm` is valid during the call.  Not used while MD5 checking is on (the write callback serves then).
 * FLAC__stream_decoder_process_single delivers one block per call. */
typedef struct {
    uint64_t sample_number;       /* of the frame's first sample */
    uint64_t offset;              /* start in `pcm`, in inter-channel samples */
    uint32_t blocksize, channels, bits_per_sample, sample_rate;
} flacgpu_block;
#define FLACGPU_BLOCK_SILENCE (~(uint64_t)0)
typedef FLAC__StreamDecoderWriteStatus (*flacgpu_block_callback)(const FLAC__StreamDecoder *decoder, const flacgpu_block *blocks,
                                                                 uint32_t nblocks, const void *pcm, uint32_t bytes_per_sample,
                                                                 void *client_data);
FLAC__bool flacgpu_stream_decoder_set_block_callback(FLAC__StreamDecoder *decoder, flacgpu_block_callback callback);

/* FLAC__stream_encoder_process_interleaved for 16-bit interleaved input (an extension beside the libFLAC entry point,
 * stream_encoder.h:1777-1824: same buffering, same return value): saves the caller the widening copy to FLAC__int32. */
FLAC__bool flacgpu_stream_encoder_process_interleaved_i16(FLAC__StreamEncoder *encoder, const int16_t *buffer, uint32_t samples);

/* Extension: a process call encodes nothing until at least `blocks` complete blocks are buffered (default 1: a frame is
 * written as soon as blocksize + 1 samples are there, libFLAC's timing, stream_encoder.h:1777-1824).  Every launch is six
 * kernels and a wait (about 0.17 ms) whatever it carries, so a caller that feeds small pieces and can take its frames in bursts
 * trades latency for throughput with this.  The bytes do not change; FLAC__stream_encoder_finish encodes what is left.  May be
 * set at any time. */
FLAC__bool flacgpu_stream_encoder_set_launch_blocks(FLAC__StreamEncoder *encoder, uint32_t blocks);

/* Host-side frame indexer: parses metadata and frame headers of a complete FLAC stream in host memory and
 * returns frame byte offsets (validated by header CRC-8 and chained by frame CRC-16).  Returns the number
 * of frames, or a negative value on error. */
int64_t flacgpu_index_frames(const uint8_t *stream, uint64_t len, uint64_t *frame_offsets, uint64_t capacity,
                             FLAC__StreamMetadata_StreamInfo *streaminfo, uint64_t *audio_offset);

/* Diagnostic (host only, no GPU): what libFLAC 1.4.3's serial reader makes of a possibly damaged stream when its read
 * callback answers with at most read_size bytes -- the replay the stream decoder runs wherever the GPU pass rejects a frame
 * (csrc/fg_refwalk.h; stream_decoder.c frame_sync_ / read_frame_, bitreader.c's 8 KiB buffer and its
 * FLAC__bitreader_rewind_to_after_last_seen_framesync).  errors: the FLAC__StreamDecoderErrorStatus sequence; frames: pairs
 * {first sample number, block size} of the frames that decode.  Returns the number of error statuses, negative without a
 * readable metadata section. */
int64_t flacgpu_refwalk_probe(const uint8_t *stream, uint64_t len, uint32_t read_size, uint32_t *errors, uint64_t errors_cap,
                              uint64_t *frames, uint64_t frames_cap, uint64_t *nframes);

#ifdef __cplusplus
}
#endif
#endif /* FLACGPU_H */
