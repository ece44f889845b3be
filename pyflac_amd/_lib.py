"""ctypes loader for libflacgpu.so (the C-ABI boundary, include/flacgpu.h).

The reference binds libFLAC through cffi API-mode modules generated from
``pyflac/builder/encoder.py`` / ``pyflac/builder/decoder.py``; cffi is not available here, so the same
entry points are bound with ctypes.  There is no fallback: if the HIP library is missing this module
raises on import of the product classes.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libflacgpu.so')
# The test-hooks build (csrc/Makefile: the same kernels, the host files with -DFG_TESTHOOKS): the only library that reads the kernel
# selectors and test hooks of csrc/fg_types.h fg_sel().  Tests get it with testhooks_lib() / batch.Context(testhooks=True); a child
# process that sets PYFLAC_AMD_TESTHOOKS=1 gets it in place of the release library (for tests of the drop-in classes).
TESTHOOKS_PATH = os.path.join(_HERE, 'libflacgpu_testhooks.so')
if os.environ.get('PYFLAC_AMD_TESTHOOKS') == '1':
    LIB_PATH = TESTHOOKS_PATH
# A/B of two builds on one GPU box (tools/ab*.sh, tools/kstat_var.sh): another build of the same library, honoured only together with
# the explicit opt-in FLACGPU_ALLOW_LIBRARY_OVERRIDE=1; what is loaded must export the build identification and says so on stderr.
_OVERRIDE = os.environ.get('FLACGPU_LIBRARY') if os.environ.get('FLACGPU_ALLOW_LIBRARY_OVERRIDE') == '1' else None
if _OVERRIDE:
    LIB_PATH = _OVERRIDE


class StreamInfo(C.Structure):
    _fields_ = [('min_blocksize', C.c_uint32), ('max_blocksize', C.c_uint32),
                ('min_framesize', C.c_uint32), ('max_framesize', C.c_uint32),
                ('sample_rate', C.c_uint32), ('channels', C.c_uint32),
                ('bits_per_sample', C.c_uint32), ('total_samples', C.c_uint64),
                ('md5sum', C.c_uint8 * 16)]


class _MetaData(C.Union):
    _fields_ = [('stream_info', StreamInfo), ('reserve_', C.c_uint8 * 160)]


class StreamMetadata(C.Structure):
    _fields_ = [('type', C.c_int), ('is_last', C.c_int), ('length', C.c_uint32), ('data', _MetaData)]


class _Number(C.Union):
    _fields_ = [('frame_number', C.c_uint32), ('sample_number', C.c_uint64)]


class FrameHeader(C.Structure):
    _fields_ = [('blocksize', C.c_uint32), ('sample_rate', C.c_uint32), ('channels', C.c_uint32),
                ('channel_assignment', C.c_int), ('bits_per_sample', C.c_uint32),
                ('number_type', C.c_int), ('number', _Number), ('crc', C.c_uint8)]


class Frame(C.Structure):
    # only the header is read on the Python side (pyflac/decoder.py:500-524)
    _fields_ = [('header', FrameHeader)]


class Settings(C.Structure):
    _fields_ = [('channels', C.c_uint32), ('bits_per_sample', C.c_uint32), ('sample_rate', C.c_uint32),
                ('blocksize', C.c_uint32), ('do_mid_side', C.c_uint32), ('loose_mid_side', C.c_uint32),
                ('max_lpc_order', C.c_uint32), ('qlp_coeff_precision', C.c_uint32),
                ('min_partition_order', C.c_uint32), ('max_partition_order', C.c_uint32),
                ('apod_parts', C.c_uint32), ('streamable_subset', C.c_uint32), ('limit_min_bitrate', C.c_uint32)]


class StreamDesc(C.Structure):
    _fields_ = [('pcm_offset', C.c_uint64), ('nsamples', C.c_uint64), ('first_frame', C.c_uint32),
                ('prev_channel_assignment', C.c_uint32)]


class EncodeStats(C.Structure):
    _fields_ = [('nblocks', C.c_uint32), ('error_flags', C.c_uint32), ('total_bytes', C.c_uint64),
                ('encode_kernel_ms', C.c_float), ('total_gpu_ms', C.c_float),
                ('last_channel_assignment', C.c_uint32), ('redo_blocks', C.c_uint32), ('stage_ms', C.c_float * 8),
                ('log_guard_subframes', C.c_uint32), ('direct_path', C.c_uint32), ('lpc_order_min_margin', C.c_double)]


class StreamRange(C.Structure):
    _fields_ = [('byte_offset', C.c_uint64), ('byte_length', C.c_uint64), ('first_frame_number', C.c_uint64),
                ('nframes', C.c_uint32), ('reserved', C.c_uint32)]


class DecodeStats(C.Structure):
    _fields_ = [('nframes', C.c_uint32), ('error_frames', C.c_uint32), ('total_samples', C.c_uint64),
                ('channels', C.c_uint32), ('bits_per_sample', C.c_uint32), ('sample_rate', C.c_uint32),
                ('max_blocksize', C.c_uint32), ('decode_kernel_ms', C.c_float), ('total_gpu_ms', C.c_float),
                ('index_ms', C.c_float), ('plane_bits', C.c_uint32), ('generic_frames', C.c_uint32),
                ('join_late_workgroups', C.c_uint32)]


ENC_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p)
ENC_SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
ENC_TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
ENC_META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(StreamMetadata), C.c_void_p)
ENC_PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p)
DEC_READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.POINTER(C.c_size_t), C.c_void_p)
DEC_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Frame), C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
DEC_META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(StreamMetadata), C.c_void_p)
DEC_ERROR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)
# flacgpu_block_callback (include/flacgpu.h): decoder, blocks, nblocks, pcm, bytes per sample, client data
DEC_BLOCK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p)

ENCODER_FUNCTIONS = [
    'new', 'delete', 'set_verify', 'set_channels', 'set_bits_per_sample', 'set_sample_rate',
    'set_compression_level', 'set_blocksize', 'set_do_mid_side_stereo', 'set_loose_mid_side_stereo',
    'set_apodization', 'set_max_lpc_order', 'set_qlp_coeff_precision', 'set_do_qlp_coeff_prec_search',
    'set_do_exhaustive_model_search', 'set_min_residual_partition_order', 'set_max_residual_partition_order',
    'set_rice_parameter_search_dist', 'set_total_samples_estimate', 'set_streamable_subset',
    'set_limit_min_bitrate', 'get_state', 'get_resolved_state_string', 'get_verify_decoder_error_stats',
    'get_verify', 'get_streamable_subset', 'get_channels', 'get_bits_per_sample', 'get_sample_rate',
    'get_blocksize', 'get_do_mid_side_stereo', 'get_loose_mid_side_stereo', 'get_max_lpc_order',
    'get_qlp_coeff_precision', 'get_do_qlp_coeff_prec_search', 'get_do_escape_coding',
    'get_do_exhaustive_model_search', 'get_min_residual_partition_order', 'get_max_residual_partition_order',
    'get_rice_parameter_search_dist', 'get_total_samples_estimate', 'get_limit_min_bitrate', 'init_stream',
    'init_ogg_stream', 'init_FILE', 'init_ogg_FILE', 'init_file', 'init_ogg_file', 'finish', 'process',
    'process_interleaved']
DECODER_FUNCTIONS = [
    'new', 'delete', 'set_md5_checking', 'set_metadata_respond', 'set_metadata_respond_application',
    'set_metadata_respond_all', 'set_metadata_ignore', 'set_metadata_ignore_application',
    'set_metadata_ignore_all', 'get_state', 'get_resolved_state_string', 'get_md5_checking',
    'get_total_samples', 'get_channels', 'get_channel_assignment', 'get_bits_per_sample', 'get_sample_rate',
    'get_blocksize', 'get_decode_position', 'init_stream', 'init_ogg_stream', 'init_FILE', 'init_ogg_FILE',
    'init_file', 'init_ogg_file', 'finish', 'flush', 'reset', 'process_single',
    'process_until_end_of_metadata', 'process_until_end_of_stream', 'skip_single_frame', 'seek_absolute']
DATA_SYMBOLS = ['FLAC__StreamEncoderStateString', 'FLAC__StreamEncoderInitStatusString',
                'FLAC__StreamDecoderStateString', 'FLAC__StreamDecoderInitStatusString',
                'FLAC__StreamDecoderErrorStatusString']
EXT_FUNCTIONS = ['flacgpu_settings_from_level', 'flacgpu_device_count', 'flacgpu_ctx_create', 'flacgpu_ctx_destroy',
                 'flacgpu_last_error', 'flacgpu_encode_streams', 'flacgpu_encode_bound', 'flacgpu_set_debug',
                 'flacgpu_copy_debug', 'flacgpu_debug_crc_tables', 'flacgpu_copy_block_results', 'flacgpu_decode_frames', 'flacgpu_decode_frames_dev',
                 'flacgpu_index_frames', 'flacgpu_refwalk_probe', 'flacgpu_stream_encoder_process_interleaved_i16', 'flacgpu_stream_encoder_set_launch_blocks', 'flacgpu_decode_stream_dev', 'flacgpu_decode_streams_dev',
                 'flacgpu_set_stage_timing', 'flacgpu_set_log_guard', 'flacgpu_set_direct', 'flacgpu_window_note', 'flacgpu_selfcheck', 'flacgpu_force_selfcheck_result', 'flacgpu_build_flags', 'flacgpu_build_id', 'flacgpu_kernel_id', 'flacgpu_host_id', 'flacgpu_md5_streams', 'flacgpu_stream_decoder_set_subframe_detail', 'flacgpu_stream_decoder_set_block_callback']

_lib = None
_testhooks = None


def lib():
    """Load libflacgpu.so and declare the signatures used from Python.  Raises if it is missing."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
        if _OVERRIDE:
            import sys
            sys.stderr.write('pyflac_amd: FLACGPU_LIBRARY override: %s (build %s, flags %d)\n' %
                             (LIB_PATH, _lib.flacgpu_build_id().decode(), _lib.flacgpu_build_flags()))
        ignored = ignored_selectors(os.environ, _lib.flacgpu_build_flags())
        if ignored:
            import sys
            sys.stderr.write('pyflac_amd: %s set, but %s is a release build and reads none of them (FLACGPU_DEVICE only): use the '
                             'test-hooks library (PYFLAC_AMD_TESTHOOKS=1) for selectors, a `make TUNING=1` build for experiments\n' %
                             (', '.join(ignored), os.path.basename(LIB_PATH)))
    return _lib


# what the release library itself reads, and what this module reads to pick a library
_READ_BY_EVERY_BUILD = ('FLACGPU_DEVICE', 'FLACGPU_LIBRARY', 'FLACGPU_ALLOW_LIBRARY_OVERRIDE')


def ignored_selectors(environ, build_flags):
    """The FLACGPU_* variables of `environ` that a library with these flacgpu_build_flags() silently ignores: a release build
    (neither bit 0, tuning, nor bit 2, test hooks) reads FLACGPU_DEVICE and nothing else (csrc/fg_types.h fg_sel / fg_tune).  A script
    that sets FLACGPU_GROUPS=1 for the release library measures the default path and believes otherwise."""
    if build_flags & 5:
        return []
    return sorted(k for k in environ if k.startswith('FLACGPU_') and k not in _READ_BY_EVERY_BUILD)


def testhooks_lib():
    """The test-hooks build of the library (kernel selectors and test hooks readable from the environment); for the cross-check tests."""
    global _testhooks
    if _testhooks is None:
        L = _load(TESTHOOKS_PATH)
        if not (L.flacgpu_build_flags() & 4):
            raise ImportError('%s is not a test-hooks build' % TESTHOOKS_PATH)
        _testhooks = L
    return _testhooks


def _load(path):
    if not os.path.exists(path):
        raise ImportError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          '(there is no CPU fallback)' % path)
    # torch carries its own copy of the HIP runtime; whichever copy is loaded first serves the whole process, and a second
    # one finds no device.  torch owns the device memory this package hands to the library, so its runtime goes first.
    import torch  # noqa: F401
    L = C.CDLL(path)
    vp = C.c_void_p
    L.FLAC__stream_encoder_new.restype = vp
    L.FLAC__stream_encoder_delete.argtypes = [vp]
    L.FLAC__stream_encoder_delete.restype = None
    for n in ('verify', 'channels', 'bits_per_sample', 'sample_rate', 'compression_level', 'blocksize',
              'streamable_subset', 'limit_min_bitrate', 'do_md5'):
        f = getattr(L, 'FLAC__stream_encoder_set_' + n)
        f.argtypes = [vp, C.c_uint32]
        f.restype = C.c_int
    for n in ('verify', 'channels', 'bits_per_sample', 'sample_rate', 'blocksize', 'streamable_subset',
              'limit_min_bitrate', 'state', 'max_lpc_order', 'qlp_coeff_precision',
              'max_residual_partition_order', 'do_mid_side_stereo', 'loose_mid_side_stereo'):
        f = getattr(L, 'FLAC__stream_encoder_get_' + n)
        f.argtypes = [vp]
        f.restype = C.c_uint32
    L.FLAC__stream_encoder_init_stream.argtypes = [vp, ENC_WRITE_CB, ENC_SEEK_CB, ENC_TELL_CB, ENC_META_CB, vp]
    L.FLAC__stream_encoder_init_stream.restype = C.c_int
    L.FLAC__stream_encoder_init_file.argtypes = [vp, C.c_char_p, ENC_PROGRESS_CB, vp]
    L.FLAC__stream_encoder_init_file.restype = C.c_int
    L.FLAC__stream_encoder_process_interleaved.argtypes = [vp, vp, C.c_uint32]
    L.FLAC__stream_encoder_process_interleaved.restype = C.c_int
    # (planar input, one pointer per channel: pyflac/builder/encoder.py:321; pyFLAC's classes never call it)
    L.FLAC__stream_encoder_process.argtypes = [vp, C.POINTER(C.POINTER(C.c_int32)), C.c_uint32]
    L.FLAC__stream_encoder_process.restype = C.c_int
    L.FLAC__stream_encoder_finish.argtypes = [vp]
    L.FLAC__stream_encoder_get_verify_decoder_error_stats.argtypes = [vp] + [vp] * 6
    L.FLAC__stream_encoder_get_verify_decoder_error_stats.restype = None
    L.FLAC__stream_encoder_finish.restype = C.c_int
    L.FLAC__stream_decoder_new.restype = vp
    L.FLAC__stream_decoder_delete.argtypes = [vp]
    L.FLAC__stream_decoder_delete.restype = None
    L.FLAC__stream_decoder_get_state.argtypes = [vp]
    L.FLAC__stream_decoder_get_state.restype = C.c_int
    L.FLAC__stream_decoder_init_stream.argtypes = [vp, DEC_READ_CB, vp, vp, vp, vp, DEC_WRITE_CB, DEC_META_CB, DEC_ERROR_CB, vp]
    L.FLAC__stream_decoder_init_stream.restype = C.c_int
    L.flacgpu_stream_decoder_set_block_callback.argtypes = [vp, DEC_BLOCK_CB]
    L.flacgpu_stream_decoder_set_block_callback.restype = C.c_int
    L.FLAC__stream_decoder_init_file.argtypes = [vp, C.c_char_p, DEC_WRITE_CB, DEC_META_CB, DEC_ERROR_CB, vp]
    L.FLAC__stream_decoder_init_file.restype = C.c_int
    for n in ('finish', 'process_single', 'process_until_end_of_stream', 'process_until_end_of_metadata', 'flush', 'reset'):
        f = getattr(L, 'FLAC__stream_decoder_' + n)
        f.argtypes = [vp]
        f.restype = C.c_int
    L.flacgpu_settings_from_level.argtypes = [C.POINTER(Settings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    L.flacgpu_settings_from_level.restype = C.c_int
    L.flacgpu_device_count.restype = C.c_int
    L.flacgpu_ctx_create.argtypes = [C.c_int]
    L.flacgpu_ctx_create.restype = vp
    L.flacgpu_ctx_destroy.argtypes = [vp]
    L.flacgpu_ctx_destroy.restype = None
    L.flacgpu_last_error.restype = C.c_char_p
    L.flacgpu_encode_streams.argtypes = [vp, C.POINTER(Settings), vp, C.c_int, C.POINTER(StreamDesc), C.c_uint32, vp,
                                         C.c_uint64, vp, C.POINTER(EncodeStats)]
    L.flacgpu_encode_streams.restype = C.c_int
    L.flacgpu_encode_bound.argtypes = [C.POINTER(Settings), C.POINTER(StreamDesc), C.c_uint32, C.POINTER(C.c_uint32)]
    L.flacgpu_encode_bound.restype = C.c_uint64
    L.flacgpu_set_debug.argtypes = [vp, C.c_int]
    L.flacgpu_set_debug.restype = None
    L.flacgpu_copy_debug.argtypes = [vp, vp, C.c_uint32, C.c_uint32]
    L.flacgpu_copy_debug.restype = C.c_int
    L.flacgpu_copy_block_results.argtypes = [vp, vp, C.c_uint32]
    L.flacgpu_copy_block_results.restype = C.c_int
    L.flacgpu_decode_frames.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp,
                                        C.POINTER(DecodeStats)]
    L.flacgpu_decode_frames.restype = C.c_int
    L.flacgpu_decode_frames_dev.argtypes = L.flacgpu_decode_frames.argtypes
    L.flacgpu_decode_frames_dev.restype = C.c_int
    L.flacgpu_decode_stream_dev.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp,
                                            C.c_uint64, vp, C.POINTER(DecodeStats)]
    L.flacgpu_decode_stream_dev.restype = C.c_int
    L.flacgpu_decode_streams_dev.argtypes = [vp, vp, C.c_uint64, C.POINTER(StreamRange), C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64,
                                             vp, C.c_uint64, vp, C.POINTER(DecodeStats)]
    L.flacgpu_decode_streams_dev.restype = C.c_int
    L.flacgpu_set_stage_timing.argtypes = [vp, C.c_int]
    L.flacgpu_set_log_guard.argtypes = [vp, C.c_double]
    L.flacgpu_set_log_guard.restype = None
    L.flacgpu_md5_streams.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, C.POINTER(C.c_float)]
    L.flacgpu_md5_streams.restype = C.c_int
    for _n in ('flacgpu_build_id', 'flacgpu_kernel_id', 'flacgpu_host_id'):
        getattr(L, _n).argtypes = []
        getattr(L, _n).restype = C.c_char_p
    L.flacgpu_set_direct.argtypes = [vp, C.c_int]
    L.flacgpu_set_direct.restype = None
    L.flacgpu_window_note.argtypes = [vp]
    L.flacgpu_window_note.restype = C.c_char_p
    L.flacgpu_selfcheck.argtypes = [vp, C.POINTER(C.c_char_p)]
    L.flacgpu_selfcheck.restype = C.c_int
    L.flacgpu_force_selfcheck_result.argtypes = [vp, C.c_int]
    L.flacgpu_force_selfcheck_result.restype = None
    L.flacgpu_set_stage_timing.restype = None
    L.flacgpu_stream_encoder_process_interleaved_i16.argtypes = [vp, vp, C.c_uint32]
    L.flacgpu_stream_encoder_process_interleaved_i16.restype = C.c_int
    L.flacgpu_stream_encoder_set_launch_blocks.argtypes = [vp, C.c_uint32]
    L.flacgpu_stream_encoder_set_launch_blocks.restype = C.c_int
    L.flacgpu_index_frames.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(StreamInfo), C.POINTER(C.c_uint64)]
    L.flacgpu_index_frames.restype = C.c_int64
    L.flacgpu_refwalk_probe.argtypes = [vp, C.c_uint64, C.c_uint32, vp, C.c_uint64, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.flacgpu_refwalk_probe.restype = C.c_int64
    L.flacgpu_build_flags.argtypes = []
    L.flacgpu_build_flags.restype = C.c_uint32
    return L


def string_table(name, n):
    return (C.c_char_p * n).in_dll(lib(), name)


def last_error():
    return (lib().flacgpu_last_error() or b'').decode()
