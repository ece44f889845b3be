"""ctypes harness around the reference's bundled libFLAC 1.4.3 binary.

TEST INFRASTRUCTURE ONLY.  This drives
``/root/reference/pyflac/libraries/linux-x86_64/libFLAC-12.1.0.so`` (the
library pyFLAC links against, ``pyflac/builder/build_args.py:37-51``) through
the same entry points pyFLAC calls (``pyflac/encoder.py:115,132,319`` and
``pyflac/decoder.py:170,196``).  It exists to (a) pin ``oracle/flac_oracle.c``
against the real thing and (b) generate the golden vectors committed under
``tests/golden/`` (``oracle/gen_golden.py``).  The binary lives under
``/root/reference`` and never travels to the GPU box, so nothing in the
product path, ``bench.py`` or the ``-m gpu`` tests may import this module.

Struct layouts follow ``pyflac/builder/decoder.py:146-231`` (FLAC__Frame) and
``pyflac/builder/encoder.py:129-248`` (FLAC__StreamMetadata).
"""
import ctypes as C
import os

import numpy as np

REF_LIB = '/root/reference/pyflac/libraries/linux-x86_64/libFLAC-12.1.0.so'


def available():
    return os.path.exists(REF_LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(REF_LIB)
        _lib.FLAC__stream_encoder_new.restype = C.c_void_p
        _lib.FLAC__stream_decoder_new.restype = C.c_void_p
    return _lib


# ---- FLAC__Frame mirror (pyflac/builder/decoder.py:146-231) -----------------
# (FLAC__Frame and its parts: tests/flac_frame.py, shared with the GPU tests of FLAC__Frame.subframes[])
from tests.flac_frame import (EntropyCodingMethod, Frame, FrameFooter, FrameHeader, PartitionedRice, RiceContents,  # noqa: E402,F401
                              SubConstant, SubFixed, SubLPC, SubVerbatim, Subframe, _subframe_info)


class StreamInfo(C.Structure):
    _fields_ = [('min_blocksize', C.c_uint32), ('max_blocksize', C.c_uint32),
                ('min_framesize', C.c_uint32), ('max_framesize', C.c_uint32),
                ('sample_rate', C.c_uint32), ('channels', C.c_uint32),
                ('bits_per_sample', C.c_uint32), ('total_samples', C.c_uint64),
                ('md5sum', C.c_uint8 * 16)]


class _MetaData(C.Union):
    _fields_ = [('stream_info', StreamInfo), ('pad', C.c_uint8 * 256)]


class StreamMetadata(C.Structure):
    _fields_ = [('type', C.c_int), ('is_last', C.c_int), ('length', C.c_uint32),
                ('data', _MetaData)]


ENC_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.c_size_t,
                           C.c_uint32, C.c_uint32, C.c_void_p)
ENC_SEEK_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p)
ENC_TELL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p)
ENC_META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(StreamMetadata), C.c_void_p)
DEC_READ_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte),
                          C.POINTER(C.c_size_t), C.c_void_p)
DEC_WRITE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Frame),
                           C.POINTER(C.POINTER(C.c_int32)), C.c_void_p)
DEC_ERROR_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


def encode(pcm, sample_rate, bps=None, level=5, blocksize=0, chunk=None,
           seekable=False, extra=None, want_metadata=False):
    """Encode ``pcm`` (int array [frames] or [frames, channels]) in stream mode.

    Returns ``(callbacks, info)`` where ``callbacks`` is the ordered list of
    ``(bytes, samples, current_frame)`` write-callback payloads, exactly what
    pyFLAC's ``_write_callback`` (``pyflac/encoder.py:429-450``) would see.
    ``extra`` is a list of ``(setter_name, value)`` applied after the level.
    """
    L = lib()
    pcm = np.asarray(pcm)
    channels = 1 if pcm.ndim == 1 else pcm.shape[1]
    if bps is None:
        bps = pcm.dtype.itemsize * 8
    data = np.ascontiguousarray(pcm).astype(np.int32).reshape(-1)
    nframes = data.size // channels
    enc = C.c_void_p(L.FLAC__stream_encoder_new())
    out = []
    filebuf = bytearray()
    pos = [0]
    meta = {}

    def _w(e, buf, n, samples, frame, cd):
        b = bytes(C.cast(buf, C.POINTER(C.c_ubyte * n)).contents) if n else b''
        out.append((b, samples, frame))
        if seekable:
            filebuf[pos[0]:pos[0] + n] = b
            pos[0] += n
        return 0

    def _s(e, off, cd):
        pos[0] = off
        return 0

    def _t(e, poff, cd):
        poff[0] = pos[0]
        return 0

    def _m(e, md, cd):
        si = md.contents.data.stream_info
        meta.update(min_blocksize=si.min_blocksize, max_blocksize=si.max_blocksize,
                    min_framesize=si.min_framesize, max_framesize=si.max_framesize,
                    sample_rate=si.sample_rate, channels=si.channels,
                    bits_per_sample=si.bits_per_sample, total_samples=si.total_samples,
                    md5sum=bytes(si.md5sum))

    wcb = ENC_WRITE_CB(_w)
    scb = ENC_SEEK_CB(_s) if seekable else None
    tcb = ENC_TELL_CB(_t) if seekable else None
    mcb = ENC_META_CB(_m) if want_metadata else None
    L.FLAC__stream_encoder_set_channels(enc, channels)
    L.FLAC__stream_encoder_set_bits_per_sample(enc, bps)
    L.FLAC__stream_encoder_set_sample_rate(enc, sample_rate)
    L.FLAC__stream_encoder_set_compression_level(enc, level)
    L.FLAC__stream_encoder_set_blocksize(enc, blocksize)
    for name, val in (extra or []):
        fn = getattr(L, 'FLAC__stream_encoder_' + name)
        fn(enc, val)
    rc = L.FLAC__stream_encoder_init_stream(
        enc, wcb,
        C.cast(scb, C.c_void_p) if scb else None,
        C.cast(tcb, C.c_void_p) if tcb else None,
        C.cast(mcb, C.c_void_p) if mcb else None, None)
    info = {'init_status': rc}
    if rc != 0:
        L.FLAC__stream_encoder_delete(enc)
        return out, info
    info['blocksize'] = L.FLAC__stream_encoder_get_blocksize(enc)
    ok = True
    step = chunk or max(nframes, 1)
    i = 0
    while i < nframes and ok:
        n = min(step, nframes - i)
        seg = data[i * channels:(i + n) * channels]
        ok = bool(L.FLAC__stream_encoder_process_interleaved(
            enc, seg.ctypes.data_as(C.POINTER(C.c_int32)), n))
        i += n
    info['process_ok'] = ok
    info['state_after_process'] = L.FLAC__stream_encoder_get_state(enc)
    info['finish_ok'] = bool(L.FLAC__stream_encoder_finish(enc))
    L.FLAC__stream_encoder_delete(enc)
    info['metadata'] = meta
    if seekable:
        info['file'] = bytes(filebuf)
    return out, info


def decode(data, read_size=8192, want_frames=True, md5_checking=False):
    """Decode a FLAC byte string.  Returns ``(pcm[frames, ch] int32, frames, errors)``."""
    L = lib()
    dec = C.c_void_p(L.FLAC__stream_decoder_new())
    pos = [0]
    blocks, frames, errors, events = [], [], [], []

    def _r(d, buf, pn, cd):
        n = min(pn[0], len(data) - pos[0], read_size)
        if n <= 0:
            pn[0] = 0
            return 1
        C.memmove(buf, data[pos[0]:pos[0] + n], n)
        pos[0] += n
        pn[0] = n
        return 0

    def _w(d, fr, bufs, cd):
        f = fr.contents
        h = f.header
        ch = [np.ctypeslib.as_array(bufs[c], shape=(h.blocksize,)).copy()
              for c in range(h.channels)]
        blocks.append(np.stack(ch, axis=1))
        events.append('f%d' % h.number.sample_number)
        if want_frames:
            frames.append({
                'blocksize': h.blocksize, 'sample_rate': h.sample_rate,
                'channels': h.channels, 'channel_assignment': h.channel_assignment,
                'bits_per_sample': h.bits_per_sample, 'number_type': h.number_type,
                'sample_number': h.number.sample_number, 'crc8': h.crc,
                'crc16': f.footer.crc,
                'subframes': [_subframe_info(f.subframes[c], h.blocksize)
                              for c in range(h.channels)]})
        return 0

    def _e(d, status, cd):
        errors.append(status)
        events.append('e%d' % status)

    rcb, wcb, ecb = DEC_READ_CB(_r), DEC_WRITE_CB(_w), DEC_ERROR_CB(_e)
    if md5_checking:
        assert L.FLAC__stream_decoder_set_md5_checking(dec, 1)
    rc = L.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb,
                                            None, ecb, None)
    assert rc == 0, rc
    ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
    state = L.FLAC__stream_decoder_get_state(dec)
    fin = L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete(dec)
    pcm = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 1), np.int32)
    return pcm, frames, {'errors': errors, 'ok': bool(ok), 'state': state, 'finish': bool(fin), 'events': events}
