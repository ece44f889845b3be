"""ctypes binding for oracle/libflac_oracle.so (the CPU restatement in flac_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from pyflac_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libflac_oracle.so')

MAX_CH, MAX_ORDER, MAX_VEC, MAX_PARTS = 8, 32, 16, 256


class Config(C.Structure):
    _fields_ = [('channels', C.c_uint32), ('bps', C.c_uint32), ('sample_rate', C.c_uint32),
                ('blocksize', C.c_uint32), ('do_mid_side', C.c_uint32), ('loose_mid_side', C.c_uint32),
                ('max_lpc_order', C.c_uint32), ('qlp_coeff_precision', C.c_uint32),
                ('min_partition_order', C.c_uint32), ('max_partition_order', C.c_uint32),
                ('apod_type', C.c_uint32), ('apod_p', C.c_float), ('apod_parts', C.c_uint32),
                ('streamable_subset', C.c_uint32), ('do_md5', C.c_uint32), ('limit_min_bitrate', C.c_uint32)]


class SubframeInfo(C.Structure):
    _fields_ = [('wasted', C.c_uint32), ('sbps', C.c_uint32), ('fixed_tot', C.c_uint64 * 5),
                ('fixed_guess', C.c_uint32), ('n_vectors', C.c_uint32),
                ('autoc', (C.c_double * (MAX_ORDER + 1)) * MAX_VEC),
                ('lpc_guess', C.c_uint32 * MAX_VEC), ('lpc_bits', C.c_uint32 * MAX_VEC),
                ('fixed_bits', C.c_uint32), ('type', C.c_uint32), ('order', C.c_uint32),
                ('precision', C.c_uint32), ('shift', C.c_int32), ('qlp', C.c_int32 * MAX_ORDER),
                ('rice_method', C.c_uint32), ('porder', C.c_uint32),
                ('rice_params', C.c_uint32 * MAX_PARTS), ('bits', C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [('blocksize', C.c_uint32), ('channel_assignment', C.c_uint32),
                ('n_candidates', C.c_uint32), ('cand', SubframeInfo * MAX_CH),
                ('frame_bytes', C.c_uint32)]


class LooseState(C.Structure):
    _fields_ = [('count', C.c_uint32), ('last_ca', C.c_uint32)]


class DecodeResult(C.Structure):
    _fields_ = [('min_blocksize', C.c_uint32), ('max_blocksize', C.c_uint32),
                ('min_framesize', C.c_uint32), ('max_framesize', C.c_uint32),
                ('sample_rate', C.c_uint32), ('channels', C.c_uint32), ('bps', C.c_uint32),
                ('total_samples', C.c_uint64), ('md5', C.c_uint8 * 16),
                ('decoded_samples', C.c_uint64), ('n_frames', C.c_uint32),
                ('n_errors', C.c_uint32), ('errors', C.c_uint32 * 64)]


_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'flac_oracle.c')):
            build()
        L = C.CDLL(_SO)
        L.flo_encode_frame.restype = C.c_size_t
        L.flo_encode_frame.argtypes = [C.POINTER(Config), C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.POINTER(LooseState), C.c_void_p, C.POINTER(FrameInfo)]
        L.flo_stream_header.restype = C.c_size_t
        L.flo_stream_header.argtypes = [C.POINTER(Config), C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_void_p, C.c_void_p]
        L.flo_encode_stream.restype = C.c_size_t
        L.flo_encode_stream.argtypes = [C.POINTER(Config), C.c_void_p, C.c_uint64, C.c_int, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32)]
        L.flo_decode_stream.restype = C.c_int
        L.flo_decode_stream.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64,
                                        C.POINTER(DecodeResult), C.c_void_p, C.c_uint32]
        L.flo_window.argtypes = [C.POINTER(Config), C.c_uint32, C.c_void_p]
        L.flo_md5_pcm.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.flo_crc16.restype = C.c_uint16
        L.flo_crc16.argtypes = [C.c_void_p, C.c_size_t]
        L.flo_crc8.restype = C.c_uint8
        L.flo_crc8.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def config(level=5, channels=2, bps=16, sample_rate=48000, blocksize=0, subset=True):
    cfg = Config()
    rc = lib().flo_config_from_level(C.byref(cfg), level, channels, bps, sample_rate, blocksize,
                                     1 if subset else 0)
    return cfg, rc


def _as_i32(pcm):
    pcm = np.asarray(pcm)
    ch = 1 if pcm.ndim == 1 else pcm.shape[1]
    return np.ascontiguousarray(pcm).astype(np.int32).reshape(-1, ch), ch


def encode_frame(cfg, pcm_block, frame_number=0, loose=None, want_info=False):
    a, ch = _as_i32(pcm_block)
    n = a.shape[0]
    out = np.zeros(n * ch * 5 + 64, np.uint8)
    info = FrameInfo() if want_info else None
    nb = lib().flo_encode_frame(C.byref(cfg), a.ctypes.data, n, frame_number,
                                C.byref(loose) if loose is not None else None, out.ctypes.data,
                                C.byref(info) if info is not None else None)
    b = out[:nb].tobytes()
    return (b, info) if want_info else b


def stream_header(cfg, minf=0, maxf=0, total=0, md5=None):
    out = np.zeros(128, np.uint8)
    m = (C.c_uint8 * 16)(*md5) if md5 is not None else None
    n = lib().flo_stream_header(C.byref(cfg), minf, maxf, total, m, out.ctypes.data)
    return out[:n].tobytes()


def encode_stream(cfg, pcm, finalize=False):
    """Returns (bytes, frame_sizes)."""
    a, ch = _as_i32(pcm)
    n = a.shape[0]
    nfr = (n + cfg.blocksize - 1) // cfg.blocksize
    cap = 86 + (a.size + cfg.blocksize * ch) * 6 + 64 * (nfr + 2)
    out = np.zeros(cap, np.uint8)
    sizes = np.zeros(nfr + 1, np.uint32)
    nf = C.c_uint32(0)
    nb = lib().flo_encode_stream(C.byref(cfg), a.ctypes.data, n, 1 if finalize else 0, out.ctypes.data,
                                 cap, sizes.ctypes.data, C.byref(nf))
    if nb == 0:
        raise RuntimeError('oracle encode failed')
    return out[:nb].tobytes(), sizes[:nf.value].copy()


def decode_stream(data, want_offsets=False):
    """Returns (pcm[frames, channels] int32, DecodeResult[, frame_offsets])."""
    buf = np.frombuffer(data, np.uint8)
    res = DecodeResult()
    L = lib()
    rc = L.flo_decode_stream(buf.ctypes.data, buf.size, None, 0, C.byref(res), None, 0)
    if rc < 0:
        raise ValueError('oracle decode failed: %d' % rc)
    ns, ch = res.decoded_samples, max(res.channels, 1)
    out = np.zeros((ns, ch), np.int32)
    offs = np.zeros(res.n_frames + 1, np.uint32)
    res2 = DecodeResult()
    L.flo_decode_stream(buf.ctypes.data, buf.size, out.ctypes.data, ns, C.byref(res2),
                        offs.ctypes.data, offs.size)
    if want_offsets:
        return out, res2, offs[:res2.n_frames].copy()
    return out, res2


def window(cfg, n):
    w = np.zeros(n, np.float32)
    lib().flo_window(C.byref(cfg), n, w.ctypes.data)
    return w


def md5_pcm(pcm, bps):
    a, ch = _as_i32(pcm)
    d = (C.c_uint8 * 16)()
    lib().flo_md5_pcm(a.ctypes.data, a.shape[0], ch, bps, d)
    return bytes(d)
