"""Minimal RIFF/WAVE reader and writer.

pyFLAC's ``FileEncoder``/``FileDecoder`` go through python-soundfile
(``pyflac/encoder.py:372-380``, ``pyflac/decoder.py:302-313``), which is not
available here; this covers what those call sites need: PCM_16 / PCM_32 (and
PCM_24 / PCM_U8 for completeness), plain ``WAVE_FORMAT_PCM`` and
``WAVE_FORMAT_EXTENSIBLE`` (``tests/data/surround.wav``) headers.
"""
import struct

import numpy as np

_KSDATAFORMAT_PCM = bytes.fromhex('0100000000001000800000aa00389b71')


class WavInfo:
    def __init__(self, sample_rate, channels, bits, frames, fmt_tag):
        self.samplerate = sample_rate
        self.channels = channels
        self.bits = bits
        self.frames = frames
        self.format_tag = fmt_tag

    @property
    def subtype(self):
        return {8: 'PCM_U8', 16: 'PCM_16', 24: 'PCM_24', 32: 'PCM_32'}.get(self.bits, 'UNKNOWN')


def _chunks(buf):
    if len(buf) < 12 or buf[:4] != b'RIFF' or buf[8:12] != b'WAVE':
        raise ValueError('not a RIFF/WAVE file')
    pos = 12
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        size = struct.unpack('<I', buf[pos + 4:pos + 8])[0]
        yield cid, buf[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)


def info(path):
    return read(path, header_only=True)[1]


def read(path, header_only=False):
    """Returns ``(ndarray[frames, channels], WavInfo)``; int16 for 16-bit, int32 for 24/32-bit."""
    with open(str(path), 'rb') as f:
        buf = f.read()
    fmt = None
    data = None
    for cid, body in _chunks(buf):
        if cid == b'fmt ':
            tag, ch, rate, _br, _align, bits = struct.unpack('<HHIIHH', body[:16])
            if tag == 0xFFFE and len(body) >= 40:
                tag = struct.unpack('<H', body[24:26])[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b'data':
            data = body
    if fmt is None or data is None:
        raise ValueError('WAV file lacks fmt or data chunk')
    tag, ch, rate, bits = fmt
    if tag != 1:
        raise ValueError('only integer PCM WAV is supported (format tag %d)' % tag)
    bytes_per = bits // 8
    frames = len(data) // (bytes_per * ch)
    wi = WavInfo(rate, ch, bits, frames, tag)
    if header_only:
        return None, wi
    data = data[:frames * bytes_per * ch]
    if bits == 16:
        a = np.frombuffer(data, '<i2').reshape(frames, ch)
    elif bits == 32:
        a = np.frombuffer(data, '<i4').reshape(frames, ch)
    elif bits == 24:
        raw = np.frombuffer(data, np.uint8).reshape(-1, 3).astype(np.int32)
        a = (raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16))
        a = ((a ^ 0x800000) - 0x800000).astype(np.int32).reshape(frames, ch)
    elif bits == 8:
        a = (np.frombuffer(data, np.uint8).astype(np.int16) - 128).reshape(frames, ch)
    else:
        raise ValueError('unsupported bit depth %d' % bits)
    return a.copy(), wi


class WavWriter:
    """Streaming writer: ``write(ndarray[frames, channels])`` then ``close()``."""

    def __init__(self, path, sample_rate, channels, bits):
        self._f = open(str(path), 'wb')
        self._rate, self._ch, self._bits = sample_rate, channels, bits
        self._bytes = 0
        self._f.write(b'\0' * (68 if channels > 2 else 44))

    def write(self, a):
        a = np.asarray(a)
        if self._bits == 16:
            b = a.astype('<i2').tobytes()
        elif self._bits == 32:
            b = a.astype('<i4').tobytes()
        elif self._bits == 24:
            v = a.astype(np.int32).reshape(-1)
            b = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
        elif self._bits == 8:
            b = (a.astype(np.int16) + 128).astype(np.uint8).tobytes()
        else:
            raise ValueError('unsupported bit depth %d' % self._bits)
        self._f.write(b)
        self._bytes += len(b)

    def close(self):
        if self._f is None:
            return
        ch, rate, bits = self._ch, self._rate, self._bits
        align = ch * bits // 8
        if ch > 2:
            fmt = struct.pack('<HHIIHHHHI', 0xFFFE, ch, rate, rate * align, align, bits, 22, bits,
                              (1 << ch) - 1) + _KSDATAFORMAT_PCM
        else:
            fmt = struct.pack('<HHIIHH', 1, ch, rate, rate * align, align, bits)
        hdr = b'RIFF' + struct.pack('<I', 4 + 8 + len(fmt) + 8 + self._bytes) + b'WAVE' + \
            b'fmt ' + struct.pack('<I', len(fmt)) + fmt + b'data' + struct.pack('<I', self._bytes)
        if self._bytes & 1:
            self._f.write(b'\0')
        self._f.seek(0)
        self._f.write(hdr)
        self._f.close()
        self._f = None


def write(path, a, sample_rate, bits=None):
    a = np.asarray(a)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    if bits is None:
        bits = a.dtype.itemsize * 8
    w = WavWriter(path, sample_rate, a.shape[1], bits)
    w.write(a)
    w.close()
