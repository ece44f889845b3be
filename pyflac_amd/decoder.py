"""pyFLAC-compatible decoder classes over libflacgpu (HIP).

Mirror of ``pyflac/decoder.py`` (reference): `StreamDecoder` (background thread + deque + Event protocol,
pyflac/decoder.py:116-241), `FileDecoder` (:244-313), `OneShotDecoder` (:316-391) and the callback
trampolines (:394-549), bound with ctypes instead of cffi.
"""
import ctypes as C
from collections import deque
from enum import Enum
import logging
from pathlib import Path
import tempfile
import threading
import time
from typing import Callable, Tuple

import numpy as np

from . import _lib
from . import wav

_L = _lib.lib()
# flacgpu_block (include/flacgpu.h)
_BLOCK_DTYPE = np.dtype([('sample_number', '<u8'), ('offset', '<u8'), ('blocksize', '<u4'), ('channels', '<u4'),
                         ('bits_per_sample', '<u4'), ('sample_rate', '<u4')])
_BLOCK_SILENCE = 0xFFFFFFFFFFFFFFFF


class DecoderState(Enum):
    """The decoder state as a Python enumeration (pyflac/decoder.py:30-46)."""
    SEARCH_FOR_METADATA = 0
    READ_METADATA = 1
    SEARCH_FOR_FRAME_SYNC = 2
    READ_FRAME = 3
    END_OF_STREAM = 4
    OGG_ERROR = 5
    SEEK_ERROR = 6
    ABORTED = 7
    MEMORY_ALLOCATION_ERROR = 8
    UNINITIALIZED = 9

    def __str__(self):
        return _lib.string_table('FLAC__StreamDecoderStateString', 10)[self.value].decode()


class DecoderInitException(Exception):
    """Raised if initialisation fails for a `StreamDecoder` or a `FileDecoder`."""
    def __init__(self, code):
        self.code = code

    def __str__(self):
        return _lib.string_table('FLAC__StreamDecoderInitStatusString', 6)[self.code].decode()


class DecoderProcessException(Exception):
    """Raised if an error occurs during the processing of audio data."""
    pass


class _Decoder:
    """Generic decoder: handles interaction with the C library (pyflac/decoder.py:73-113)."""

    def __init__(self):
        self._error = None
        self._decoder = _L.FLAC__stream_decoder_new()
        self.logger = logging.getLogger(__name__)
        self._make_trampolines()

    def __del__(self):
        dec = getattr(self, '_decoder', None)
        if dec:
            _L.FLAC__stream_decoder_delete(dec)
            self._decoder = None

    def finish(self):
        _L.FLAC__stream_decoder_finish(self._decoder)

    @property
    def state(self) -> DecoderState:
        return DecoderState(_L.FLAC__stream_decoder_get_state(self._decoder))

    def process(self):
        raise NotImplementedError

    # -- trampolines (pyflac/decoder.py:394-549)
    def _make_trampolines(self):
        def _read(_dec, byte_buffer, num_bytes, _client):
            try:
                if getattr(self, '_eof_when_empty', False) and not self._buffer:
                    num_bytes[0] = 0
                    return 1   # END_OF_STREAM: a one-shot buffer has nothing more to come
                maximum_bytes = int(num_bytes[0])
                while True:
                    # wait until there is something in the buffer, an error occurred, or finish() was called
                    self._event.wait()
                    if self._error:
                        return 2   # ABORT
                    self._lock.acquire()
                    if self._buffer or self._done:
                        break
                    # woken with nothing to hand over (the flag outlived the data it announced): wait again
                    self._event.clear()
                    self._lock.release()
                if self._done and not self._buffer:
                    self._lock.release()
                    num_bytes[0] = 0
                    return 1   # END_OF_STREAM
                taken = []
                try:
                    # whole queued items while they fit, then the head of the next one; items are memoryviews, so taking
                    # a head is O(1) (slicing a bytes object here would copy the remainder on every call)
                    while self._buffer and maximum_bytes > 0:
                        head = self._buffer[0]
                        take = min(len(head), maximum_bytes)
                        taken.append(head[:take])
                        maximum_bytes -= take
                        if take == len(head):
                            self._buffer.popleft()
                        else:
                            self._buffer[0] = head[take:]
                    if len(self._buffer) == 0 and not self._done:
                        self._event.clear()
                finally:
                    self._lock.release()
                # straight from each item into the decoder's buffer (outside the lock)
                dst = C.cast(byte_buffer, C.c_void_p).value
                actual_bytes = 0
                for part in taken:
                    C.memmove(dst + actual_bytes, np.frombuffer(part, np.uint8).ctypes.data, len(part))
                    actual_bytes += len(part)
                num_bytes[0] = actual_bytes
                return 0   # CONTINUE
            except Exception:   # noqa: BLE001  (def_extern(error=ABORT) in the reference)
                return 2

        def _write(_dec, frame, buffer, _client):
            try:
                h = frame.contents.header
                if h.bits_per_sample not in (16, 32) and not getattr(self, '_allow_any_bps', False):
                    raise ValueError('Only int16/int32 data type is supported')
                n, nch = int(h.blocksize), int(h.channels)
                base = C.addressof(buffer[0].contents)
                if all(C.addressof(buffer[ch].contents) == base + 4 * n * ch for ch in range(1, nch)):
                    # libflacgpu hands the channels over as one [channels][blocksize] plane: one view, one transposed copy
                    planes = np.ctypeslib.as_array(C.cast(buffer[0], C.POINTER(C.c_int32 * (n * nch))).contents).reshape(nch, n)
                    output = planes.T.astype(np.int16 if h.bits_per_sample == 16 else np.int32)
                else:
                    channels = []
                    for ch in range(0, nch):
                        npbuffer = np.ctypeslib.as_array(buffer[ch], shape=(n,))
                        channels.append(npbuffer.astype(np.int16) if h.bits_per_sample == 16 else npbuffer.copy())
                    output = np.column_stack(channels)
                self.write_callback(output, int(h.sample_rate), int(h.channels), int(h.blocksize))
                return 0   # CONTINUE
            except Exception:   # noqa: BLE001  (def_extern(error=ABORT) in the reference)
                return 1

        def _blocks(_dec, blocks, nblocks, pcm, bytes_per_sample, _client):
            # libflacgpu's block delivery (include/flacgpu.h, flacgpu_block_callback): a whole round of decoded frames at once,
            # interleaved and already int16 for streams of up to 16 bits.  One copy into an array of ours, then a view of it
            # per frame for the write callback -- what _write produces, without the per-frame crossing through ctypes.
            try:
                rec = np.frombuffer((C.c_uint8 * (32 * nblocks)).from_address(blocks), dtype=_BLOCK_DTYPE)
                real = rec['offset'] != _BLOCK_SILENCE
                # the round's layout follows its decoded frames (a silence block repeats the header of the frame before it)
                nch = int(rec['channels'][int(np.argmax(real))]) if real.any() else int(rec['channels'][0])
                bps = rec['bits_per_sample']
                good = rec['channels'] == nch
                if not getattr(self, '_allow_any_bps', False):
                    good &= (bps == 16) | (bps == 32)
                # a frame this class cannot hand out ends the stream where the per-frame path would end it: the frames in front
                # of it are delivered first, then the decoder is told to abort
                count = int(nblocks) if good.all() else int(np.argmin(good))
                if count:
                    rec = rec[:count]
                    real = real[:count]
                    offs, sizes, rates = rec['offset'].tolist(), rec['blocksize'].tolist(), rec['sample_rate'].tolist()
                    # one copy of the samples this call delivers -- [first, last) of the round's buffer, not its whole prefix
                    lo = int(rec['offset'][real].min()) if real.any() else 0
                    hi = int((rec['offset'][real] + rec['blocksize'][real]).max()) if real.any() else 0
                    ctype = C.c_int16 if bytes_per_sample == 2 else C.c_int32
                    audio = np.frombuffer((ctype * ((hi - lo) * nch)).from_address(pcm + lo * nch * bytes_per_sample),
                                          dtype=np.int16 if bytes_per_sample == 2 else np.int32)
                    audio = audio.reshape(hi - lo, nch).copy()
                    if bytes_per_sample == 2 and not np.all(rec['bits_per_sample'] == 16):
                        audio = audio.astype(np.int32)            # (the per-frame path hands out int16 for 16-bit frames only)
                    cb = self.write_callback
                    for o, n, sr in zip(offs, sizes, rates):
                        cb(audio[o - lo:o - lo + n] if o != _BLOCK_SILENCE else np.zeros((n, nch), audio.dtype), sr, nch, n)
                if count != int(nblocks):
                    raise ValueError('Only int16/int32 data type is supported (or the channel count changes inside a round)')
                return 0   # CONTINUE
            except Exception:   # noqa: BLE001
                return 1

        def _error(_dec, status, _client):
            message = _lib.string_table('FLAC__StreamDecoderErrorStatusString', 5)[status].decode()
            self.logger.error(f'Error in libFLAC decoder: {message}')
            self._error = message
            ev = getattr(self, '_event', None)
            if ev is not None:
                ev.set()

        self._c_read = _lib.DEC_READ_CB(_read)
        self._c_write = _lib.DEC_WRITE_CB(_write)
        self._c_error = _lib.DEC_ERROR_CB(_error)
        self._c_blocks = _lib.DEC_BLOCK_CB(_blocks)
        self._c_meta_null = C.cast(None, _lib.DEC_META_CB)

    def _init_stream(self):
        _L.flacgpu_stream_decoder_set_block_callback(self._decoder, self._c_blocks)
        rc = _L.FLAC__stream_decoder_init_stream(self._decoder, self._c_read, None, None, None, None, self._c_write,
                                                 self._c_meta_null, self._c_error, None)
        if rc != 0:
            raise DecoderInitException(rc)


class StreamDecoder(_Decoder):
    """Converts a stream of FLAC bytes back to raw audio (pyflac/decoder.py:116-241).

    Data goes in through `process` (non-blocking); blocks come back through
    ``write_callback(audio: ndarray[blocksize, channels], sample_rate, num_channels, num_samples)`` on a
    background thread.  `finish` must be called at the end.
    """

    def __init__(self, write_callback: Callable[[np.ndarray, int, int, int], None]):
        super().__init__()
        self._done = False
        self._buffer = deque()
        self._event = threading.Event()
        self._lock = threading.Lock()
        self.write_callback = write_callback
        self._init_stream()
        self._thread = threading.Thread(target=self._process)
        self._thread.daemon = True
        self._thread.start()

    def _process(self):
        if not _L.FLAC__stream_decoder_process_until_end_of_stream(self._decoder):
            self._error = 'A fatal read, write, or memory allocation error occurred'

    def process(self, data: bytes):
        """Hand some FLAC bytes to the decoder (non-blocking)."""
        view = memoryview(data).cast('B')
        if len(view) == 0:
            return
        # the flag is raised under the lock: the reader clears it under the same lock only when the deque is empty
        self._lock.acquire()
        self._buffer.append(view)
        self._event.set()
        self._lock.release()

    def finish(self):
        """Drain the buffer, stop the thread, reset the decoder.

        Raises:
            DecoderProcessException: if any fatal read, write, or memory allocation error occurred.
        """
        while self._thread.is_alive() and self._error is None and len(self._buffer) > 0:
            time.sleep(0.01)
        with self._lock:   # under the lock, so that the reader cannot clear the flag after it was raised for good
            self._done = True
            self._event.set()
        self._thread.join()
        super().finish()
        if self._error:
            raise DecoderProcessException(self._error)


class FileDecoder(_Decoder):
    """Reads a FLAC file and writes a WAV file (pyflac/decoder.py:244-313)."""

    def __init__(self, input_file: Path, output_file: Path = None):
        super().__init__()
        self.__output = None
        self.__bits = None
        self.write_callback = self._write_callback
        if output_file:
            self.__output_file = output_file
        else:
            self.__tmp = tempfile.NamedTemporaryFile(suffix='.wav')
            self.__output_file = Path(self.__tmp.name)
        _L.flacgpu_stream_decoder_set_block_callback(self._decoder, self._c_blocks)
        rc = _L.FLAC__stream_decoder_init_file(self._decoder, str(input_file).encode('utf-8'), self._c_write,
                                               self._c_meta_null, self._c_error, None)
        if rc != 0:
            raise DecoderInitException(rc)

    def process(self) -> Tuple[np.ndarray, int]:
        """Decode the file.  Returns (audio ndarray[frames, channels], sample_rate).

        Raises:
            DecoderProcessException: if any fatal read, write, or memory allocation error occurred.
        """
        result = _L.FLAC__stream_decoder_process_until_end_of_stream(self._decoder)
        if self.state != DecoderState.END_OF_STREAM and not result:
            raise DecoderProcessException(str(self.state))
        self.finish()
        if self.__output:
            self.__output.close()
            data, info = wav.read(str(self.__output_file))
            # the reference returns soundfile.read(..., always_2d=True): float64 in [-1, 1), the integer samples divided by
            # 2^(bits - 1) (libsndfile's normalisation)
            return data.astype(np.float64) / float(1 << (info.bits - 1)), info.samplerate

    def _write_callback(self, data: np.ndarray, sample_rate: int, num_channels: int, num_samples: int):
        if self.__output is None:
            self.__output = wav.WavWriter(self.__output_file, sample_rate, num_channels, data.dtype.itemsize * 8)
        self.__output.write(data)


class OneShotDecoder(_Decoder):
    """Decodes one buffer of FLAC bytes, blocking, no thread (pyflac/decoder.py:316-391)."""

    def __init__(self, write_callback: Callable[[np.ndarray, int, int, int], None], buffer: bytes):
        super().__init__()
        self._done = False
        self._eof_when_empty = True
        self._buffer = deque()
        if len(buffer):
            self._buffer.append(memoryview(buffer).cast('B'))
        self._event = threading.Event()
        self._event.set()
        self._lock = threading.Lock()
        self.write_callback = write_callback
        self._init_stream()
        while len(self._buffer) > 0:
            if not _L.FLAC__stream_decoder_process_single(self._decoder):
                break       # (aborted: nothing will read the rest)
        self._done = True
        self._event.set()
        # The reference stops here (pyflac/decoder.py:387-391), which drops the frames libFLAC still holds in its
        # read buffer.  This library reads in larger units, so the remainder is drained explicitly: every frame of
        # the buffer is delivered.
        _L.FLAC__stream_decoder_process_until_end_of_stream(self._decoder)
        super().finish()
