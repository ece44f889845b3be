"""Drive the FLAC__stream_decoder_* entry points of libflacgpu.so the way pyFLAC's StreamDecoder does
(pyflac/decoder.py:170-196 init_stream + process_until_end_of_stream; callbacks decoder.py:394-549) and record what
the callbacks see: every delivered frame's header and samples and the error-callback status sequence."""
import ctypes as C
import hashlib

import numpy as np


def decode(data, read_size=8192, md5_checking=False):
    from pyflac_amd import _lib
    L = _lib.lib()
    dec = C.c_void_p(L.FLAC__stream_decoder_new())
    pos = [0]
    frames, blocks, errors = [], [], []

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
        h = fr.contents.header
        blk = np.stack([np.ctypeslib.as_array(bufs[c], shape=(h.blocksize,)).copy() for c in range(h.channels)], axis=1)
        blocks.append(blk)
        frames.append([int(h.number.sample_number), int(h.blocksize),
                       hashlib.sha256(np.ascontiguousarray(blk, np.int32).tobytes()).hexdigest()[:16]])
        return 0

    def _e(d, status, cd):
        errors.append(int(status))

    rcb, wcb, ecb = _lib.DEC_READ_CB(_r), _lib.DEC_WRITE_CB(_w), _lib.DEC_ERROR_CB(_e)
    if md5_checking:
        assert L.FLAC__stream_decoder_set_md5_checking(dec, 1)
    rc = L.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, C.cast(None, _lib.DEC_META_CB), ecb, None)
    assert rc == 0, rc
    ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
    state = L.FLAC__stream_decoder_get_state(dec)
    fin = L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete(dec)
    return {'frames': frames, 'errors': errors, 'state': int(state), 'ok': bool(ok), 'blocks': blocks, 'finish': bool(fin)}
