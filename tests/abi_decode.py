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
    frames, blocks, errors, events = [], [], [], []

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
        events.append('f%d' % h.number.sample_number)
        return 0

    def _e(d, status, cd):
        errors.append(int(status))
        events.append('e%d' % int(status))

    rcb, wcb, ecb = _lib.DEC_READ_CB(_r), _lib.DEC_WRITE_CB(_w), _lib.DEC_ERROR_CB(_e)
    if md5_checking:
        assert L.FLAC__stream_decoder_set_md5_checking(dec, 1)
    rc = L.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, C.cast(None, _lib.DEC_META_CB), ecb, None)
    assert rc == 0, rc
    ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
    state = L.FLAC__stream_decoder_get_state(dec)
    fin = L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete(dec)
    return {'frames': frames, 'errors': errors, 'state': int(state), 'ok': bool(ok), 'blocks': blocks, 'finish': bool(fin), 'events': events}


def decode_script(data, script, read_size=8192, blocks=False):
    """Run a list of steps on one decoder: ('until',) process_until_end_of_stream; ('single',) process_single until the state is
    END_OF_STREAM; ('flush',) FLAC__stream_decoder_flush; ('source', offset) move the client's read position.  Returns the
    (sample number, block size) pairs delivered and the error statuses, per step."""
    from pyflac_amd import _lib
    L = _lib.lib()
    dec = C.c_void_p(L.FLAC__stream_decoder_new())
    pos = [0]
    frames, errors = [], []
    dt = np.dtype([('sample_number', '<u8'), ('offset', '<u8'), ('blocksize', '<u4'), ('channels', '<u4'), ('bits_per_sample', '<u4'),
                   ('sample_rate', '<u4')])

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
        frames.append((int(h.number.sample_number), int(h.blocksize)))
        return 0

    def _b(d, blks, n, pcm, nbytes, cd):
        for r in np.frombuffer((C.c_uint8 * (32 * n)).from_address(blks), dtype=dt):
            frames.append((int(r['sample_number']), int(r['blocksize'])))
        return 0

    def _e(d, status, cd):
        errors.append(int(status))

    rcb, wcb, ecb, bcb = _lib.DEC_READ_CB(_r), _lib.DEC_WRITE_CB(_w), _lib.DEC_ERROR_CB(_e), _lib.DEC_BLOCK_CB(_b)
    if blocks:
        assert L.flacgpu_stream_decoder_set_block_callback(dec, bcb)
    rc = L.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, C.cast(None, _lib.DEC_META_CB), ecb, None)
    assert rc == 0, rc
    out = []
    for step in script:
        nf, ne = len(frames), len(errors)
        ok = 1
        if step[0] == 'until':
            ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
        elif step[0] == 'single':
            for _ in range(100000):
                if L.FLAC__stream_decoder_get_state(dec) == 4:
                    break
                ok = L.FLAC__stream_decoder_process_single(dec)
                if not ok:
                    break
        elif step[0] == 'flush':
            ok = L.FLAC__stream_decoder_flush(dec)
        elif step[0] == 'source':
            pos[0] = step[1]
        out.append({'ok': bool(ok), 'state': int(L.FLAC__stream_decoder_get_state(dec)), 'frames': frames[nf:], 'errors': errors[ne:]})
    L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete(dec)
    return out


def decode_blocks(data, read_size=8192):
    """The same through libflacgpu's block delivery (include/flacgpu.h, flacgpu_block_callback): what the callbacks see, in the
    form decode() records it."""
    from pyflac_amd import _lib
    L = _lib.lib()
    dec = C.c_void_p(L.FLAC__stream_decoder_new())
    pos = [0]
    frames, errors, events, calls = [], [], [], [0]
    dt = np.dtype([('sample_number', '<u8'), ('offset', '<u8'), ('blocksize', '<u4'), ('channels', '<u4'), ('bits_per_sample', '<u4'),
                   ('sample_rate', '<u4')])

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
        raise AssertionError('the write callback must not be used beside the block callback')

    def _b(d, blocks, n, pcm, nbytes, cd):
        calls[0] += 1
        rec = np.frombuffer((C.c_uint8 * (32 * n)).from_address(blocks), dtype=dt)
        for r in rec:
            ch, bs = int(r['channels']), int(r['blocksize'])
            if int(r['offset']) == 0xFFFFFFFFFFFFFFFF:
                blk = np.zeros((bs, ch), np.int32)
            else:
                ct = C.c_int16 if nbytes == 2 else C.c_int32
                blk = np.frombuffer((ct * (bs * ch)).from_address(pcm + int(r['offset']) * ch * nbytes),
                                    dtype=np.int16 if nbytes == 2 else np.int32).reshape(bs, ch).astype(np.int32)
            frames.append([int(r['sample_number']), bs, hashlib.sha256(np.ascontiguousarray(blk, np.int32).tobytes()).hexdigest()[:16]])
            events.append('f%d' % int(r['sample_number']))
        return 0

    def _e(d, status, cd):
        errors.append(int(status))
        events.append('e%d' % int(status))

    rcb, wcb, ecb, bcb = _lib.DEC_READ_CB(_r), _lib.DEC_WRITE_CB(_w), _lib.DEC_ERROR_CB(_e), _lib.DEC_BLOCK_CB(_b)
    assert L.flacgpu_stream_decoder_set_block_callback(dec, bcb)
    rc = L.FLAC__stream_decoder_init_stream(dec, rcb, None, None, None, None, wcb, C.cast(None, _lib.DEC_META_CB), ecb, None)
    assert rc == 0, rc
    ok = L.FLAC__stream_decoder_process_until_end_of_stream(dec)
    state = L.FLAC__stream_decoder_get_state(dec)
    fin = L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete(dec)
    return {'frames': frames, 'errors': errors, 'state': int(state), 'ok': bool(ok), 'finish': bool(fin), 'events': events, 'calls': calls[0]}


# ---- FLAC__StreamMetadata mirror (pyflac/builder/encoder.py:129-248) for metadata-callback tests -------------------
class _SI(C.Structure):
    _fields_ = [('min_blocksize', C.c_uint32), ('max_blocksize', C.c_uint32), ('min_framesize', C.c_uint32),
                ('max_framesize', C.c_uint32), ('sample_rate', C.c_uint32), ('channels', C.c_uint32),
                ('bits_per_sample', C.c_uint32), ('total_samples', C.c_uint64), ('md5sum', C.c_ubyte * 16)]


class _App(C.Structure):
    _fields_ = [('id', C.c_ubyte * 4), ('data', C.POINTER(C.c_ubyte))]


class _SeekPoint(C.Structure):
    _fields_ = [('sample_number', C.c_uint64), ('stream_offset', C.c_uint64), ('frame_samples', C.c_uint32)]


class _SeekTable(C.Structure):
    _fields_ = [('num_points', C.c_uint32), ('points', C.POINTER(_SeekPoint))]


class _VCEntry(C.Structure):
    _fields_ = [('length', C.c_uint32), ('entry', C.POINTER(C.c_ubyte))]


class _VC(C.Structure):
    _fields_ = [('vendor_string', _VCEntry), ('num_comments', C.c_uint32), ('comments', C.POINTER(_VCEntry))]


class _CueIndex(C.Structure):
    _fields_ = [('offset', C.c_uint64), ('number', C.c_ubyte)]


class _CueTrack(C.Structure):
    # the two 1-bit fields share byte 22 (bit 0 type, bit 1 pre_emphasis) under the Itanium C ABI; ctypes would start a
    # fresh 32-bit unit for them, so the byte is mirrored explicitly
    _fields_ = [('offset', C.c_uint64), ('number', C.c_ubyte), ('isrc', C.c_char * 13), ('flags', C.c_ubyte),
                ('num_indices', C.c_ubyte), ('indices', C.POINTER(_CueIndex))]


class _Cue(C.Structure):
    _fields_ = [('media_catalog_number', C.c_char * 129), ('lead_in', C.c_uint64), ('is_cd', C.c_int),
                ('num_tracks', C.c_uint32), ('tracks', C.POINTER(_CueTrack))]


class _Pic(C.Structure):
    _fields_ = [('type', C.c_int), ('mime_type', C.c_char_p), ('description', C.c_char_p), ('width', C.c_uint32),
                ('height', C.c_uint32), ('depth', C.c_uint32), ('colors', C.c_uint32), ('data_length', C.c_uint32),
                ('data', C.POINTER(C.c_ubyte))]


class _Unknown(C.Structure):
    _fields_ = [('data', C.POINTER(C.c_ubyte))]


class _MData(C.Union):
    _fields_ = [('stream_info', _SI), ('application', _App), ('seek_table', _SeekTable), ('vorbis_comment', _VC),
                ('cue_sheet', _Cue), ('picture', _Pic), ('unknown', _Unknown)]


class Metadata(C.Structure):
    _fields_ = [('type', C.c_int), ('is_last', C.c_int), ('length', C.c_uint32), ('data', _MData)]


META_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Metadata), C.c_void_p)


def _blob(p, n):
    return bytes(p[:n]).hex() if n and p else ''


def metadata_to_dict(m):
    d = {'type': int(m.type), 'is_last': int(m.is_last), 'length': int(m.length)}
    t = m.type
    if t == 0:
        s = m.data.stream_info
        d.update(min_blocksize=s.min_blocksize, max_blocksize=s.max_blocksize, min_framesize=s.min_framesize,
                 max_framesize=s.max_framesize, sample_rate=s.sample_rate, channels=s.channels,
                 bits_per_sample=s.bits_per_sample, total_samples=int(s.total_samples), md5=bytes(s.md5sum).hex())
    elif t == 2:
        d.update(id=bytes(m.data.application.id).hex(), data=_blob(m.data.application.data, m.length - 4))
    elif t == 3:
        st = m.data.seek_table
        d['points'] = [[int(st.points[i].sample_number), int(st.points[i].stream_offset), int(st.points[i].frame_samples)]
                       for i in range(st.num_points)]
    elif t == 4:
        vc = m.data.vorbis_comment
        d['vendor'] = _blob(vc.vendor_string.entry, vc.vendor_string.length)
        d['comments'] = [_blob(vc.comments[i].entry, vc.comments[i].length) for i in range(vc.num_comments)]
    elif t == 5:
        cs = m.data.cue_sheet
        d.update(mcn=cs.media_catalog_number.hex(), lead_in=int(cs.lead_in), is_cd=int(cs.is_cd))
        d['tracks'] = [{'offset': int(cs.tracks[i].offset), 'number': int(cs.tracks[i].number), 'isrc': cs.tracks[i].isrc.hex(),
                        'type': int(cs.tracks[i].flags & 1), 'pre_emphasis': int((cs.tracks[i].flags >> 1) & 1),
                        'indices': [[int(cs.tracks[i].indices[k].offset), int(cs.tracks[i].indices[k].number)]
                                    for k in range(cs.tracks[i].num_indices)]} for i in range(cs.num_tracks)]
    elif t == 6:
        pc = m.data.picture
        d.update(ptype=int(pc.type), mime=(pc.mime_type or b'').hex(), description=(pc.description or b'').hex(), width=pc.width,
                 height=pc.height, depth=pc.depth, colors=pc.colors, data=_blob(pc.data, pc.data_length))
    elif t != 1:
        d['data'] = _blob(m.data.unknown.data, m.length)
    return d


def read_metadata(L, data, setup):
    """Blocks the metadata callback of library `L` (ours or the reference binary) receives for stream `data`.
    `setup`: list of ('respond'|'ignore', type) / ('respond_all',) / ('ignore_all',) / ('respond_app'|'ignore_app', b'abcd')."""
    L = C.CDLL(L._name)       # a private wrapper object: the signatures set below must not leak into the shared one
    L.FLAC__stream_decoder_new.restype = C.c_void_p
    dec = C.c_void_p(L.FLAC__stream_decoder_new())
    pos = [0]
    blocks = []
    rcb_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_ubyte), C.POINTER(C.c_size_t), C.c_void_p)
    wcb_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    ecb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

    def _r(d, buf, pn, cd):
        n = min(pn[0], len(data) - pos[0])
        if n <= 0:
            pn[0] = 0
            return 1
        C.memmove(buf, data[pos[0]:pos[0] + n], n)
        pos[0] += n
        pn[0] = n
        return 0

    rcb, wcb, ecb = rcb_t(_r), wcb_t(lambda *a: 0), ecb_t(lambda *a: None)
    mcb = META_CB(lambda d, m, cd: blocks.append(metadata_to_dict(m.contents)))
    for op in setup:
        if op[0] in ('respond', 'ignore'):
            f = getattr(L, 'FLAC__stream_decoder_set_metadata_' + op[0])
            f.argtypes = [C.c_void_p, C.c_int]
            assert f(dec, op[1])
        elif op[0] in ('respond_all', 'ignore_all'):
            f = getattr(L, 'FLAC__stream_decoder_set_metadata_' + op[0])
            f.argtypes = [C.c_void_p]
            assert f(dec)
        else:
            f = getattr(L, 'FLAC__stream_decoder_set_metadata_' + ('respond' if op[0] == 'respond_app' else 'ignore') + '_application')
            f.argtypes = [C.c_void_p, C.c_char_p]
            assert f(dec, op[1])
    init = L.FLAC__stream_decoder_init_stream
    init.argtypes = [C.c_void_p, rcb_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, wcb_t, META_CB, ecb_t, C.c_void_p]
    assert init(dec, rcb, None, None, None, None, wcb, mcb, ecb, None) == 0
    L.FLAC__stream_decoder_process_until_end_of_metadata.argtypes = [C.c_void_p]
    ok = L.FLAC__stream_decoder_process_until_end_of_metadata(dec)
    L.FLAC__stream_decoder_finish.argtypes = [C.c_void_p]
    L.FLAC__stream_decoder_finish(dec)
    L.FLAC__stream_decoder_delete.argtypes = [C.c_void_p]
    L.FLAC__stream_decoder_delete(dec)
    return {'ok': bool(ok), 'blocks': blocks}
