"""Builders of FLAC__StreamMetadata blocks (ctypes mirrors in tests/abi_decode.py) for FLAC__stream_encoder_set_metadata:
the same objects are handed to the reference binary (oracle/gen_golden_setmeta.py, build container) and to libflacgpu
(tests/test_gpu_api.py); what each writes during init_stream is compared byte for byte."""
import ctypes as C

from tests.abi_decode import Metadata, _SeekPoint, _VCEntry, _CueIndex, _CueTrack

_keep = []          # everything the blocks point to stays alive here


def _bytes(b):
    buf = (C.c_ubyte * max(len(b), 1))(*b)
    _keep.append(buf)
    return C.cast(buf, C.POINTER(C.c_ubyte))


def padding(n):
    m = Metadata()
    m.type, m.length = 1, n
    return m


def application(app_id, data):
    m = Metadata()
    m.type, m.length = 2, 4 + len(data)
    m.data.application.id = (C.c_ubyte * 4)(*app_id)
    m.data.application.data = _bytes(data)
    return m


def seektable(points):
    m = Metadata()
    arr = (_SeekPoint * max(len(points), 1))()
    for i, (sn, off, fs) in enumerate(points):
        arr[i].sample_number, arr[i].stream_offset, arr[i].frame_samples = sn, off, fs
    _keep.append(arr)
    m.type, m.length = 3, 18 * len(points)
    m.data.seek_table.num_points = len(points)
    m.data.seek_table.points = C.cast(arr, C.POINTER(_SeekPoint))
    return m


def vorbis_comment(vendor, comments):
    m = Metadata()
    m.type = 4
    vc = m.data.vorbis_comment
    vc.vendor_string.length = len(vendor)
    vc.vendor_string.entry = _bytes(vendor)
    arr = (_VCEntry * max(len(comments), 1))()
    for i, c in enumerate(comments):
        arr[i].length = len(c)
        arr[i].entry = _bytes(c)
    _keep.append(arr)
    vc.num_comments = len(comments)
    vc.comments = C.cast(arr, C.POINTER(_VCEntry))
    m.length = 4 + len(vendor) + 4 + sum(4 + len(c) for c in comments)
    return m


def cuesheet(mcn, lead_in, is_cd, tracks):
    """tracks: [(offset, number, isrc, type, pre_emphasis, [(index offset, index number), ...]), ...]"""
    m = Metadata()
    m.type = 5
    cs = m.data.cue_sheet
    cs.media_catalog_number = mcn
    cs.lead_in, cs.is_cd, cs.num_tracks = lead_in, is_cd, len(tracks)
    tarr = (_CueTrack * max(len(tracks), 1))()
    n = 396
    for i, (off, num, isrc, typ, pre, idx) in enumerate(tracks):
        tarr[i].offset, tarr[i].number, tarr[i].isrc = off, num, isrc
        tarr[i].flags = (1 if typ else 0) | (2 if pre else 0)
        iarr = (_CueIndex * max(len(idx), 1))()
        for k, (io, inum) in enumerate(idx):
            iarr[k].offset, iarr[k].number = io, inum
        _keep.append(iarr)
        tarr[i].num_indices = len(idx)
        tarr[i].indices = C.cast(iarr, C.POINTER(_CueIndex))
        n += 36 + 12 * len(idx)
    _keep.append(tarr)
    cs.tracks = C.cast(tarr, C.POINTER(_CueTrack))
    m.length = n
    return m


def picture(ptype, mime, desc, w, h, depth, colors, data):
    m = Metadata()
    m.type = 6
    p = m.data.picture
    p.type, p.mime_type, p.description = ptype, mime, desc
    p.width, p.height, p.depth, p.colors, p.data_length = w, h, depth, colors, len(data)
    p.data = _bytes(data)
    m.length = 32 + len(mime) + len(desc) + len(data)
    return m


def unknown(mtype, data):
    m = Metadata()
    m.type, m.length = mtype, len(data)
    m.data.unknown.data = _bytes(data)
    return m


def block_array(blocks):
    arr = (C.POINTER(Metadata) * max(len(blocks), 1))()
    for i, b in enumerate(blocks):
        arr[i] = C.pointer(b)
    _keep.append((arr, blocks))
    return arr


# name -> list of blocks (what a tagger would hand to the encoder)
def cases():
    pic = bytes(range(200)) * 3
    return {
        'padding_only': [padding(1000)],
        'vorbis_only': [vorbis_comment(b'someone else 1.0', [b'TITLE=Ode', b'ARTIST=A. Tester'])],
        'vorbis_not_first': [padding(16), application(b'test', b'\x01\x02\x03\x04\x05'), vorbis_comment(b'', [b'ALBUM=x' * 20])],
        'seektable_verbatim': [seektable([(0, 0, 0), (44100, 0, 0), (88200, 0, 0), (0xFFFFFFFFFFFFFFFF, 0, 0)]), padding(8)],
        'picture_and_cue': [picture(3, b'image/png', 'cover é'.encode('utf-8'), 32, 32, 24, 0, pic),
                            cuesheet(b'1234567890123', 88200, 1,
                                     [(0, 1, b'ABCDE1234567', 0, 0, [(0, 1), (588, 2)]), (441000, 170, b'', 0, 0, [])]),
                            padding(0)],
        'unknown_type': [unknown(42, b'opaque bytes'), application(b'abcd', b'')],
        'streaminfo_refused': [unknown(0, b'\x00' * 34)],
        'two_vorbis_refused': [vorbis_comment(b'', []), vorbis_comment(b'', [b'A=b'])],
        'bad_seektable_refused': [seektable([(1000, 0, 0), (500, 0, 0)])],
    }


# name -> list of blocks: CUESHEET / PICTURE blocks libFLAC's encoder refuses (format.c FLAC__format_cuesheet_is_legal /
# FLAC__format_picture_is_legal) next to legal neighbours of each; only the init status is compared (tests/golden/legality_vectors.json)
def legality_cases():
    T = lambda off, num, idx: (off, num, b'', 0, 0, idx)      # noqa: E731
    ok_tracks = [T(0, 1, [(0, 1)]), T(588 * 100, 170, [])]
    return {
        'cue_legal_cd': [cuesheet(b'', 88200, 1, ok_tracks)],
        'cue_legal_not_cd': [cuesheet(b'', 0, 0, [T(0, 1, [(0, 0), (5, 1)]), T(1000, 255, [])])],
        'cue_no_tracks': [cuesheet(b'', 88200, 1, [])],
        'cue_cd_short_lead_in': [cuesheet(b'', 88199, 1, ok_tracks)],
        'cue_cd_lead_in_not_cd_frames': [cuesheet(b'', 88200 + 1, 1, ok_tracks)],
        'cue_cd_last_track_not_170': [cuesheet(b'', 88200, 1, [T(0, 1, [(0, 1)]), T(588 * 100, 2, [])])],
        'cue_track_zero': [cuesheet(b'', 0, 0, [T(0, 0, [(0, 1)]), T(1000, 255, [])])],
        'cue_cd_track_100': [cuesheet(b'', 88200, 1, [T(0, 100, [(0, 1)]), T(588 * 100, 170, [])])],
        'cue_cd_offset_not_cd_frames': [cuesheet(b'', 88200, 1, [T(1, 1, [(0, 1)]), T(588 * 100, 170, [])])],
        'cue_track_without_index': [cuesheet(b'', 0, 0, [T(0, 1, []), T(1000, 255, [])])],
        'cue_first_index_2': [cuesheet(b'', 0, 0, [T(0, 1, [(0, 2)]), T(1000, 255, [])])],
        'cue_cd_index_offset_not_cd_frames': [cuesheet(b'', 88200, 1, [T(0, 1, [(0, 1), (589, 2)]), T(588 * 100, 170, [])])],
        'cue_index_numbers_skip': [cuesheet(b'', 0, 0, [T(0, 1, [(0, 1), (10, 3)]), T(1000, 255, [])])],
        'cue_duplicate_track_numbers_ok': [cuesheet(b'', 0, 0, [T(0, 1, [(0, 1)]), T(10, 1, [(0, 1)]), T(1000, 255, [])])],
        'pic_legal': [picture(3, b'image/png', b'plain', 1, 1, 24, 0, b'x')],
        'pic_mime_control_char': [picture(3, b'image/\x01png', b'', 1, 1, 24, 0, b'x')],
        'pic_mime_high_bit': [picture(3, b'image/p\xe9g', b'', 1, 1, 24, 0, b'x')],
        'pic_description_bad_utf8': [picture(3, b'image/png', b'caf\xe9', 1, 1, 24, 0, b'x')],
        'pic_description_overlong_utf8': [picture(3, b'image/png', b'\xc0\xaf', 1, 1, 24, 0, b'x')],
        'pic_description_good_utf8': [picture(3, b'image/png', 'caf\u00e9 \u20ac'.encode('utf-8'), 1, 1, 24, 0, b'x')],
    }
