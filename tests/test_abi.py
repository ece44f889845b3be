"""The C-ABI library loads on a CPU-only host and exports every symbol include/flacgpu.h declares
(the names pyFLAC's cffi cdef binds: pyflac/builder/encoder.py:266-322, pyflac/builder/decoder.py:387-475)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, 'pyflac_amd', 'libflacgpu.so')):
        g.build()
    from pyflac_amd import _lib
    return _lib.lib()


def test_header_symbols_exported(L):
    with open(os.path.join(ROOT, 'include', 'flacgpu.h')) as f:
        text = f.read()
    names = set(re.findall(r'\b(FLAC__stream_(?:en|de)coder_\w+)\s*\(', text)) | set(re.findall(r'\b(flacgpu_\w+)\s*\(', text))
    names |= set(re.findall(r'extern const char \*const (\w+)\[\]', text)) | {'FLAC__VERSION_STRING', 'FLAC__VENDOR_STRING'}
    assert len(names) > 95
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_string_tables(L):
    from pyflac_amd import _lib
    assert _lib.string_table('FLAC__StreamEncoderStateString', 9)[1] == b'FLAC__STREAM_ENCODER_UNINITIALIZED'
    assert _lib.string_table('FLAC__StreamEncoderInitStatusString', 14)[11] == b'FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE'
    assert _lib.string_table('FLAC__StreamDecoderStateString', 10)[9] == b'FLAC__STREAM_DECODER_UNINITIALIZED'
    assert _lib.string_table('FLAC__StreamDecoderErrorStatusString', 5)[2] == b'FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH'
    assert C.c_char_p.in_dll(L, 'FLAC__VENDOR_STRING').value == b'reference libFLAC 1.4.3 20230623'


def test_settings_resolution_matches_oracle(L):
    """flacgpu_settings_from_level applies libFLAC's init checks in the same order as the oracle."""
    from oracle import oracle as O
    from pyflac_amd import _lib
    cases = [(5, 2, 16, 48000, 0, True), (0, 2, 16, 44100, 0, True), (8, 2, 24, 96000, 4096, True),
             (5, 2, 16, 2000000, 0, True), (5, 2, 16, 44100, 1000000, True), (5, 2, 16, 44100, 65535, True),
             (5, 2, 16, 44100, 65535, False), (5, 9, 16, 44100, 0, True), (5, 2, 3, 44100, 0, True),
             (5, 2, 16, 48000, 4609, True), (5, 2, 16, 96000, 16385, True), (5, 2, 17, 48000, 0, True),
             (5, 1, 8, 8000, 192, True), (3, 6, 16, 48000, 0, True), (5, 2, 16, 48000, 15, True)]
    for level, ch, bps, sr, bs, subset in cases:
        cfg, rc = O.config(level, ch, bps, sr, bs, subset)
        s = _lib.Settings()
        rc2 = L.flacgpu_settings_from_level(C.byref(s), level, ch, bps, sr, bs, 1 if subset else 0)
        assert rc == rc2, (level, ch, bps, sr, bs, subset)
        if rc == 0:
            assert (s.blocksize, s.qlp_coeff_precision, s.do_mid_side, s.loose_mid_side, s.max_lpc_order,
                    s.min_partition_order, s.max_partition_order) == \
                   (cfg.blocksize, cfg.qlp_coeff_precision, cfg.do_mid_side, cfg.loose_mid_side, cfg.max_lpc_order,
                    cfg.min_partition_order, cfg.max_partition_order)
            assert (s.apod_parts >= 2) == (cfg.apod_type == 1) and (s.apod_parts if s.apod_parts >= 2 else 0) == (cfg.apod_parts if cfg.apod_type == 1 else 0)


def test_encoder_object_setters_without_gpu(L):
    """Setter / getter round trips work before init (reference: tests/test_encoder.py:39-88)."""
    vp = C.c_void_p
    enc = vp(L.FLAC__stream_encoder_new())
    assert L.FLAC__stream_encoder_get_state(enc) == 1
    assert L.FLAC__stream_encoder_set_channels(enc, 2) and L.FLAC__stream_encoder_get_channels(enc) == 2
    assert L.FLAC__stream_encoder_set_bits_per_sample(enc, 24) and L.FLAC__stream_encoder_get_bits_per_sample(enc) == 24
    assert L.FLAC__stream_encoder_set_blocksize(enc, 128) and L.FLAC__stream_encoder_get_blocksize(enc) == 128
    assert L.FLAC__stream_encoder_set_compression_level(enc, 8)
    assert L.FLAC__stream_encoder_get_max_lpc_order(enc) == 12 and L.FLAC__stream_encoder_get_max_residual_partition_order(enc) == 6
    assert not L.FLAC__stream_encoder_get_limit_min_bitrate(enc)
    L.FLAC__stream_encoder_delete(enc)


def test_host_frame_indexer_on_fixture(L):
    """flacgpu_index_frames is host code: frame boundaries of a reference fixture equal the oracle's."""
    import numpy as np
    from oracle import oracle as O
    from pyflac_amd import batch
    from tests import cases
    with open(os.path.join(cases.GOLDEN, 'data', 'stereo.flac'), 'rb') as f:
        data = f.read()
    _pcm, res, offs = O.decode_stream(data, want_offsets=True)
    mine, si = batch.index_frames(data)
    assert list(mine[:-1]) == list(offs) and mine[-1] == len(data)
    assert (si.channels, si.bits_per_sample, si.sample_rate, si.total_samples) == (2, 16, 44100, 66150)


def test_no_cpu_fallback_without_gpu(L):
    """Without a HIP device the product refuses to run instead of falling back to a CPU path."""
    if L.flacgpu_device_count() > 0:
        pytest.skip('a GPU is present')
    assert not L.flacgpu_ctx_create(0)
    from pyflac_amd import _lib
    assert 'no CPU fallback' in _lib.last_error()


@pytest.mark.parametrize('name', sorted(__import__('tests.cases', fromlist=['x']).DAMAGE_CASES))
def test_frame_index_resynchronises_after_damage(L, name):
    """Host frame index (flacgpu_index_frames) of the damaged streams of tests/cases.py: every frame whose header
    survived is still delimited at its own start, so one damaged frame never swallows the frames behind it."""
    import numpy as np
    from tests import cases
    from pyflac_amd import batch
    src, edits = cases.DAMAGE_CASES[name]
    with open(os.path.join(cases.GOLDEN, 'data', src + '.flac'), 'rb') as f:
        clean = f.read()
    offs, _ = batch.index_frames(clean)
    got, _ = batch.index_frames(cases.damaged_stream(name))
    got = set(int(x) for x in got)

    def moved(p):
        q = p
        for e in edits:
            if e[0] == 'del' and e[1] < p:
                q -= e[2]
            elif e[0] == 'ins' and e[1] <= p:
                q += e[2]
        return q

    def header_hit(p):
        for e in edits:
            if e[0] == 'trunc' and p + 16 > e[1]:
                return True
            if e[0] in ('flip', 'del') and p <= e[1] < p + 16:
                return True
        return False

    kept = [moved(int(p)) for p in offs[:-1] if not header_hit(int(p))]
    assert len(kept) >= len(offs) - 3 or any(e[0] == 'trunc' for e in edits) or name == 'many_flips'
    assert all(p in got for p in kept), sorted(set(kept) - got)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('seed', [1, 5, 17, 40])
def test_host_frame_index_on_frames_longer_than_any_encoder_makes_them(seed):
    """Valid streams whose frames exceed the verbatim size (escape-coded partitions with more raw bits than the sample
    size: tests/tools/dec_stream_fuzz.py builds them bit by bit) -- the host index bounds a frame by that size to
    resynchronise after damage, so these go through its resync path.  Seed 1 (round 2): the LAST frame of such a stream
    made the search repeat forever (a hang in FLAC__stream_decoder_process_*); seed 5: a chance zero of the running
    CRC-16 inside a 200 KB frame was taken for a frame end.  The offsets must be the oracle's."""
    import importlib.util
    from oracle import oracle as O
    from pyflac_amd import batch
    spec = importlib.util.spec_from_file_location('dsf', os.path.join(os.path.dirname(__file__), 'tools', 'dec_stream_fuzz.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    c = m.case(seed)
    assert c is not None
    data = c[0]
    _p, _r, ooffs = O.decode_stream(data, want_offsets=True)
    hoffs, _si = batch.index_frames(data)
    assert [int(x) for x in hoffs] == [int(x) for x in ooffs] + [len(data)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize('args', [['tests/tools/index_damage_fuzz.py', '0', '500'], ['tests/tools/dec_stream_fuzz.py', 'index', '0', '80']],
                         ids=['damaged_small_block_streams', 'constructed_streams'])
def test_host_frame_index_fuzz_slice(args):
    """The host frame index needs no GPU: a slice of its two fuzzers (random damage on small-block streams under a watchdog;
    constructed streams against the oracle's offsets) runs with the CPU suite."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable] + args, cwd=root, capture_output=True, text=True, timeout=550)
    out = p.stdout.strip().splitlines()
    assert p.returncode == 0 and out and re.search(r'\b0 bad', out[-1]), (out[-6:], p.stderr[-400:])


def test_cuesheet_and_picture_legality_matches_libflac():
    """FLAC__stream_encoder_init_stream refuses the CUESHEET / PICTURE blocks libFLAC refuses (INVALID_METADATA, before any
    device is touched) and no others: the statuses were recorded from the reference binary (oracle/gen_golden_setmeta.py,
    tests/golden/legality_vectors.json).  Without a GPU an acceptable list ends in ENCODER_ERROR instead of OK."""
    import ctypes as C
    import json
    from pyflac_amd import _lib
    from tests import metadata_build as MB
    L = _lib.lib()
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'legality_vectors.json')) as f:
        want = json.load(f)
    INVALID = 12
    assert set(want.values()) == {0, INVALID}
    L.FLAC__stream_encoder_new.restype = C.c_void_p
    L.FLAC__stream_encoder_set_metadata.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.FLAC__stream_encoder_init_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    wcb = _lib.ENC_WRITE_CB(lambda *a: 0)
    seen = 0
    for name, blocks in MB.legality_cases().items():
        enc = C.c_void_p(L.FLAC__stream_encoder_new())
        L.FLAC__stream_encoder_set_channels(enc, 2)
        L.FLAC__stream_encoder_set_bits_per_sample(enc, 16)
        L.FLAC__stream_encoder_set_sample_rate(enc, 44100)
        L.FLAC__stream_encoder_set_compression_level(enc, 5)
        assert L.FLAC__stream_encoder_set_metadata(enc, MB.block_array(blocks), len(blocks))
        rc = L.FLAC__stream_encoder_init_stream(enc, wcb, None, None, None, None)
        assert (rc == INVALID) == (want[name] == INVALID), (name, rc, want[name])
        L.FLAC__stream_encoder_finish(enc)
        L.FLAC__stream_encoder_delete(enc)
        seen += 1
    assert seen == len(want) == 20


def _refwalk(L, data, read_size):
    import ctypes as C
    import numpy as np
    buf = np.frombuffer(data, np.uint8)
    errs = np.zeros(1024, np.uint32)
    frames = np.zeros(2 * 1024, np.uint64)
    nf = C.c_uint64(0)
    ne = L.flacgpu_refwalk_probe(buf.ctypes.data, buf.size, read_size, errs.ctypes.data, errs.size, frames.ctypes.data, frames.size, C.byref(nf))
    assert ne >= 0
    return [int(e) for e in errs[:ne]], [(int(frames[2 * i]), int(frames[2 * i + 1])) for i in range(nf.value)]


def _check_refwalk(L, want, data, read_size):
    """The host replay of libFLAC's reader (csrc/fg_refwalk.h) against the reference's recorded behaviour: the same error
    statuses in the same order; the frames it decodes are frames the reference delivered (the rest of the reference's
    frames are the silence it fills gaps with), and every frame the reference delivered with a signal in it is among them."""
    import hashlib
    import numpy as np
    errs, frames = _refwalk(L, data, read_size)
    assert errs == want['errors']
    delivered = {(f[0], f[1]): f[2] for f in want['frames']}
    assert all(f in delivered for f in frames)
    decoded = set(frames)
    for (sn, bs), h in delivered.items():
        if (sn, bs) in decoded:
            continue
        # a frame the reference delivered and the replay did not decode must be one of the reference's frames of silence
        assert any(h == hashlib.sha256(np.zeros((bs, ch), np.int32).tobytes()).hexdigest()[:16] for ch in range(1, 9)), (sn, bs)


@pytest.mark.parametrize('name', sorted(__import__('tests.cases', fromlist=['x']).DAMAGE_CASES))
@pytest.mark.parametrize('read_size', __import__('tests.cases', fromlist=['x']).DAMAGE_READ_SIZES)
def test_reader_replay_matches_reference_on_damage(L, damage_golden, name, read_size):
    from tests import cases
    _check_refwalk(L, damage_golden[name if read_size == 8192 else '%s@%d' % (name, read_size)], cases.damaged_stream(name), read_size)


@pytest.mark.parametrize('seed', __import__('tests.cases', fromlist=['x']).DAMAGE_FUZZ_SEEDS)
def test_reader_replay_matches_reference_on_random_damage(L, damage_golden, seed):
    from tests import cases
    _src, data, read_size = cases.fuzz_damaged_stream(seed)
    _check_refwalk(L, damage_golden['fuzz%d' % seed], data, read_size)


@pytest.mark.parametrize('seed', __import__('tests.cases', fromlist=['x']).SMALL_DAMAGE_SEEDS)
def test_reader_replay_matches_reference_on_small_streams(L, damage_golden, seed):
    """Frames of a few bytes read in answers of a few bytes: the refills of libFLAC's reader fall everywhere around the damage.
    Round 3's replay walked in a circle on 44 of the first 3000 seeds (a refill 2..7 bytes behind a damaged frame's sync code) and
    gave a wrong sequence on 8; the reader is simulated since (Reader in fg_refwalk.h)."""
    import hashlib
    from tests import cases
    data, read_size = cases.small_damaged_stream(seed)
    want = damage_golden['small%d' % seed]
    assert hashlib.sha256(data).hexdigest()[:16] == want['sha']
    _check_refwalk(L, want, data, read_size)


def test_reader_replay_ends_when_a_refill_falls_just_behind_a_damaged_sync_code(L):
    """STREAMINFO, then 11-byte mono frames, frame 1 with a bad CRC-16 -- at 55-byte answers the refill falls a few
    bytes behind frame 1's sync code, the step back goes to a buffer front at or before it, and the search finds the same frame
    again.  libFLAC reads it a second time without a refill and steps to sync + 2; the replay must end (every read size) and
    report one CRC mismatch per reading plus the search that follows."""
    import numpy as np
    from oracle import oracle as O
    cfg, _ = O.config(5, 1, 16, 44100, 16, subset=False)
    data, sizes = O.encode_stream(cfg, np.full(16 * 6, 1234, np.int32))
    data = bytearray(data)
    second = 86 + int(sizes[0])
    data[second + int(sizes[1]) - 1] ^= 0x55            # CRC-16 of frame 1
    for read_size in list(range(40, 70)) + [1, 2, 3, 7, 8, 9, 8192]:
        errs, frames = _refwalk(L, bytes(data), read_size)
        assert (16, 16) not in frames and (0, 16) in frames and (32, 16) in frames, read_size
        assert errs.count(2) in (1, 2) and len(errs) <= 6, (read_size, errs)


def test_a_release_build_names_the_selectors_it_ignores():
    """scripts that set a kernel selector for the release library measured the default path.  pyflac_amd._lib
    warns on stderr; the rule itself: a build with neither the tuning nor the test-hooks bit reads FLACGPU_DEVICE only."""
    from pyflac_amd import _lib
    env = {'FLACGPU_GROUPS': '1', 'FLACGPU_DEVICE': '0', 'FLACGPU_DEC_GATE': '0', 'PATH': '/bin', 'FLACGPU_LIBRARY': 'x'}
    assert _lib.ignored_selectors(env, 0) == ['FLACGPU_DEC_GATE', 'FLACGPU_GROUPS']
    assert _lib.ignored_selectors(env, 4) == [] and _lib.ignored_selectors(env, 1) == []
    assert _lib.ignored_selectors({'FLACGPU_DEVICE': '1'}, 0) == []
    assert not (_lib.lib().flacgpu_build_flags() & 5)          # (the library in the tree is the release build)
