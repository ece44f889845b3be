"""GPU parity tests of the decoder: reference fixtures, golden libFLAC streams, error behaviour."""
import os

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from pyflac_amd import batch
    return batch.Context(0)


def _decode(ctx, data):
    import torch
    from pyflac_amd import batch
    offs, si = batch.index_frames(data)
    buf = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
    cap = (len(offs) - 1) * max(si.max_blocksize, 16)
    pcm, status, st = ctx.decode(buf, offs, si.channels, si.bits_per_sample, cap)
    return pcm.cpu().numpy(), status, si


@pytest.mark.parametrize('name', ['mono', 'stereo', 'surround', '32bit', '8bit'])
def test_reference_fixtures(ctx, name):
    """tests/data/*.flac of the reference (libFLAC 1.3.3 / 1.4.2 output) decode to the oracle's PCM; the
    STREAMINFO MD5 pins the PCM for the four pairs the reference ships WAVs for."""
    from oracle import oracle as O
    with open(os.path.join(cases.GOLDEN, 'data', name + '.flac'), 'rb') as f:
        data = f.read()
    want, res = O.decode_stream(data)
    got, status, si = _decode(ctx, data)
    assert int(status[:, 0].max()) == 0
    assert np.array_equal(got, want)
    if name != '8bit':
        assert O.md5_pcm(got, si.bits_per_sample) == bytes(si.md5sum)


def test_golden_streams(ctx, small_streams):
    """Streams produced by the reference's libFLAC 1.4.3 binary (tests/golden/small_streams.npz)."""
    for name, data in sorted(small_streams.items()):
        pcm, bps = cases.make_pcm(cases.ENCODE_CASES[name][0])
        got, status, si = _decode(ctx, data)
        assert int(status[:, 0].max()) == 0, name
        assert np.array_equal(got, np.asarray(pcm).astype(np.int32).reshape(got.shape)), name


def test_crc_mismatch_gives_silence_and_flag(ctx, small_streams):
    data = bytearray(small_streams['cfg1_passthrough'])
    from pyflac_amd import batch
    offs, _si = batch.index_frames(bytes(data))
    # corrupt one byte in the middle of frame 3 (index built from the intact stream)
    data[int(offs[3]) + 40] ^= 0x10
    import torch
    buf = torch.frombuffer(data + bytearray(64), dtype=torch.uint8).cuda()
    pcm, status, st = ctx.decode(buf, offs, 1, 16, 11 * 4096)
    pcm = pcm.cpu().numpy()
    assert status[3, 0] != 0 and (status[np.arange(len(status)) != 3, 0] == 0).all()
    assert not pcm[3 * 4096:4 * 4096].any()
    want, _ = cases.make_pcm({'kind': 'cfg1'})
    assert np.array_equal(pcm[:3 * 4096, 0], want[:3 * 4096, 0])


@pytest.mark.parametrize('seconds,ch', [(330.0, 2), (200.0, 1), (40.0, 2)])
def test_frames_per_wave_shapes(ctx, seconds, ch):
    """Decode of long and short streams, stereo and mono, stays the identity, the short last frame included.  (The parameters are
    those of the frames-per-wave packings of round 2's lane-serial decoder, which left the tree in round 5; the shapes remain.)"""
    import torch
    from pyflac_amd import batch, synth
    pcm = synth.config2_stereo16(seconds, 3)[:, :ch].copy()
    pcm = pcm[:len(pcm) - 777]                                   # ragged tail frame
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(5, ch, 16, 48000, 4096)
    out, offs, st = ctx.encode(s, t)
    dec, status, _dst = ctx.decode(out[:st.total_bytes], offs.cpu().numpy(), ch, 16, len(pcm))
    assert int(status[:, 0].max()) == 0
    assert torch.equal(dec.reshape(-1, ch), t)


@pytest.mark.parametrize('ch,bps,bs,g1', [(2, 16, 256, None), (2, 16, 1152, '48'), (2, 24, 576, '41'), (1, 16, 256, '47'), (2, 16, 4096, '33')])
def test_more_than_32_frames_per_workgroup(ctx, monkeypatch, ch, bps, bs, g1):
    """Launches of more than 32 frames per CU put up to 48 frames into a workgroup of the fused decoder (three rounds of 16
    rows in its output wave instead of two): decode stays the identity -- stereo decorrelation, 24-bit, mono (the general
    output form), a ragged last frame, and group sizes that leave the last round and the last workgroup part empty."""
    import torch
    from pyflac_amd import batch, synth
    nfr = 256 * 36 + 5 if g1 is None else 700
    n = nfr * bs - bs // 3
    pcm = synth.config2_stereo16(n / 48000.0 + 0.01, 11)[:n, :ch].astype(np.int32)
    if bps == 24:
        pcm = pcm * 181 + (np.arange(n, dtype=np.int32)[:, None] % 7)
    t = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    s = batch.settings(5, ch, bps, 48000, bs)
    out, offs, st = ctx.encode(s, t)
    assert st.nblocks == nfr
    dec, status, _dst = ctx.decode(out[:st.total_bytes], offs, ch, bps, n)
    assert int(status[:, 0].max()) == 0
    assert torch.equal(dec.reshape(-1, ch), t)
    dec2, status2, _ = ctx.decode_stream(out[:st.total_bytes], ch, bps, n, nframes=nfr)
    assert int(status2[:, 0].max()) == 0 and torch.equal(dec2.reshape(-1, ch), t)


def test_scans_over_more_than_one_tile(ctx):
    """Launches of more than 8192 frames scan their sizes with one workgroup per tile of 8192 (encoder: byte offsets of the
    frames; decoder: sample offsets): a batch of three ragged streams, 9502 blocks, gives the bytes of the three streams
    encoded one by one (single-tile scans), decodes to the input, and a PCM buffer that is too short is still reported."""
    import torch
    from pyflac_amd import batch, synth
    from pyflac_amd.batch import FlacGpuError
    bs = 256
    lengths = [bs * 4000 + 100, bs * 3000 + 7, bs * 2500]
    total = sum(lengths)
    pcm = synth.config2_stereo16(total / 48000.0 + 0.01, 17)[:total].astype(np.int32)
    t = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    s = batch.settings(5, 2, 16, 48000, bs)
    out, offs, st = ctx.encode(s, t, stream_lengths=lengths)
    assert st.nblocks == 4001 + 3001 + 2500
    whole = out[:st.total_bytes].cpu().numpy().tobytes()
    h_offs = offs.cpu().numpy()
    assert int(h_offs[-1]) == st.total_bytes and (np.diff(h_offs.astype(np.int64)) > 0).all()
    parts, pos = [], 0
    for n in lengths:
        o1, _f1, s1 = ctx.encode(s, t[pos:pos + n].contiguous())
        parts.append(o1[:s1.total_bytes].cpu().numpy().tobytes())
        pos += n
    assert whole == b''.join(parts)
    dec, status, _ = ctx.decode(out[:st.total_bytes], offs, 2, 16, total)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, 2), t)
    with pytest.raises(FlacGpuError):
        ctx.decode(out[:st.total_bytes], offs, 2, 16, total - 1000)


def test_parked_channel_switches_to_32_bits_inside_a_frame(ctx):
    """The fused decoder parks the first coded channel of a stereo frame in HBM as 16-bit values tile by tile (64 samples)
    for as long as they fit, and goes on in 32 bits from the first tile that holds a larger value.  Hand-made frames (16-bit
    stereo, block size 4096, fixed / verbatim subframes): a right-side frame whose side channel L - R is small for the first
    1000 samples and needs 17 bits from there on, a mid-side frame at full scale, a right-side frame that needs 17 bits from
    its first sample, a left-side frame; and the same with a short last frame (general output form)."""
    import torch
    from oracle import gen_golden_handmade as H
    r = np.random.default_rng(77)
    n = 4096
    t = np.arange(n)
    loud = (30000 * np.sign(np.sin(t * 0.05))).astype(np.int64) + r.integers(-200, 200, n)
    quiet = r.integers(-900, 900, n)
    frames, pcms = [], []
    def add(x, ca, m=n):
        specs = [{'type': 'fixed', 'order': 2, 'po': 3, 'method': 0, 'part_mode': H.mode_mix(0.0), 'wasted': 0},
                 {'type': 'verbatim', 'wasted': 0}]
        frames.append(H.frame(r, x[:m], 16, 48000, len(frames), False, ca, specs))
        pcms.append(x[:m])
    a = np.stack([np.where(t < 1000, quiet, loud), np.where(t < 1000, quiet // 2, -loud)], axis=1)
    add(a, 2)
    add(np.stack([loud, -loud + 3], axis=1), 3)
    add(np.stack([loud, -loud], axis=1), 2)
    add(np.stack([loud, -loud], axis=1), 1)
    add(a, 2, 4096 - 1000)
    total = sum(len(p) for p in pcms)
    data = H.streaminfo(n, n, 48000, 2, 16, total) + b''.join(frames)
    want = np.concatenate(pcms).astype(np.int32)
    got, status, si = _decode(ctx, data)
    assert int(status[:, 0].max()) == 0
    assert np.array_equal(got.reshape(-1, 2), want)
    buf = torch.frombuffer(bytearray(data[42:]) + bytearray(64), dtype=torch.uint8).cuda()
    dec2, status2, _ = ctx.decode_stream(buf[:len(data) - 42], 2, 16, total, nframes=len(frames))
    assert int(status2[:, 0].max()) == 0 and np.array_equal(dec2.cpu().numpy().reshape(-1, 2), want)


def test_device_resident_frame_index(ctx):
    """flacgpu_decode_frames_dev: the frame index the encoder wrote is consumed from HBM; same result as the host index."""
    import torch
    from pyflac_amd import batch, synth
    pcm = synth.config2_stereo16(3.0, 5)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(5, 2, 16, 48000, 4096)
    out, offs, st = ctx.encode(s, t)
    assert offs.is_cuda
    a, sa, _ = ctx.decode(out[:st.total_bytes], offs, 2, 16, len(pcm))
    b, sb, _ = ctx.decode(out[:st.total_bytes], offs.cpu().numpy(), 2, 16, len(pcm))
    assert int(sa[:, 0].max()) == 0 and np.array_equal(sa, sb)
    assert torch.equal(a, b) and torch.equal(a.reshape(-1, 2), t)


@pytest.mark.parametrize('name', ['eight_ch', 'escape16', 'rice2_24', 'side33', 'variable'])
def test_handmade_streams(ctx, name, handmade_streams):
    """Streams no libFLAC encoder writes (escape-coded partitions under both coding methods, partition order 8, LPC
    order 32 / 15-bit coefficients / shift 0, variable block sizes with every header form and 36-bit sample numbers,
    wasted bits, 8 channels, 33-bit side channels): same PCM as the reference binary, through the batch entry point and
    through the FLAC__stream_decoder_* callbacks."""
    import torch
    from pyflac_amd import batch
    from tests import abi_decode
    data, pcm = handmade_streams[name]
    offs, si = batch.index_frames(data)
    buf = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
    got, status, st = ctx.decode(buf, offs, si.channels, si.bits_per_sample, len(pcm))
    assert int(status[:, 0].max()) == 0
    assert np.array_equal(got.cpu().numpy().reshape(pcm.shape), pcm)
    res = abi_decode.decode(data)
    assert not res['errors'] and np.array_equal(np.concatenate(res['blocks']).reshape(pcm.shape), pcm)


# ---------------------------------------------------------------------------------------------- frame index on the GPU
def _decode_from_bytes(ctx, data, nframes_known):
    """flacgpu_decode_stream_dev: the audio frames go to the device as bytes, nothing else."""
    import torch
    from pyflac_amd import batch
    offs, si = batch.index_frames(data)            # host index: only to locate the audio and as the expected answer
    audio = data[int(offs[0]):]
    buf = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()[:len(audio)]
    nfr = len(offs) - 1
    got_offs = torch.zeros(nfr + 1, dtype=torch.int64, device='cuda')
    pcm, status, st = ctx.decode_stream(buf, si.channels, si.bits_per_sample, nfr * max(si.max_blocksize, 16),
                                        nframes=nfr if nframes_known else 0, offsets_out=got_offs if nframes_known else None)
    return pcm.cpu().numpy(), status, st, got_offs.cpu().numpy(), offs - offs[0]


@pytest.mark.parametrize('known', [True, False])
@pytest.mark.parametrize('name', ['mono', 'stereo', 'surround', '32bit'])
def test_gpu_frame_index_reference_fixtures(ctx, name, known):
    """The index made on the GPU from the bytes equals the host indexer's, and the decode that follows is the oracle's PCM."""
    from oracle import oracle as O
    with open(os.path.join(cases.GOLDEN, 'data', name + '.flac'), 'rb') as f:
        data = f.read()
    want, _res = O.decode_stream(data)
    got, status, st, goffs, hoffs = _decode_from_bytes(ctx, data, known)
    assert st.nframes == len(hoffs) - 1
    if known:
        assert np.array_equal(goffs.astype(np.uint64), hoffs.astype(np.uint64))
    assert int(status[:, 0].max()) == 0
    assert np.array_equal(got, want)


def test_gpu_frame_index_golden_streams(ctx):
    """Every golden stream of the encoder corpus (all levels, 8..32 bit, odd block sizes, tails) indexes and decodes from
    its bytes alone."""
    from oracle import oracle as O
    sm = np.load(os.path.join(cases.GOLDEN, 'small_streams.npz'))
    n = 0
    for key in sm.files:
        data = sm[key].tobytes()
        want, _res = O.decode_stream(data)
        got, status, st, goffs, hoffs = _decode_from_bytes(ctx, data, True)
        assert np.array_equal(goffs.astype(np.uint64), hoffs.astype(np.uint64)), key
        assert int(status[:, 0].max()) == 0, key
        assert np.array_equal(got, want), key
        n += 1
    assert n > 0


def test_gpu_frame_index_reports_duplicate_frame_numbers(ctx):
    """Two headers that claim the same frame number cannot be filed: the call fails loudly (the host indexer handles it)."""
    import torch
    from pyflac_amd import batch
    with open(os.path.join(cases.GOLDEN, 'data', 'stereo.flac'), 'rb') as f:
        data = f.read()
    offs, si = batch.index_frames(data)
    a, b = int(offs[1]), int(offs[2])
    audio = data[int(offs[0]):b] + data[a:]          # frame 1 twice
    buf = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()[:len(audio)]
    with pytest.raises(batch.FlacGpuError, match='ambiguous'):
        ctx.decode_stream(buf, si.channels, si.bits_per_sample, len(offs) * si.max_blocksize, nframes=len(offs) - 1)


def test_index_finds_nothing_in_garbage(ctx):
    """decode_stream over bytes that are not FLAC, with a frame count to look for (what STREAMINFO would have said): every
    slot of the device index stays empty, the header pass must reject the frames without reading through the empty slots
    (a device fault here takes the process down), and the call reports all frames as bad instead of decoding anything."""
    import torch
    from pyflac_amd import batch
    rng = np.random.default_rng(1)
    for n, nf in ((1 << 20, 100), (5000, 3), (1 << 24, 5000)):
        data = torch.from_numpy(rng.integers(0, 256, n, dtype=np.uint8)).cuda()
        dec, status, st = ctx.decode_stream(data, 2, 16, 1 << 20, nframes=nf)
        assert st.nframes == nf and st.error_frames == nf and st.total_samples == 0
        assert (status[:, 0] != 0).all()
    # a table of device-resident offsets that point outside the stream (flacgpu_decode_frames_dev does not look at them on the host)
    data = torch.from_numpy(rng.integers(0, 256, 4096, dtype=np.uint8)).cuda()
    offs = torch.tensor([0, 1 << 40, 5, 4096], dtype=torch.int64).cuda()
    dec, status, st = ctx.decode(data, offs, 2, 16, 1 << 16)
    assert st.error_frames == 3 and st.total_samples == 0


def test_index_candidate_queue_overflow_fails_loudly(ctx):
    """A stretch of the stream that is nothing but sync codes overflows the index kernel's candidate queue (512 per
    workgroup): the call must say so (the host indexer then takes over in the stream decoder) -- no fault, no silent
    result."""
    import torch
    from pyflac_amd import batch
    data = torch.from_numpy(np.frombuffer(b'\xff\xf8' * (1 << 17), np.uint8).copy()).cuda()
    with pytest.raises(batch.FlacGpuError):
        ctx.decode_stream(data, 2, 16, 1 << 20, nframes=64)
    # and the context is still good for a real stream afterwards
    from pyflac_amd import synth
    pcm = torch.from_numpy(synth.config2_stereo16(1.0, 5).astype(np.int32)).cuda()
    s = batch.settings(5, 2, 16, 48000, 4096, True)
    out, offs, st = ctx.encode(s, pcm)
    dec, status, dst = ctx.decode_stream(out[:st.total_bytes], 2, 16, pcm.shape[0], nframes=st.nblocks)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec[:pcm.shape[0]], pcm)


def test_codes_of_hundreds_of_bytes_leave_the_fast_decoder(ctx):
    """Fuzz seed 79157 (found in round 2): three 32-bit channels of step signals under a small Rice parameter -- single codes
    run over hundreds of bytes, the parser fetches its way through them word by word and leaps past everything the feeding
    wave of the fused kernel expects.  Groups still in flight then shared ring slots with newer ones and overwrote them
    (status 4 on a good frame).  The leap is now detected and the frame handed to the generic decoder; the result is the
    input."""
    import torch
    from oracle import oracle as O
    from tests import fuzzgen
    c = fuzzgen.case(79157)
    cfg, _rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
    a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
    want, _ = O.encode_stream(cfg, a32)
    _pcm, _res, offs = O.decode_stream(want, want_offsets=True)
    o = np.asarray(list(offs) + [len(want)], np.uint64) - 86
    body = torch.from_numpy(np.frombuffer(want[86:], np.uint8).copy()).cuda()
    dec, status, st = ctx.decode(body, o, c['ch'], c['bps'], len(a32))
    assert status[:, 0].tolist() == [0] * (len(o) - 1)
    assert torch.equal(dec.reshape(-1, c['ch'])[:len(a32)].cpu(), torch.from_numpy(a32))


def test_many_streams_decode_from_bytes_alone(ctx):
    """flacgpu_decode_streams_dev (config 5, decode side): several streams laid back to back, every stream numbering its frames
    from 0, are indexed and decoded from their bytes alone in one launch; the index equals the encoder's offsets."""
    import torch
    from pyflac_amd import batch, synth
    streams = [synth.config5_stream(s, 0.3 + 0.11 * s) for s in range(7)]          # ragged lengths: every stream ends in a tail block
    s = batch.settings(5, 2, 16, 48000, 4096, True)
    t = torch.from_numpy(np.concatenate(streams).astype(np.int32)).cuda()
    out, offs, st = ctx.encode(s, t, stream_lengths=[len(x) for x in streams])
    h_offs = offs.cpu().numpy().astype(np.int64)
    ranges, fi = [], 0
    for x in streams:
        nfr = -(-len(x) // 4096)
        ranges.append((int(h_offs[fi + nfr] - h_offs[fi]), nfr))
        fi += nfr
    goffs = torch.zeros(fi + 1, dtype=torch.int64, device='cuda')
    dec, status, dst = ctx.decode_streams(out[:st.total_bytes], ranges, 2, 16, t.shape[0], offsets_out=goffs)
    assert dst.nframes == fi and int(status[:, 0].max()) == 0
    assert np.array_equal(goffs.cpu().numpy(), h_offs[:fi + 1])
    assert torch.equal(dec, t)
    # a frame count that is wrong for one stream: its surplus frames are not found (status != 0), nothing is decoded out of place
    bad = list(ranges)
    bad[2] = (bad[2][0], bad[2][1] + 1)
    dec2, status2, dst2 = ctx.decode_streams(out[:st.total_bytes], bad, 2, 16, t.shape[0] + 4096)
    assert dst2.nframes == fi + 1 and dst2.error_frames >= 1


def test_count_mode_with_frames_shorter_than_any_guess(ctx):
    """Block size 16, mono, silence: frames of 11 or 12 bytes.  With the frame count unknown the status rows are sized by a
    guess; the library writes no more rows than it is told there is room for and reports the count."""
    import torch
    from pyflac_amd import batch
    from oracle import oracle as O
    cfg, _ = O.config(5, 1, 16, 44100, 16, False)
    pcm = np.zeros((16 * 400, 1), np.int16)      # (400 frames: a workgroup of the index kernel parks 512 candidates at most)
    data, sizes = O.encode_stream(cfg, pcm)
    assert max(sizes) <= 13                   # (below the 16 bytes the old guess assumed)
    offs, si = batch.index_frames(data)
    audio = data[int(offs[0]):]
    buf = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()[:len(audio)]
    dec, status, st = ctx.decode_stream(buf, 1, 16, len(pcm), nframes=0)
    assert st.nframes == 400 and status.shape[0] == 400 and int(status[:, 0].max()) == 0
    assert np.array_equal(dec.cpu().numpy(), pcm.astype(np.int32))
    # the raw call with room for three rows: three rows are written, the count is reported
    from pyflac_amd import _lib
    import ctypes as C
    rows = np.full((8, 2), 0xAAAAAAAA, np.uint32)
    st2 = _lib.DecodeStats()
    out = torch.empty((len(pcm), 1), dtype=torch.int32, device='cuda')
    rc = _lib.lib().flacgpu_decode_stream_dev(ctx._h, buf.data_ptr(), buf.numel(), 0, 0, 1, 16, out.data_ptr(), len(pcm), rows.ctypes.data, 3,
                                              None, C.byref(st2))
    assert rc == 0 and st2.nframes == 400
    assert (rows[:3, 0] == 0).all() and (rows[3:] == 0xAAAAAAAA).all()


def test_parser_on_its_own_keeps_every_frame_when_the_parts_of_the_plane_are_rounded(ctx):
    """the parser that starts from the offsets alone places frame f's part of the residual plane at f x stride, the
    stride rounded up to 16 bytes.  With block size x channels no multiple of four the rounding adds up; the scratch area now has
    16 bytes a frame for it, so a long stream of odd blocks stays in the wave parser (flacgpu_decode_stats.generic_frames == 0).
    Before, the frames behind frame ~22 000 of such a stream all went to the generic decoder -- same samples, no sign of it."""
    import torch
    from pyflac_amd import batch, synth
    bs, nfr = 257, 30000
    n = bs * nfr - 100
    pcm = synth.config2_stereo16(n / 48000.0 + 0.01, 4)[:n, :1].astype(np.int32)
    t = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    s = batch.settings(5, 1, 16, 48000, bs)
    out, offs, st = ctx.encode(s, t)
    assert st.nblocks == nfr
    dec, status, dst = ctx.decode_stream(out[:st.total_bytes], 1, 16, n, nframes=nfr)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, 1), t)
    assert dst.generic_frames == 0


class TestResidualPlaneWidth:
    """Round 4: for streams of up to 16 bits the residual plane between the wave parser and the restore kernel is 16 bits wide
    (flac_dec_wave.hip P16; flacgpu_decode_stats.plane_bits).  A frame with a value beyond 16 bits -- a side channel at full scale,
    residuals of loud noise under Rice parameters of 13 and more -- ends the parse with status 6, the call is repeated with 32-bit
    planes and the context keeps them for its next calls.  Same samples either way."""

    @staticmethod
    def roundtrip(pcm16, level=5):
        import torch
        from pyflac_amd import batch
        ctx = batch.Context(0)            # (a context of its own: the hold-off of one case must not reach the next)
        s = batch.settings(level, 2, 16, 48000, 4096, True)
        t = torch.from_numpy(pcm16.astype(np.int32)).cuda()
        out, offs, est = ctx.encode(s, t)
        dec, status, dst = ctx.decode_stream(out[:est.total_bytes], 2, 16, t.shape[0], nframes=est.nblocks)
        assert int(status[:, 0].max()) == 0 and torch.equal(dec, t)
        dec2, status2, dst2 = ctx.decode_stream(out[:est.total_bytes], 2, 16, t.shape[0], nframes=est.nblocks)
        assert int(status2[:, 0].max()) == 0 and torch.equal(dec2, t)
        return dst.plane_bits, dst2.plane_bits

    def test_ordinary_material_keeps_the_16_bit_plane(self):
        from pyflac_amd import synth
        assert self.roundtrip(synth.config2_stereo16(2.0, 3)) == (16, 16)
        assert self.roundtrip(synth.config2_hard16(2.0, 7), level=8) == (16, 16)

    def test_side_channel_at_full_scale_takes_32_bit_planes(self):
        """R = -L - 1 near full scale: mid is constant and the side channel 2 L + 1 -- odd, so no wasted bit takes its 17th bit away."""
        r = np.random.default_rng(5)
        left = (r.integers(-32000, 32000, 48000)).astype(np.int32)
        pcm = np.stack([left, -left - 1], axis=1).astype(np.int16)
        assert self.roundtrip(pcm) == (32, 32)

    def test_full_scale_noise_takes_32_bit_planes(self):
        """Uniform noise over the whole 16-bit range, channels independent: Rice parameters of 13 and 14, verbatim subframes."""
        r = np.random.default_rng(6)
        pcm = r.integers(-32768, 32768, (60000, 2)).astype(np.int16)
        bits = self.roundtrip(pcm)
        assert bits[1] == bits[0] and bits[0] in (16, 32)          # (whatever the material needs; the samples are checked above)

    def test_odd_block_sizes_and_orders_with_the_16_bit_plane(self):
        """Batches that start at odd samples (odd predictor orders, odd block sizes): 2-byte aligned 16-byte stores."""
        import torch
        from pyflac_amd import batch, synth
        ctx = batch.Context(0)
        pcm = synth.config2_stereo16(1.5, 11)
        for level, bs in ((3, 1153), (5, 4095), (8, 2047), (2, 777)):
            s = batch.settings(level, 2, 16, 48000, bs, False)
            t = torch.from_numpy(pcm.astype(np.int32)).cuda()
            out, offs, est = ctx.encode(s, t)
            dec, status, dst = ctx.decode_stream(out[:est.total_bytes], 2, 16, t.shape[0], nframes=est.nblocks)
            assert int(status[:, 0].max()) == 0 and torch.equal(dec, t), (level, bs)
            assert dst.plane_bits == 16


@pytest.mark.parametrize('level,bs', [(1, 1152), (3, 576), (5, 4096), (8, 4096)])
def test_wasted_bits_and_every_channel_assignment_in_whole_regular_tiles(ctx, level, bs):
    """Workgroups of 32 whole stereo frames take the restore kernel's form without a test per sample (round 5): wasted bits in SOME
    frames and channels only (the shift is decided per workgroup), all four channel assignments (left/right, left/side, right/side,
    mid/side), the last workgroup ragged; with the encoder's index and from the bytes alone."""
    import torch
    from pyflac_amd import batch, synth
    n = bs * 70 + 333
    pcm = synth.config2_stereo16(n / 48000.0 + 0.01, 5)[:n].astype(np.int32)
    r = np.random.default_rng(7)
    for f in range(0, 70):
        a, b = f * bs, (f + 1) * bs
        k = f % 7
        if k == 1: pcm[a:b, 0] = (pcm[a:b, 0] >> 2) << 2                     # wasted bits in the left channel
        elif k == 2: pcm[a:b, 1] = (pcm[a:b, 1] >> 3) << 3                   # ... in the right one
        elif k == 3: pcm[a:b] = (pcm[a:b] >> 1) << 1                         # ... in both
        elif k == 4: pcm[a:b, 1] = pcm[a:b, 0] + r.integers(-3, 4, b - a)    # nearly equal channels: a side channel pays
        elif k == 5: pcm[a:b, 1] = -pcm[a:b, 0]                              # opposite channels: mid is nearly nothing
        elif k == 6:                                                         # isolated impulses (wasted bits in a sparse signal)
            pcm[a:b] = 0
            pcm[a + r.integers(0, bs, 5), :] = (r.integers(-8000, 8000, (5, 1)) << 2)
    pcm = np.clip(pcm, -32768, 32767)
    s = batch.settings(level, 2, 16, 48000, bs, False)
    t = torch.from_numpy(np.ascontiguousarray(pcm)).cuda()
    out, offs, est = ctx.encode(s, t)
    data = out[:est.total_bytes].clone()
    dec, status, dst = ctx.decode_stream(data, 2, 16, n, nframes=est.nblocks)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec, t)
    dec, status, dst = ctx.decode(data, offs, 2, 16, n)
    assert int(status[:, 0].max()) == 0 and torch.equal(dec.reshape(-1, 2), t)


_SELF_SCRIPT = r'''
import sys, hashlib, json
import numpy as np, torch
sys.path.insert(0, %r)
from pyflac_amd import batch, synth
ctx = batch.Context(0)
res = []
late = []
pcm = synth.config2_stereo16(3.0, 21)
for level, bs, damage in ((5, 4096, None), (8, 1152, None), (5, 4096, 'header'), (5, 4096, 'payload'), (5, 4096, 'cut'), (3, 4095, None)):
    s = batch.settings(level, 2, 16, 48000, bs, bs != 4095)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    out, offs, est = ctx.encode(s, t)
    data = out[:est.total_bytes].clone()
    o = offs.cpu().numpy()
    if damage == 'header': data[int(o[5]) + 2] ^= 0x40           # the block size code of frame 5
    if damage == 'payload': data[int(o[7]) + 40] ^= 0x10         # a residual bit of frame 7: CRC-16 mismatch or a parse error
    if damage == 'cut': data = data[:int(o[est.nblocks - 1]) + 20].clone()     # the last frame cut short
    dec, status, dst = ctx.decode_stream(data, 2, 16, t.shape[0], nframes=est.nblocks)
    res.append([hashlib.sha256(dec.cpu().numpy().tobytes()).hexdigest(), status[:, 0].tolist(), int(dst.total_samples)])
    late.append(int(dst.join_late_workgroups))
print('RESULT ' + json.dumps(res))
print('LATE ' + json.dumps(late))
'''


def test_parser_on_its_own_equals_parser_behind_the_header_pass():
    """Round 4 (FgDecSelf): decoding one stream from its bytes, the wave parser starts from the frame offsets and the header records
    of the index pass, header pass + scan and the CRC pass beside it.  Same samples and the same status of every frame as with the
    parser queued behind header pass and scan (FLACGPU_DEC_SELF=0, a child process each: the selector is read once) -- whole
    streams, a damaged header, a damaged payload, a last frame cut short, an odd block size."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for v in ('1', '0'):
        env = dict(os.environ, FLACGPU_DEC_SELF=v, PYFLAC_AMD_TESTHOOKS='1')        # (the selector is read by the test-hooks build only)
        p = subprocess.run([sys.executable, '-c', _SELF_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[7:]))
    assert outs[0] == outs[1]
    assert any(any(x != 0 for x in case[1]) for case in outs[0])         # (the damaged streams do report something)


def test_fork_and_join_through_words_in_memory_equal_the_events_and_survive_a_timeout():
    """Round 5: no event record or wait on the main stream of a decode launch.  The resolve kernel raises a word in pinned memory and
    the host queues the side streams' kernels (CRC pass; header pass + scan) when it sees it; their last kernel raises a word the
    restore kernel looks at before it reads the frame table and the verdicts.  Same samples and status words as with the events
    (FLACGPU_DEC_GATE=0, test-hooks build, a child process each); and with the join word muted (FLACGPU_DEC_GATE=2) the restore
    kernel's bounded wait times out, the call is repeated with events and still returns the same."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for v in ('1', '0', '2'):
        env = dict(os.environ, FLACGPU_DEC_GATE=v, PYFLAC_AMD_TESTHOOKS='1')
        p = subprocess.run([sys.executable, '-c', _SELF_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
        outs.append(json.loads(line[7:]))
    assert outs[0] == outs[1] == outs[2]


def test_the_join_word_arriving_late_is_waited_for_and_the_tables_behind_it_are_the_final_ones():
    """Round 6: the restore kernel's wait for its join word made to turn.  FLACGPU_DEC_DELAY_US (test-hooks build) queues a wave
    that idles for 300 us in front of the CRC pass and in front of header pass + scan on their side streams, so that the restore
    kernel -- queued behind the parser on the main stream, which does not wait for them -- starts while the frame table, the block
    offsets and the CRC verdicts are not there yet.  Its workgroups must wait (flacgpu_decode_stats.join_late_workgroups > 0), take
    their acquire fence, and then read final tables: same samples and the same status words, damaged streams included, as a run
    that joins through events (FLACGPU_DEC_GATE=0).  A run with the delay and events checks the hook itself changes nothing."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs, lates = [], []
    for gate, delay in (('1', '300'), ('0', '0'), ('0', '300'), ('1', '0')):
        env = dict(os.environ, FLACGPU_DEC_GATE=gate, FLACGPU_DEC_DELAY_US=delay, PYFLAC_AMD_TESTHOOKS='1')
        p = subprocess.run([sys.executable, '-c', _SELF_SCRIPT % root], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1][7:]))
        lates.append(json.loads([l for l in p.stdout.splitlines() if l.startswith('LATE ')][-1][5:]))
    assert outs[0] == outs[1] == outs[2] == outs[3]
    assert all(n > 0 for n in lates[0]), lates[0]          # every call of the delayed run had workgroups that waited
    assert not any(lates[1]) and not any(lates[2])         # (events: the word is not used)


def test_a_fresh_context_whose_first_decode_is_small_after_another_context_left_its_memory_behind():
    """The join word the restore kernel looks at lives behind the index pass's counters; a context's first decode call of few frames
    keeps the 4 KB its count pass allocated, and that memory may have belonged to a context that is gone (round 5: zeroed for every
    buffer it has not been zeroed for).  One context works on a long stream and is closed; a new one decodes a short stream from its
    bytes alone as its first call -- twenty times over, so that some allocation lands on recycled bytes."""
    import torch
    from pyflac_amd import batch, synth
    big = torch.from_numpy(synth.config2_stereo16(20.0, 9).astype(np.int32)).cuda()
    small = synth.config2_stereo16(0.6, 4).astype(np.int32)                     # 8 frames of 4096
    ts = torch.from_numpy(small).cuda()
    s = batch.settings(5, 2, 16, 48000, 4096, False)
    for _ in range(20):
        a = batch.Context(0)
        out, offs, est = a.encode(s, big)
        dec, status, dst = a.decode_stream(out[:est.total_bytes].clone(), 2, 16, big.shape[0], nframes=est.nblocks)
        assert int(status[:, 0].max()) == 0 and torch.equal(dec, big)
        o2, f2, e2 = a.encode(s, ts)
        data = o2[:e2.total_bytes].clone()
        a.close()
        b = batch.Context(0)
        dec, status, dst = b.decode_stream(data, 2, 16, ts.shape[0], nframes=e2.nblocks)
        assert int(status[:, 0].max()) == 0 and torch.equal(dec, ts)
        b.close()
