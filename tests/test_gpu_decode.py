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


@pytest.mark.parametrize('wps,seconds,ch', [('1', 330.0, 2), ('2', 330.0, 2), ('1', 200.0, 1), ('8', 40.0, 2)])
def test_frames_per_wave_shapes(ctx, monkeypatch, wps, seconds, ch):
    """The parse / restore kernels pack several frames into a wave when a launch has many frames (FLACGPU_DEC_WPS sets
    the target waves per SIMD): decode of a long stream must stay the identity for every packing (here 1..8 frames per
    wave), including the short last frame."""
    import torch
    from pyflac_amd import batch, synth
    monkeypatch.setenv('FLACGPU_DEC_WPS', wps)
    pcm = synth.config2_stereo16(seconds, 3)[:, :ch].copy()
    pcm = pcm[:len(pcm) - 777]                                   # ragged tail frame
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(5, ch, 16, 48000, 4096)
    out, offs, st = ctx.encode(s, t)
    dec, status, _dst = ctx.decode(out[:st.total_bytes], offs.cpu().numpy(), ch, 16, len(pcm))
    assert int(status[:, 0].max()) == 0
    assert torch.equal(dec.reshape(-1, ch), t)


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
