"""The five configurations of BASELINE.json exercised WITHOUT a GPU, at reduced length, on the ISA-level emulator of tests/emu: the
library's host code and its compiled gfx950 kernels, every result compared with the CPU oracle bit for bit.  Not a measurement (the
emulator has no clock worth quoting): what it records is that each configuration's path runs and is bit-exact on the code in the tree,
with the kernels that ran and the instructions they executed.   usage: python tools/emu_configs.py [--json profiles/r06_emu_configs.json]"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    ap.add_argument('--streams', type=int, default=1024)
    args = ap.parse_args()
    import emurun
    shim, L = emurun.load()
    import numpy as np
    import torch
    import pyflac_amd
    from oracle import oracle as O
    from pyflac_amd import batch, shard, synth, _lib
    from pyflac_amd.encoder import stream_header_bytes
    res = {'what': 'BASELINE.json configs on the ISA-level emulator (tests/emu), reduced length, bit-exact against oracle/flac_oracle.c',
           'kernel_id': _lib.lib().flacgpu_kernel_id().decode(), 'build_id': _lib.lib().flacgpu_build_id().decode(), 'configs': {}}

    def stats():
        st = json.loads(shim.gfx950emu_stats_json().decode())
        shim.gfx950emu_reset_stats()
        return {k.split('fg_')[-1].split('EPK')[0].split('EvPK')[0][:70]: {'launches': v['launches'], 'wave_insts': v['wave_insts']} for k, v in st.items() if v['launches']}

    # ---- configs[0]: StreamEncoder -> StreamDecoder passthrough, mono 16-bit 44.1 kHz, 1 s sine (examples/passthrough.py), full size
    t0 = time.time()
    pcm = synth.config1_sine(44100, 1.0)
    pcm = pcm.reshape(-1, 1) if pcm.ndim == 1 else pcm
    chunks, blocks = [], []
    enc = pyflac_amd.StreamEncoder(44100, lambda b, n, s, f: chunks.append(b))
    enc.process(pcm)
    ok = enc.finish()
    stream = b''.join(chunks)
    cfg, _ = O.config(5, 1, 16, 44100, 0)
    want, _ = O.encode_stream(cfg, pcm, finalize=True)
    dec = pyflac_amd.StreamDecoder(lambda a, r, c, n: blocks.append(a))
    dec.process(stream)
    dec.finish()
    got = np.concatenate(blocks, axis=0)
    hdr = 4 + 4 + 34 + 4 + 40
    res['configs']['0 passthrough mono 16-bit 44.1 kHz 1 s sine (full size)'] = {
        'finish': bool(ok), 'frames_equal_oracle': stream[hdr:] == want[hdr:], 'decoded_equals_input': bool(np.array_equal(got.reshape(pcm.shape), pcm)),
        'bytes': len(stream), 'seconds_of_audio': 1.0, 'kernels': stats(), 'wall_s': round(time.time() - t0, 1)}

    ctx = batch.Context(0)

    def single(name, level, bps, sr, seconds, pcm_fn):
        t0 = time.time()
        pcm = pcm_fn(seconds)
        s = batch.settings(level, 2, bps, sr, 4096, True)
        t = torch.from_numpy(pcm.astype(np.int32)).cuda()
        out, offs, est = ctx.encode(s, t)
        body = out[:est.total_bytes].cpu().numpy().tobytes()
        cfg, _ = O.config(level, 2, bps, sr, 4096)
        want, _ = O.encode_stream(cfg, pcm)
        enc_ok = stream_header_bytes(s) + body == want
        ke = stats()
        data = out[:est.total_bytes].clone()
        dec, status, dst = ctx.decode_stream(data, 2, bps, t.shape[0], nframes=est.nblocks)
        dec_ok = bool(torch.equal(dec, t)) and int(status[:, 0].max()) == 0
        kd = stats()
        return {'encode_equals_oracle_whole_stream': enc_ok, 'sha256': hashlib.sha256(body).hexdigest(), 'blocks': int(est.nblocks),
                'decode_from_bytes_equals_source': dec_ok, 'plane_bits': int(dst.plane_bits), 'generic_frames': int(dst.generic_frames),
                'seconds_of_audio': seconds, 'encode_kernels': ke, 'decode_kernels': kd, 'wall_s': round(time.time() - t0, 1)}
    # ---- configs[1] + [2]: single stream, stereo 16-bit 48 kHz, blocksize 4096, level 5 (BASELINE: 600 s; here 6 s)
    res['configs']['1+2 single-stream encode L5 16-bit 48 kHz stereo, then decode of its output from the bytes (6 s of 600)'] = \
        single('stream16', 5, 16, 48000, 6.0, lambda sec: synth.config2_stereo16(sec, 0, 48000))
    # ---- configs[3]: 24-bit 96 kHz stereo, level 8 (BASELINE: 300 s; here 1.5 s)
    res['configs']['3 24-bit 96 kHz stereo encode L8 (1.5 s of 300)'] = \
        single('stream24', 8, 24, 96000, 1.5, lambda sec: synth.config4_stereo24(sec, 1, 96000))
    # ---- configs[4]: 1024 independent stereo streams, level 5, sharded 128 per GPU over 8 GPUs: the eight ranks one after the other on
    # the one emulated device, each with the streams shard.streams_for_rank gives it, one launch a rank (BASELINE: 60 s a stream; here 0.1 s)
    t0 = time.time()
    nst, world, secs = args.streams, 8, 0.1
    s = batch.settings(5, 2, 16, 48000, 4096, True)
    cfg, _ = O.config(5, 2, 16, 48000, 4096)
    hdr_bytes = stream_header_bytes(s)
    all_ok, dec_ok, nblocks, checked = True, True, 0, 0
    for rank in range(world):
        mine = shard.streams_for_rank(nst, rank, world)
        pcms = [synth.config5_stream(i, secs, 48000) for i in mine]
        lengths = [len(p) for p in pcms]
        t = torch.from_numpy(np.concatenate(pcms).astype(np.int32)).cuda()
        out, offs, est = ctx.encode(s, t, stream_lengths=lengths)
        o = offs.cpu().numpy()
        body = out[:est.total_bytes].cpu().numpy().tobytes()
        nblocks += int(est.nblocks)
        # every stream of the rank against the oracle
        fpos = 0
        ranges = []
        for p in pcms:
            want, sizes = O.encode_stream(cfg, p)
            nb = len(sizes)
            got = body[int(o[fpos]):int(o[fpos + nb])]
            all_ok = all_ok and (hdr_bytes + got == want)
            ranges.append((int(o[fpos + nb]) - int(o[fpos]), nb))
            fpos += nb
            checked += 1
        dec, status, dst = ctx.decode_streams(out[:est.total_bytes].clone(), ranges, 2, 16, t.shape[0])
        dec_ok = dec_ok and bool(torch.equal(dec, t)) and int(status[:, 0].max()) == 0
    res['configs']['4 batch of %d independent stereo streams L5, %d per rank over %d ranks (0.1 s a stream of 60)' % (nst, nst // world, world)] = {
        'every_stream_equals_oracle': all_ok, 'streams_checked': checked, 'decode_of_every_rank_from_bytes_equals_source': dec_ok, 'blocks': nblocks,
        'kernels': stats(), 'wall_s': round(time.time() - t0, 1),
        'note': 'the ranks run one after the other on the one emulated device; across real GPUs they share nothing but the 86-byte stream header (pyflac_amd/shard.py)'}
    fault = shim.gfx950emu_last_fault().decode()
    res['fault'] = fault or None
    txt = json.dumps(res, indent=1)
    print(txt if len(txt) < 6000 else txt[:6000] + ' ...')
    if args.json:
        with open(args.json, 'w') as fh:
            fh.write(txt + '\n')
    bad = [k for k, v in res['configs'].items() if not all(val for key, val in v.items() if isinstance(val, bool))]
    if bad or fault:
        raise SystemExit('NOT bit-exact: %s %s' % (bad, fault))


if __name__ == '__main__':
    main()
