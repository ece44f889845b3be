"""Decoder fuzz on constructed streams: random VALID FLAC streams with features no libFLAC encoder writes in these
combinations (escape-coded partitions under both coding methods, any partition order the block allows, LPC orders up to 32
with 2..15-bit coefficients and shifts 0..14, wasted bits, verbatim / constant subframes, every stereo assignment, 1..8
channels, 8..32 bits, odd block sizes) are assembled bit by bit (oracle/gen_golden_handmade.py's writer); what every decoder
must return is what the reference binary returns (which is the construction itself in all but a few 33-bit cases).

  python tests/tools/dec_stream_fuzz.py ref [first] [count]    build container: the reference binary and the oracle
  python tests/tools/dec_stream_fuzz.py index [first] [count]  anywhere: the host frame index against the oracle's offsets
  python tests/tools/dec_stream_fuzz.py gpu [first] [count]    GPU box: batch decode (host index and device index) + the
                                                               FLAC__stream_decoder_* callbacks, and the oracle again
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gen_golden_handmade as H
from oracle import oracle as O


def case(seed):
    r = np.random.default_rng(800000 + seed)
    ch = int(r.choice([1, 2, 2, 2, 3, 5, 8]))
    bps = int(r.choice([8, 12, 16, 16, 20, 24, 24, 32]))
    sr = int(r.choice(list(H.SR_CODES)))
    n = int(r.choice([192, 576, 1152, 256, 512, 1024, 2048, 4096, 4608, int(r.integers(16, 5000))]))
    nfr = int(r.integers(1, 5))
    frames, pcms = [], []
    for i in range(nfr):
        m = n if i + 1 < nfr or r.random() < 0.5 else int(r.integers(1, n + 1))       # a shorter last block
        x = H.signal(r, m, ch, bps, kind='noise' if r.random() < 0.3 else 'sine')
        ca = int(r.integers(0, 4)) if ch == 2 else 0
        if bps == 32 and ca:
            x = x >> 1                                                                # keep the 33-bit side inside int64 arithmetic comfortably
        specs = []
        for c in range(ch):
            w = int(r.choice([0, 0, 0, 1, 3]))
            kind = r.choice(['fixed', 'fixed', 'lpc', 'lpc', 'verbatim', 'constant'])
            if kind == 'constant':
                x[:, c] = int(r.integers(-(1 << (bps - 2)), 1 << (bps - 2)))
            if w and ca == 0:
                x[:, c] = (x[:, c] >> w) << w
            else:
                w = 0
            po = 0
            order = 0
            if kind == 'fixed':
                order = int(r.integers(0, max(1, min(5, m))))
            elif kind == 'lpc':
                order = int(r.integers(1, max(2, min(33, m)))) if m > 1 else 0
            while po < 8 and m % (2 << po) == 0 and (m >> (po + 1)) >= max(order, 1) and r.random() < 0.8:
                po += 1
            method = int(r.integers(0, 2))
            pesc = float(r.choice([0.0, 0.0, 0.3, 1.0]))
            if kind == 'lpc' and bps < 32 and order >= 1:
                shift = int(r.integers(0, 15))
                prec = int(r.integers(max(2, min(15, shift + 2)), 16))
                scale = 1 << shift
                q = [int(round(0.9 * scale))] + [int(r.integers(-scale // 16 - 1, scale // 16 + 2)) for _ in range(order - 1)]
                lim = (1 << (prec - 1)) - 1
                q = [max(-lim - 1, min(lim, v)) for v in q]
                specs.append({'type': 'lpc', 'order': order, 'q': q, 'shift': shift, 'prec': prec, 'po': po, 'method': method,
                              'part_mode': H.mode_mix(pesc), 'wasted': w})
            elif kind in ('fixed', 'lpc'):
                specs.append({'type': 'fixed', 'order': min(order, 4), 'po': po, 'method': method, 'part_mode': H.mode_mix(pesc), 'wasted': w})
            else:
                specs.append({'type': str(kind), 'wasted': w})
        if ch == 2 and ca:
            # wasted bits / constants were chosen per input channel: with a stereo assignment the coded channels are others
            for sp in specs:
                sp['wasted'] = 0
                if sp['type'] == 'constant':
                    sp['type'] = 'verbatim'
        try:
            frames.append(H.frame(r, x, bps, sr, i, False, ca, specs))
        except AssertionError:
            return None                                  # (a residual outside 32 bits, a partition smaller than the order: skip)
        pcms.append(x)
    total = sum(len(p) for p in pcms)
    return H.streaminfo(n, n, sr, ch, bps, total) + b''.join(frames), np.concatenate(pcms).astype(np.int64), (ch, bps, n, nfr)


def main():
    mode = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    bad = ran = odd = 0
    if mode == 'gpu':
        import torch
        from pyflac_amd import batch
        from tests import abi_decode
        ctx = batch.Context(0)
    elif mode == 'ref':
        from oracle import libflac_ref as R
    else:
        from pyflac_amd import batch        # 'index': the host frame index (no GPU needed) against the oracle's frame offsets
    import time
    t_start = time.time()
    for seed in range(first, first + count):
        if (seed - first) % 100 == 99:
            print('... %d cases, %.0f s' % (seed - first + 1, time.time() - t_start), flush=True)
        c = case(seed)
        if c is None:
            continue
        data, pcm, info = c
        ran += 1
        tag = 'seed %d ch%d bps%d n%d frames%d' % ((seed,) + info)
        try:
            got, _res = O.decode_stream(data)
        except ValueError as e:
            got = None
        if mode == 'ref':
            # the pin: the oracle's decoder equals the reference binary on this stream (whether or not the stream says
            # what the construction meant -- a few 33-bit side-channel constructions do not, and are counted apart)
            g2, _frames, st = R.decode(data, want_frames=False)
            same = got is not None and not st['errors'] and got.shape == g2.shape and np.array_equal(got, g2)
            if not same and not (got is None and st['errors']):
                print('ORACLE != REFERENCE', tag, st['errors'][:4]); bad += 1
            elif got is not None and (got.shape != pcm.shape or not np.array_equal(got, pcm.astype(np.int32))):
                odd += 1
            continue
        if got is None:
            continue
        want = got
        if mode == 'index':
            _p, _r, ooffs = O.decode_stream(data, want_offsets=True)
            hoffs, _si = batch.index_frames(data)
            if [int(x) for x in hoffs] != [int(x) for x in ooffs] + [len(data)]:
                print('HOST INDEX DIFF', tag, [int(x) for x in hoffs][:8], [int(x) for x in ooffs][:8]); bad += 1
            continue
        offs, si = batch.index_frames(data)
        buf = torch.frombuffer(bytearray(data) + bytearray(64), dtype=torch.uint8).cuda()
        d1, status, _st = ctx.decode(buf, offs, si.channels, si.bits_per_sample, len(want))
        if int(status[:, 0].max()) != 0 or not np.array_equal(d1.cpu().numpy().reshape(want.shape), want):
            print('GPU DECODE DIFF', tag, status[:, 0].tolist()); bad += 1
            continue
        res = abi_decode.decode(data)
        if res['errors'] or not np.array_equal(np.concatenate(res['blocks']).reshape(want.shape), want):
            print('GPU API DECODE DIFF', tag, res['errors'][:4]); bad += 1
            continue
        if info[3] > 1 and len(set(np.diff(offs))) >= 1:
            audio = data[int(offs[0]):]
            ab = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()[:len(audio)]
            try:
                d2, status2, st2 = ctx.decode_stream(ab, si.channels, si.bits_per_sample, len(want), nframes=len(offs) - 1)
                if st2.nframes != len(offs) - 1 or int(status2[:, 0].max()) != 0 or not np.array_equal(d2.cpu().numpy()[:len(want)].reshape(want.shape), want):
                    print('GPU INDEX DECODE DIFF', tag, status2[:, 0].tolist()); bad += 1
            except batch.FlacGpuError as e:
                if 'ambiguous' not in str(e):
                    print('GPU INDEX DECODE ERROR', tag, e); bad += 1
    print('constructed-stream fuzz (%s) %d..%d: ran %d, %d bad%s' % (mode, first, first + count - 1, ran, bad,
          ', %d streams decode to something else than their construction (generator)' % odd if odd else ''))


if __name__ == '__main__':
    main()
