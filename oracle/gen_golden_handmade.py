"""Generate tests/golden/handmade_streams.npz: FLAC streams that libFLAC's encoder never writes but its decoder accepts.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_handmade
A small bit writer assembles frames with features outside the encoder presets: escape-coded (raw) Rice partitions with
0..24 raw bits under both coding methods, partition order 8, LPC order 32 with 15-bit coefficients and shift 0, variable
block sizes with every block-size / sample-rate header form and multi-byte sample numbers, wasted bits, verbatim and
constant subframes in all stereo assignments, 8 channels.  Each stream is decoded with the reference's libFLAC 1.4.3 binary
(pyflac/decoder.py:170-196 path); the stream and the PCM it returned are stored, so the decoders under test (oracle, GPU)
are checked against the reference on inputs their own encoders cannot produce.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import libflac_ref as R  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests import cases  # noqa: E402


class BW:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((int(v) >> i) & 1)

    def unary(self, q):
        self.bits.extend([0] * int(q))
        self.bits.append(1)

    def align(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def bytes(self):
        assert len(self.bits) % 8 == 0
        a = np.array(self.bits, np.uint8).reshape(-1, 8)
        return bytes(np.packbits(a, axis=1).ravel().tolist())


def utf8(v):
    if v < 0x80:
        return [v]
    out, n = [], 0
    for n, lim in ((1, 0x800), (2, 0x10000), (3, 0x200000), (4, 0x4000000), (5, 0x80000000), (6, 1 << 36)):
        if v < lim:
            break
    lead = {1: 0xC0, 2: 0xE0, 3: 0xF0, 4: 0xF8, 5: 0xFC, 6: 0xFE}[n]
    out.append(lead | (v >> (6 * n)) & (0xFF >> (n + 2) if n < 6 else 0))
    for k in range(n - 1, -1, -1):
        out.append(0x80 | ((v >> (6 * k)) & 0x3F))
    return out


BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def subframe(bw, r, x, sb, spec):
    """x: int64 samples of this subframe (already shifted right by wasted bits)."""
    n = len(x)
    kind = spec['type']
    wasted = spec.get('wasted', 0)
    hdr = {'constant': 0x00, 'verbatim': 0x02}.get(kind)
    if kind == 'fixed':
        hdr = 0x10 | (spec['order'] << 1)
    elif kind == 'lpc':
        hdr = 0x40 | ((spec['order'] - 1) << 1)
    bw.put(hdr | (1 if wasted else 0), 8)
    if wasted:
        bw.unary(wasted - 1)
    mask = (1 << sb) - 1
    if kind == 'constant':
        bw.put(int(x[0]) & mask, sb)
        return
    if kind == 'verbatim':
        for v in x:
            bw.put(int(v) & mask, sb)
        return
    order = spec['order']
    for v in x[:order]:
        bw.put(int(v) & mask, sb)
    if kind == 'fixed':
        co = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        shift = 0
    else:
        co, shift, prec = spec['q'], spec['shift'], spec['prec']
        bw.put(prec - 1, 4)
        bw.put(shift & 31, 5)
        for c in co:
            bw.put(c & ((1 << prec) - 1), prec)
    res = np.zeros(n, np.int64)
    for i in range(order, n):
        pred = sum(int(co[j]) * int(x[i - 1 - j]) for j in range(order)) >> shift
        res[i] = int(x[i]) - pred
    assert np.abs(res).max() < 2 ** 31
    method, po = spec['method'], spec['po']
    bw.put(method, 2)
    bw.put(po, 4)
    plen, esc = (5, 31) if method else (4, 15)
    psz = n >> po
    assert n % (1 << po) == 0 and (po == 0 or psz >= order)
    for p in range(1 << po):
        lo = order if p == 0 else p * psz
        part = res[lo:(p + 1) * psz]
        mode = spec['part_mode'](p, r)
        if mode == 'escape':
            need = 0 if not len(part) or not part.any() else int(max(int(part.max()).bit_length(), int(-part.min() - 1).bit_length() if part.min() < 0 else 0)) + 1
            raw = min(31, need + int(r.integers(0, 2)))
            if need == 0 and r.random() < 0.7:
                raw = 0
            bw.put(esc, plen)
            bw.put(raw, 5)
            for v in part:
                bw.put(int(v) & ((1 << raw) - 1), raw)
        else:
            mean = float(np.abs(part).mean()) if len(part) else 0.0
            k = int(max(0, np.floor(np.log2(mean + 1)))) + int(r.integers(-1, 2))
            k = int(min(max(k, 0), esc - 1))
            bw.put(k, plen)
            for v in part:
                u = (int(v) << 1) ^ (int(v) >> 63) if v >= 0 else ((-int(v)) << 1) - 1
                bw.unary(u >> k)
                bw.put(u & ((1 << k) - 1), k)


def frame(r, pcm, bps, sr, number, variable, ca, specs, bs_form='auto', sr_form='auto', bps_form='auto'):
    """pcm: int64[n, ch]; returns frame bytes."""
    n, ch = pcm.shape
    bw = BW()
    bw.put(0x3FFE, 14)
    bw.put(0, 1)
    bw.put(1 if variable else 0, 1)
    if bs_form == 'auto' and n in BS_CODES:
        bsc = BS_CODES[n]
    else:
        bsc = 6 if (n <= 256 and bs_form != '16') else 7
    bw.put(bsc, 4)
    if sr_form == 'auto' and sr in SR_CODES:
        src = SR_CODES[sr]
    elif sr_form == 'streaminfo':
        src = 0
    elif sr % 1000 == 0 and sr // 1000 < 256 and sr_form in ('auto', 'khz'):
        src = 12
    elif sr < 65536 and sr_form in ('auto', 'hz'):
        src = 13
    else:
        assert sr % 10 == 0
        src = 14
    bw.put(src, 4)
    bw.put((ch - 1) if ca == 0 else 7 + ca, 4)
    bw.put(0 if bps_form == 'streaminfo' else BPS_CODES[bps], 3)
    bw.put(0, 1)
    for b in utf8(number):
        bw.put(b, 8)
    if bsc == 6:
        bw.put(n - 1, 8)
    elif bsc == 7:
        bw.put(n - 1, 16)
    if src == 12:
        bw.put(sr // 1000, 8)
    elif src == 13:
        bw.put(sr, 16)
    elif src == 14:
        bw.put(sr // 10, 16)
    hb = bw.bytes()
    bw.put(int(O.lib().flo_crc8(hb, len(hb))), 8)
    chans = [pcm[:, c].astype(np.int64) for c in range(ch)]
    if ca == 1:
        chans = [chans[0], chans[0] - chans[1]]
    elif ca == 2:
        chans = [chans[0] - chans[1], chans[1]]
    elif ca == 3:
        chans = [(chans[0] + chans[1]) >> 1, chans[0] - chans[1]]
    for c, x in enumerate(chans):
        sb = bps + (1 if (ca == 1 and c == 1) or (ca == 2 and c == 0) or (ca == 3 and c == 1) else 0)
        w = specs[c].get('wasted', 0)
        assert not (x & ((1 << w) - 1)).any()
        subframe(bw, r, x >> w, sb - w, specs[c])
    bw.align()
    fb = bw.bytes()
    crc = int(O.lib().flo_crc16(fb, len(fb)))
    return fb + bytes([crc >> 8, crc & 0xFF])


def streaminfo(min_bs, max_bs, sr, ch, bps, total):
    bw = BW()
    bw.put(0x664C6143, 32)
    bw.put(1, 1); bw.put(0, 7); bw.put(34, 24)
    bw.put(min_bs, 16); bw.put(max_bs, 16); bw.put(0, 24); bw.put(0, 24)
    bw.put(sr, 20); bw.put(ch - 1, 3); bw.put(bps - 1, 5); bw.put(total, 36)
    for _ in range(16):
        bw.put(0, 8)
    return bw.bytes()


def signal(r, n, ch, bps, wasted=0, kind='sine'):
    amp = (1 << (bps - 2)) - 1
    t = np.arange(n)[:, None]
    if kind == 'noise':
        x = r.normal(0, amp * 0.3, (n, ch))
    else:
        x = amp * 0.6 * np.sin(t * r.uniform(0.01, 0.2, ch) + r.uniform(0, 6, ch)) + r.normal(0, amp * 10 ** r.uniform(-4, -2), (n, ch))
    x = np.clip(np.round(x), -amp, amp).astype(np.int64)
    return (x >> wasted) << wasted


def mode_mix(p_escape):
    return lambda p, r: 'escape' if r.random() < p_escape else 'rice'


def build_streams():
    out = {}
    r = np.random.default_rng(1234)

    def lpc_spec(order, prec, shift, po, method, pesc, wasted=0):
        # a stable, mildly predictive filter: first tap ~0.9, the rest small
        scale = (1 << shift)
        q = [int(round(0.9 * scale))] + [int(r.integers(-scale // 16 - 1, scale // 16 + 2)) for _ in range(order - 1)]
        lim = (1 << (prec - 1)) - 1
        q = [max(-lim - 1, min(lim, v)) for v in q]
        return {'type': 'lpc', 'order': order, 'q': q, 'shift': shift, 'prec': prec, 'po': po, 'method': method,
                'part_mode': mode_mix(pesc), 'wasted': wasted}

    def fixed_spec(order, po, method, pesc, wasted=0):
        return {'type': 'fixed', 'order': order, 'po': po, 'method': method, 'part_mode': mode_mix(pesc), 'wasted': wasted}

    # 1: escape partitions, 16 bit stereo, fixed blocksize 4096, all four assignments
    frames, pcms = [], []
    for i, ca in enumerate([0, 1, 2, 3, 3, 1]):
        x = signal(r, 4096, 2, 16, kind='noise' if i % 2 else 'sine')
        specs = [fixed_spec(int(r.integers(0, 5)), int(r.integers(0, 7)), 0, 0.4), lpc_spec(int(r.integers(1, 13)), 12, 10, int(r.integers(0, 7)), 0, 0.4)]
        frames.append(frame(r, x, 16, 48000, i, False, ca, specs))
        pcms.append(x)
    out['escape16'] = (streaminfo(4096, 4096, 48000, 2, 16, 4096 * 6) + b''.join(frames), np.concatenate(pcms))

    # 2: 24 bit, RICE2 with 5-bit parameters and escapes, partition order 8, LPC order 32 / precision 15 / shift 0 and 14
    frames, pcms = [], []
    cfgs = [(32, 15, 14, 7), (32, 15, 0, 3), (12, 9, 8, 8), (13, 14, 13, 0), (1, 2, 0, 5)]
    for i, (o, prec, sh, po) in enumerate(cfgs):
        x = signal(r, 4096, 2, 24)
        if sh == 0:
            specs = [{'type': 'lpc', 'order': o, 'q': [1] + [0] * (o - 1), 'shift': 0, 'prec': prec, 'po': po, 'method': 1, 'part_mode': mode_mix(0.3)},
                     fixed_spec(4, po, 1, 0.5)]
        else:
            specs = [lpc_spec(o, prec, sh, po, 1, 0.3), fixed_spec(int(r.integers(0, 5)), po, 1, 0.5)]
        frames.append(frame(r, x, 24, 96000, i, False, [3, 0, 1, 2, 3][i], specs))
        pcms.append(x)
    out['rice2_24'] = (streaminfo(4096, 4096, 96000, 2, 24, 4096 * len(cfgs)) + b''.join(frames), np.concatenate(pcms))

    # 3: variable block sizes, every header form, large sample numbers, mono
    frames, pcms = [], []
    pos = (1 << 31) - 5000      # crosses 2^31: 6- and 7-byte UTF-8 numbers
    forms = [(192, 'auto', 'auto'), (17, 'auto', 'khz'), (256, '16', 'hz'), (1000, 'auto', 'tens'), (4608, 'auto', 'streaminfo'),
             (4096, 'auto', 'auto'), (32768, 'auto', 'auto'), (5, 'auto', 'auto'), (65535, 'auto', 'auto')]
    for i, (n, bsf, srf) in enumerate(forms):
        x = signal(r, n, 1, 16)
        po = 0
        while po < 4 and n % (2 << po) == 0 and (n >> (po + 1)) >= 4:
            po += 1
        specs = [fixed_spec(min(2, n - 1), po, 0, 0.3)] if n > 4 else [{'type': 'verbatim'}]
        frames.append(frame(r, x, 16, 32000, pos, True, 0, specs, bs_form=bsf, sr_form=srf, bps_form='streaminfo' if i % 2 else 'auto'))
        pcms.append(x)
        pos += n
    out['variable'] = (streaminfo(5, 65535, 32000, 1, 16, 0) + b''.join(frames), np.concatenate(pcms))

    # 4: 8 channels, 20 bit, wasted bits, constant / verbatim / escape-only subframes, blocksize 1152
    frames, pcms = [], []
    for i in range(4):
        x = signal(r, 1152, 8, 20)
        x[:, 1] = 12345 << 3
        x[:, 2] = (x[:, 2] >> 5) << 5
        x[:, 5] = 0
        specs = [fixed_spec(2, 3, 0, 1.0), {'type': 'constant', 'wasted': 3}, fixed_spec(1, 2, 1, 0.5, wasted=5), {'type': 'verbatim'},
                 lpc_spec(8, 10, 9, 5, 0, 0.2), fixed_spec(0, 7, 0, 1.0), lpc_spec(12, 15, 14, 0, 1, 0.0), fixed_spec(4, 1, 0, 0.0)]
        frames.append(frame(r, x, 20, 44100, i, False, 0, specs))
        pcms.append(x)
    out['eight_ch'] = (streaminfo(1152, 1152, 44100, 8, 20, 1152 * 4) + b''.join(frames), np.concatenate(pcms))

    # 5: 32 bit stereo with a 33-bit side channel in every assignment, verbatim and escape-coded
    frames, pcms = [], []
    for i, ca in enumerate([1, 2, 3, 3]):
        x = signal(r, 1024, 2, 32)
        x[:, 1] = -x[:, 0] + r.integers(-2, 3, 1024)
        x = x * 2 + (1 if i == 3 else 0)
        x = np.clip(x, -2 ** 31, 2 ** 31 - 1)
        side = {'type': 'verbatim'} if i == 0 else fixed_spec(1 + i, 3, 1, 0.5)
        other = fixed_spec(2, 4, 1, 0.3)
        specs = [side, other] if ca == 2 else [other, side]
        frames.append(frame(r, x, 32, 48000, i, False, ca, specs))
        pcms.append(x)
    out['side33'] = (streaminfo(1024, 1024, 48000, 2, 32, 1024 * 4) + b''.join(frames), np.concatenate(pcms))
    return out


def main():
    streams = build_streams()
    store = {}
    for name, (data, pcm) in sorted(streams.items()):
        got, frames, st = R.decode(data, want_frames=False)
        ok = not st['errors'] and got.shape == pcm.shape and np.array_equal(got, pcm.astype(np.int32).reshape(got.shape))
        print('%-12s %7d bytes %3d frames  reference decode %s' % (name, len(data), len(frames), 'matches the construction' if ok else 'DIFFERS: %s' % st))
        assert ok, name
        store[name + '.flac'] = np.frombuffer(data, np.uint8)
        store[name + '.pcm'] = got.astype(np.int32)
    np.savez_compressed(os.path.join(cases.GOLDEN, 'handmade_streams.npz'), **store)


if __name__ == '__main__':
    main()
