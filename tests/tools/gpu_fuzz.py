"""Differential fuzz on the GPU box: libflacgpu's batch encoder against the CPU oracle (byte-identical frames) and the
GPU decoder against the input, on tests/fuzzgen.py cases.  usage: python tests/tools/gpu_fuzz.py [first] [count]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import oracle as O
from pyflac_amd import batch
from pyflac_amd.encoder import stream_header_bytes
from tests import fuzzgen

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = batch.Context(0)
bad = ran = 0
for seed in range(first, first + count):
    c = fuzzgen.case(seed)
    tag = 'seed %d ch%d bps%d l%d bs%d sr%d n%d %s%s' % (seed, c['ch'], c['bps'], c['level'], c['bs'], c['sr'], len(c['pcm']), c['kind'],
                                                     ' lmb' if c['limit_min_bitrate'] else '')
    cfg, rc = O.config(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
    try:
        s = batch.settings(c['level'], c['ch'], c['bps'], c['sr'], c['bs'], c['subset'])
        grc = 0
    except batch.FlacGpuError:
        grc = 1
    if (rc != 0) != (grc != 0):
        print('INIT STATUS', tag, rc, grc); bad += 1
        continue
    if rc:
        continue
    ran += 1
    cfg.limit_min_bitrate = 1 if c['limit_min_bitrate'] else 0
    s.limit_min_bitrate = cfg.limit_min_bitrate
    a32 = np.ascontiguousarray(c['pcm'].astype(np.int32))
    want, _ = O.encode_stream(cfg, a32)
    t = torch.from_numpy(a32).cuda()
    try:
        out, offs, st = ctx.encode(s, t)
    except batch.FlacGpuError as e:
        print('ENCODE ERROR', tag, e); bad += 1
        continue
    got = stream_header_bytes(s) + out[:st.total_bytes].cpu().numpy().tobytes()
    if got != want:
        k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        h = offs.cpu().numpy()
        fr = int(np.searchsorted(h, k - 86, side='right') - 1) if k >= 86 else -1
        print('ENCODE DIFF', tag, 'len', len(got), len(want), 'first diff byte', k, 'frame', fr); bad += 1
        continue
    dec, status, _ = ctx.decode(out[:st.total_bytes], offs, c['ch'], c['bps'], len(a32))
    if int(status[:, 0].max()) != 0 or not torch.equal(dec.reshape(-1, c['ch']), t):
        print('DECODE DIFF', tag, 'status', status[:, 0].tolist()[:8]); bad += 1
        continue
    # the same bytes again without the encoder's index: the frames are found on the GPU (sync code, header, CRC-8, frame number)
    try:
        dec2, status2, st2 = ctx.decode_stream(out[:st.total_bytes], c['ch'], c['bps'], len(a32), nframes=st.nblocks)
        if st2.nframes != st.nblocks or int(status2[:, 0].max()) != 0 or not torch.equal(dec2.reshape(-1, c['ch'])[:len(a32)], t):
            print('INDEX DECODE DIFF', tag, 'frames', st2.nframes, st.nblocks, 'status', status2[:, 0].tolist()[:8]); bad += 1
    except batch.FlacGpuError as e:
        # (a chance header inside the frames that the positions cannot settle: the call says so and the host indexer takes over)
        if 'ambiguous' not in str(e):
            print('INDEX DECODE ERROR', tag, e); bad += 1
print('cases %d..%d: ran %d, %d bad' % (first, first + count - 1, ran, bad))
