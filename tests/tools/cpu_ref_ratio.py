#!/usr/bin/env python3
"""Measure how much faster the reference binary (pyFLAC's bundled libFLAC 1.4.3, AVX2 build) is than the scalar oracle
(oracle/flac_oracle.c) on the bench's own input, in the BUILD container (the binary does not travel to the GPU box).
bench.py reports the oracle's rate as cpu_baseline (kind "port") and quotes this ratio as a number.

  python tests/tools/cpu_ref_ratio.py [seconds] > profiles/r02_cpu_ref_ratio.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import libflac_ref as R     # noqa: E402
from oracle import oracle as O           # noqa: E402
from pyflac_amd import synth             # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
pcm = synth.config2_stereo16(secs, 0, 48000)
a32 = np.ascontiguousarray(pcm.astype(np.int32))
cfg, _ = O.config(5, 2, 16, 48000, 4096, True)
# warm both paths (library load, window tables)
O.encode_stream(cfg, a32[:4096 * 4])
R.encode(pcm[:4096 * 4], 48000, level=5, blocksize=4096, extra=[('set_do_md5', 0)])


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts)


t_or = best(lambda: O.encode_stream(cfg, a32))
t_ref = best(lambda: R.encode(pcm, 48000, level=5, blocksize=4096, extra=[('set_do_md5', 0)]))
stream, _ = O.encode_stream(cfg, a32)
t_od = best(lambda: O.decode_stream(stream)) / 2          # oracle.decode_stream makes two passes
t_rd = best(lambda: R.decode(stream, want_frames=False))
n = a32.size
print(json.dumps({
    'what': 'reference binary (libFLAC 1.4.3 as bundled with pyFLAC 3.0.0) vs oracle/flac_oracle.c, 1 thread, MD5 off, best of 5',
    'host': 'build container (%d CPUs)' % (os.cpu_count() or 0),
    'sample': '%.0f s stereo 16-bit 48 kHz, level 5, blocksize 4096 (pyflac_amd.synth.config2_stereo16)' % secs,
    'command': 'python tests/tools/cpu_ref_ratio.py %.0f' % secs,
    'oracle_encode_msamples_per_s': round(n / t_or / 1e6, 2), 'reference_encode_msamples_per_s': round(n / t_ref / 1e6, 2),
    'oracle_decode_msamples_per_s': round(n / t_od / 1e6, 2), 'reference_decode_msamples_per_s': round(n / t_rd / 1e6, 2),
    'reference_over_oracle_encode': round(t_or / t_ref, 3), 'reference_over_oracle_decode': round(t_od / t_rd, 3),
}, indent=1))
