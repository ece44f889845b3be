"""Per-stage clock stamps of the encode kernel (debug records) + stream-encoder init probe."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth, _lib

ctx = batch.Context(0)
for level, secs in ((5, 2.0), (5, 600.0)):
    pcm = synth.config2_stereo16(secs, 0)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(level, 2, 16, 48000, 4096)
    out, offs, st = ctx.encode(s, t, debug=True)
    out, offs, st = ctx.encode(s, t, debug=True)
    nb = st.nblocks
    recs = ctx.debug_records(0, min(nb, 2000))
    T = np.array([[r.t[k] for k in range(10)] for r in recs], dtype=np.float64)
    d = np.diff(T, axis=1)
    names = ['stage', 'sums+baseline', '-', 'autocorr', 'lpc_decide', 'eval(fixed+lpc)', 'choose', 'pack', 'crc/finish']
    chain = T[:, 3].mean()
    T[:, 3] = T[:, 2]
    tot = (T[:, 9] - T[:, 0]).mean()
    print('level %d, %d blocks, kernel %.3f ms; mean clock64 ticks per stage:' % (level, nb, st.encode_kernel_ms))
    for k, nme in enumerate(names):
        print('   %-18s %10.0f  (%.1f%%)' % (nme, d[:, k].mean(), 100 * d[:, k].mean() / tot))
    print('   total %.0f   (autocorr: chain part %.0f)' % (tot, chain))
    X = np.array([[r.t[k] for k in range(10, 16)] for r in recs], dtype=np.float64).mean(axis=0)
    print('   pack split: other %.0f  passA %.0f  passB %.0f;  eval split: fir %.0f  search %.0f  setup %.0f' % (X[0], X[1], X[2], X[3], X[4], X[5]))
import pyflac_amd
try:
    enc = pyflac_amd.StreamEncoder(48000, lambda b, n, s, f: None, compression_level=5, blocksize=4096)
    enc.process(synth.config2_stereo16(0.5, 3))
    print('stream encoder ok', enc.finish())
except Exception as e:
    print('EXC', repr(e), str(e), _lib.last_error())
