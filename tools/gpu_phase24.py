"""Per-stage clock stamps of the encode kernel on the 24-bit / level-8 workload (BASELINE configs[3])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyflac_amd import batch, synth

ctx = batch.Context(0)
for level, bps, gen, sr, secs in ((8, 24, synth.config4_stereo24, 96000, 60.0), (5, 24, synth.config4_stereo24, 96000, 60.0), (8, 16, synth.config2_stereo16, 48000, 120.0)):
    pcm = gen(secs, 1, sr) if bps == 24 else gen(secs, 0, sr)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    s = batch.settings(level, 2, bps, sr, 4096)
    out, offs, st = ctx.encode(s, t, debug=True)
    out, offs, st = ctx.encode(s, t, debug=True)
    nb = st.nblocks
    recs = ctx.debug_records(0, min(nb, 1500))
    T = np.array([[r.t[k] for k in range(10)] for r in recs], dtype=np.float64)
    d = np.diff(T, axis=1)
    names = ['stage', 'sums+baseline', '-', 'autocorr', 'lpc_decide', 'eval(fixed+lpc)', 'choose', 'pack', 'crc/finish']
    chain = T[:, 3].mean()
    T[:, 3] = T[:, 2]
    tot = (T[:, 9] - T[:, 0]).mean()
    print('level %d %d-bit, %d blocks, kernel %.3f ms (debug on); mean clock64 ticks per stage:' % (level, bps, nb, st.encode_kernel_ms))
    for k, nme in enumerate(names):
        print('   %-18s %10.0f  (%.1f%%)' % (nme, d[:, k].mean(), 100 * d[:, k].mean() / tot))
    X = np.array([[r.t[k] for k in range(10, 16)] for r in recs], dtype=np.float64).mean(axis=0)
    print('   total %.0f;  pack split: other %.0f  passA %.0f  passB %.0f;  eval split: fir %.0f  search %.0f  setup %.0f' % (tot, X[0], X[1], X[2], X[3], X[4], X[5]))
