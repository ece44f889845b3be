"""Direct packing path vs the chunk form on the headline workload: first differing frame, repeated calls."""
import sys, hashlib
import numpy as np, torch
sys.path.insert(0, '.')
from pyflac_amd import batch, synth, _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pcm = synth.config2_stereo16(secs, 3)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
a, b = batch.Context(0), batch.Context(0)
_lib.lib().flacgpu_set_direct(b._h, 0)
s = batch.settings(5, 2, 16, 48000, 4096)
ob, fb, sb = b.encode(s, t)
ref = ob[:sb.total_bytes].clone(); fref = fb.clone()
oa = fa = None
for r in range(reps):
    oa, fa, sa = a.encode(s, t, out=oa, offsets=fa)
    same_idx = torch.equal(fa, fref)
    same = sa.total_bytes == sb.total_bytes and torch.equal(oa[:sa.total_bytes], ref)
    print('rep', r, 'direct_path', sa.direct_path, 'bytes', sa.total_bytes, sb.total_bytes, 'index equal', same_idx, 'bytes equal', same, 'gpu ms %.3f' % sa.total_gpu_ms)
    if not same:
        fo = fa.cpu().numpy(); fr = fref.cpu().numpy()
        bad = np.nonzero(fo != fr)[0]
        print(' first index difference at frame', bad[:5], fo[bad[:5]], fr[bad[:5]])
        x = oa[:sb.total_bytes].cpu().numpy(); y = ref.cpu().numpy()
        d = np.nonzero(x != y)[0]
        if len(d):
            import bisect
            f = bisect.bisect_right(list(fr), int(d[0])) - 1
            print(' first byte difference at', d[0], 'frame', f, 'offset in frame', d[0] - fr[f], 'frame bytes', fr[f + 1] - fr[f], 'differing bytes', len(d))
            fs = sorted(set(bisect.bisect_right(list(fr), int(q)) - 1 for q in d[:100000:50]))
            print(' frames hit (sample):', fs[:40], '... total sample', len(fs))
