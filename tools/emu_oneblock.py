"""The serial path of a one-block encode call -- StreamEncoder.process() with libFLAC's timing -- WITHOUT a GPU: per kernel of the call the
instructions its LONGEST wave executes (tests/emu counts them per wave), with fg_pipe_autoc1_kernel (round 6: a workgroup a block, one
wave on the chains, eight staging) and with fg_pipe_autoc_kernel forced on the same call (FLACGPU_AUTOC1=0, test-hooks library: a wave
a block, which stages its own chunks).  Counts, not times: what they bound is the issue time of the wave everything else waits for.
usage: python tools/emu_oneblock.py [--json file] > profiles/r06_emu_oneblock.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, ROOT)
import emurun  # noqa: E402
shim, L = emurun.load()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from pyflac_amd import batch, synth, _lib  # noqa: E402

res = {}
print('# kernel id %s; one block of 4096 stereo samples per call; longest = instructions executed by the wave that executes most' % _lib.lib().flacgpu_kernel_id().decode())
for level, bps in ((5, 16), (8, 24)):
    sr = 48000 if bps == 16 else 96000
    pcm = (synth.config2_stereo16(0.2, 0, sr) if bps == 16 else synth.config4_stereo24(0.2, 1, sr))[:4096]
    for hooks, sel, what in ((False, None, 'release library (fg_pipe_autoc1_kernel)'), (True, '0', 'FLACGPU_AUTOC1=0 (fg_pipe_autoc_kernel)')):
        if sel is not None:
            os.environ['FLACGPU_AUTOC1'] = sel
        ctx = batch.Context(0, testhooks=hooks)
        s = batch.settings(level, 2, bps, sr, 4096, True)
        t = torch.from_numpy(pcm.astype(np.int32)).cuda()
        ctx.encode(s, t)
        shim.gfx950emu_reset_stats()
        ctx.encode(s, t)
        torch.cuda.synchronize()
        st = json.loads(shim.gfx950emu_stats_json().decode())
        os.environ.pop('FLACGPU_AUTOC1', None)
        print('== level %d, %d bit: %s' % (level, bps, what))
        tot = 0
        for k, v in sorted(st.items(), key=lambda kv: -kv[1]['max_wave_insts']):
            name = k.split('fg_')[-1][:48]
            print('   %-50s waves %3d   all %7d   longest %6d   matrix %5d' % (name, v['waves'], v['wave_insts'], v['max_wave_insts'], v['mfma']))
            tot += v['max_wave_insts']
        print('   sum of the longest waves of the call\'s kernels (they run one behind the other): %d' % tot)
        res['l%d_%dbit_%s' % (level, bps, 'release' if sel is None else 'autoc1_0')] = {'serial_path': tot, 'longest': {k.split('fg_')[-1].split('IL')[0].split('EP')[0]: v['max_wave_insts'] for k, v in st.items()}}
if '--json' in sys.argv:
    with open(sys.argv[sys.argv.index('--json') + 1], 'w') as fh:
        json.dump(res, fh, indent=1)
