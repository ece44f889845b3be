"""Dynamic instruction counts of the encode and the decode launch WITHOUT a GPU: the headline configuration's shape (48 kHz 16-bit
stereo, level 5, block size 4096 -- or another level / width) on a stream of a few blocks, run on the ISA-level emulator of tests/emu,
which executes the gfx950 instructions of the library in the tree and counts them per kernel.  What SQ_INSTS_VALU / SQ_WAVES count on
the hardware (profiles/r05_pmc.json: 178.9 M VALU wave-instructions per encode launch of 7032 blocks = 25 441 a block = 199
lane-instructions a channel-sample) comes out here per block, exactly, for the code in the tree -- the figure VERDICT round 5 asks the
encoder's instruction diet to be measured by when no GPU is at hand.  (Counts are of the instructions executed; nothing here is a
time.)   usage: python tools/emu_counts.py [--blocks 16] [--level 5] [--bps 16] [--json profiles/r06_emu_counts.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=16)
    ap.add_argument('--level', type=int, default=5)
    ap.add_argument('--bps', type=int, default=16)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    import emurun
    shim, L = emurun.load()
    import numpy as np
    import torch
    from oracle import oracle as O
    from pyflac_amd import batch, synth
    from pyflac_amd.encoder import stream_header_bytes
    sr = 48000 if args.bps == 16 else 96000
    n = args.blocks * 4096
    pcm = (synth.config2_stereo16(n / sr + 0.01, 0, sr) if args.bps == 16 else synth.config4_stereo24(n / sr + 0.01, 1, sr))[:n]
    # (a launch of a few blocks takes fg_pipe_autoc1_kernel since round 6; what is counted here is the headline launch's code on a short
    # stream, so the test-hooks library is told to use the headline's autocorrelation kernel)
    os.environ['FLACGPU_AUTOC1'] = '0'
    ctx = batch.Context(0, testhooks=True)
    s = batch.settings(args.level, 2, args.bps, sr, 4096, True)
    t = torch.from_numpy(pcm.astype(np.int32)).cuda()
    ctx.encode(s, t)                               # (first call: descriptor upload, table set-up)
    shim.gfx950emu_reset_stats()
    out, offs, est = ctx.encode(s, t)
    torch.cuda.synchronize()
    enc = json.loads(shim.gfx950emu_stats_json().decode())
    body = out[:est.total_bytes].cpu().numpy().tobytes()
    cfg, _ = O.config(args.level, 2, args.bps, sr, 4096)
    want, _ = O.encode_stream(cfg, pcm)
    assert stream_header_bytes(s) + body == want, 'the emulated encoder differs from the oracle'
    data = out[:est.total_bytes].clone()
    ctx.decode_stream(data, 2, args.bps, n, nframes=est.nblocks)
    shim.gfx950emu_reset_stats()
    dec, status, dst = ctx.decode_stream(data, 2, args.bps, n, nframes=est.nblocks)
    torch.cuda.synchronize()
    decs = json.loads(shim.gfx950emu_stats_json().decode())
    assert torch.equal(dec, t) and int(status[:, 0].max()) == 0, 'the emulated decoder differs from the input'
    chs = n * 2

    def table(st, title):
        rows = []
        tot = {'valu': 0, 'valu_lanes': 0, 'wave_insts': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'global_load_bytes': 0, 'global_store_bytes': 0}
        for k, v in sorted(st.items(), key=lambda kv: -kv[1]['valu']):
            short = k.split('fg_')[-1][:64] if 'fg_' in k else k[:64]
            rows.append((short, v))
            for key in tot:
                tot[key] += v[key]
        print('== %s: %d blocks, %d channel-samples' % (title, args.blocks, chs))
        print('%-66s %8s %12s %12s %10s %8s %8s' % ('kernel', 'launches', 'VALU/block', 'all/block', 'lane/smp', 'LDS/blk', 'VMEM/blk'))
        for short, v in rows:
            print('%-66s %8d %12.0f %12.0f %10.1f %8.0f %8.0f' % (short, v['launches'], v['valu'] / args.blocks, v['wave_insts'] / args.blocks,
                                                                  v['valu_lanes'] / chs, v['lds'] / args.blocks, v['vmem'] / args.blocks))
        print('%-66s %8s %12.0f %12.0f %10.1f %8.0f %8.0f   (VALU wave-instructions x 64 / channel-samples = %.1f)' %
              ('TOTAL', '', tot['valu'] / args.blocks, tot['wave_insts'] / args.blocks, tot['valu_lanes'] / chs, tot['lds'] / args.blocks,
               tot['vmem'] / args.blocks, tot['valu'] * 64.0 / chs))
        print('global memory bytes per channel-sample: %.2f read, %.2f written (algorithmic: what the instructions ask for, no cache in between)' %
              (tot['global_load_bytes'] / chs, tot['global_store_bytes'] / chs))
        return tot
    te = table(enc, 'encode launch (level %d, %d-bit)' % (args.level, args.bps))
    td = table(decs, 'decode launch, from the bytes alone')
    if args.json:
        from pyflac_amd import _lib
        res = {'what': 'instructions executed on the ISA-level emulator (tests/emu), per launch; bytes == oracle checked in the same run',
               'kernel_id': _lib.lib().flacgpu_kernel_id().decode(), 'blocks': args.blocks, 'level': args.level, 'bps': args.bps, 'channel_samples': chs,
               'encode': {'valu_wave_insts_per_block': te['valu'] / args.blocks, 'valu_lane_insts_per_channel_sample': te['valu_lanes'] / chs,
                          'valu_wave_insts_x64_per_channel_sample': te['valu'] * 64.0 / chs, 'kernels': enc},
               'decode': {'valu_wave_insts_per_block': td['valu'] / args.blocks, 'valu_lane_insts_per_channel_sample': td['valu_lanes'] / chs,
                          'valu_wave_insts_x64_per_channel_sample': td['valu'] * 64.0 / chs, 'kernels': decs}}
        with open(args.json, 'w') as fh:
            json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main()
