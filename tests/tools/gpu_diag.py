"""GPU bring-up diagnostic: batch-encode parity cases on the device and compare with the CPU oracle,
stage by stage (uses the debug records the kernel writes).  Not part of the product."""
import sys, os, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pyflac_amd import batch, synth
from oracle import oracle as O
from tests import cases


def compare_case(ctx, name, pcm, bps, sr, level, bs, subset=True, verbose=True, maxblocks=None):
    arr = np.asarray(pcm)
    ch = 1 if arr.ndim == 1 else arr.shape[1]
    a32 = arr.astype(np.int32).reshape(-1, ch)
    cfg, rc = O.config(level, ch, bps, sr, bs, subset)
    s = batch.settings(level, ch, bps, sr, bs, subset)
    assert rc == 0
    ref, sizes = O.encode_stream(cfg, a32)
    t = torch.from_numpy(a32).cuda()
    try:
        out, offs, st = ctx.encode(s, t, debug=True)
    except Exception as e:
        import ctypes as C
        from pyflac_amd import _lib
        res = np.zeros((8, 8), np.uint32)
        _lib.lib().flacgpu_copy_block_results(ctx._h, res.ctypes.data, min(8, (a32.shape[0] + cfg.blocksize - 1) // cfg.blocksize))
        print('%-22s EXC %s\n%s' % (name, e, res)); return False
    torch.cuda.synchronize()
    ob = out[:st.total_bytes].cpu().numpy().tobytes()
    oo = offs.cpu().numpy()
    ok = ob == ref[86:]
    print('%-22s %s  frames=%d bytes=%d/%d kernel=%.3fms' % (name, 'OK' if ok else 'MISMATCH', st.nblocks, len(ob), len(ref) - 86, st.encode_kernel_ms))
    if ok:
        dec, status, dst = ctx.decode(out[:st.total_bytes], oo, ch, bps, a32.shape[0])
        dok = bool(status[:, 0].max() == 0) and dec.shape[0] == a32.shape[0] and bool(torch.equal(dec, t))
        if not dok:
            bad = np.nonzero(status[:, 0])[0]
            print('   DECODE MISMATCH status-bad frames %s, kernel=%.3fms' % (bad[:8], dst.decode_kernel_ms))
            if dec.shape[0] == a32.shape[0]:
                d = (dec != t).any(dim=1).nonzero()[:4].flatten().tolist()
                print('   first differing samples', d, [dec[i].tolist() for i in d], [t[i].tolist() for i in d])
        return ok and dok
    if not verbose:
        return ok
    pos = 86
    nshow = 0
    for b in range(st.nblocks):
        fb = int(sizes[b])
        mine = ob[int(oo[b]):int(oo[b + 1])]
        if mine != ref[pos:pos + fb]:
            blk = a32[b * cfg.blocksize:(b + 1) * cfg.blocksize]
            _b, info = O.encode_frame(cfg, blk, b, want_info=True)
            rec = ctx.debug_records(b, 1)[0]
            print('  frame %d: size %d vs %d  ca ref=%d' % (b, len(mine), fb, info.channel_assignment))
            for c in range(info.n_candidates if info.n_candidates <= 4 else 4):
                oc, gc = info.cand[c], rec.cand[c]
                diffs = []
                if oc.wasted != gc.wasted: diffs.append('wasted %d/%d' % (gc.wasted, oc.wasted))
                if list(oc.fixed_tot) != list(gc.fixed_tot): diffs.append('fixed_tot %s/%s' % (list(gc.fixed_tot), list(oc.fixed_tot)))
                if oc.fixed_guess != gc.fixed_guess: diffs.append('fixed_guess %d/%d' % (gc.fixed_guess, oc.fixed_guess))
                if oc.fixed_bits != gc.fixed_bits: diffs.append('fixed_bits %d/%d' % (gc.fixed_bits, oc.fixed_bits))
                nv = oc.n_vectors
                for v in range(nv):
                    oa = np.array(oc.autoc[v][:cfg.max_lpc_order + 1]); ga = np.array(gc.autoc[v][:cfg.max_lpc_order + 1])
                    if not np.array_equal(oa, ga):
                        diffs.append('autoc[%d] maxrel %.3g' % (v, np.max(np.abs(oa - ga) / (np.abs(oa) + 1e-300))))
                    if oc.lpc_guess[v] != gc.lpc_guess[v]: diffs.append('lpc_guess[%d] %d/%d' % (v, gc.lpc_guess[v], oc.lpc_guess[v]))
                    if oc.lpc_bits[v] != gc.lpc_bits[v]: diffs.append('lpc_bits[%d] %d/%d' % (v, gc.lpc_bits[v], oc.lpc_bits[v]))
                for f in ('type', 'order', 'precision', 'shift', 'porder', 'rice_method', 'bits'):
                    if getattr(oc, f) != getattr(gc, f): diffs.append('%s %d/%d' % (f, getattr(gc, f), getattr(oc, f)))
                if list(oc.qlp) != list(gc.qlp): diffs.append('qlp %s/%s' % (list(gc.qlp)[:oc.order], list(oc.qlp)[:oc.order]))
                if list(oc.rice_params) != list(gc.rice_params): diffs.append('rice_params differ %s/%s' % (list(gc.rice_params)[:8], list(oc.rice_params)[:8]))
                print('    cand %d: %s' % (c, '; '.join(diffs) if diffs else 'analysis identical'))
            # first differing byte
            r = ref[pos:pos + fb]
            k = next((i for i in range(min(len(mine), len(r))) if mine[i] != r[i]), min(len(mine), len(r)))
            print('    first differing byte %d: %s | %s' % (k, mine[max(0, k - 4):k + 8].hex(), r[max(0, k - 4):k + 8].hex()))
            nshow += 1
            if nshow >= 2:
                break
        pos += fb
    return ok


def main():
    print(torch.cuda.get_device_name(0))
    ctx = batch.Context(0)
    names = sys.argv[1:] or ['zeros16_mono', 'const16_st', 'cfg1_passthrough', 'noise16_st', 'sines16_bs16', 'cfg2_1s_l0',
                             'cfg2_1s_l2', 'cfg2_1s_l3', 'cfg2_1s_l5', 'cfg2_1s_l8', 'hard16_l5', 'cfg4_1s_l5', 'cfg4_1s_l8',
                             'wasted4_st', 'lr_equal', 'sines16_ch3', 'sines16_ch8', 'sines16_bs1000', 'sines16_tail3',
                             'sines8_st', 'sines12_st', 'sines20_st', 'sines24_l8_bs4608', 'walk32_l8', 'fixture_stereo_l5',
                             'fixture_32bit_l5', 'fixture_surround_l5', 'cfg2_1s_l1', 'cfg2_1s_l4', 'cfg2_1s_l6', 'sines16_sr12345_lax']
    nok = 0
    for n in names:
        spec, sr, level, bs, subset = cases.ENCODE_CASES[n]
        pcm, bps = cases.make_pcm(spec)
        try:
            nok += bool(compare_case(ctx, n, pcm, bps, sr, level, bs, subset))
        except Exception as e:
            import traceback; traceback.print_exc()
    print('%d / %d cases bit-exact' % (nok, len(names)))


if __name__ == '__main__':
    main()
