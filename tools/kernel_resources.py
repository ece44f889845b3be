"""Registers, scratch, spills and LDS of every kernel of a HIP source (device assembly through hipcc -S): what the occupancy of
a kernel follows from.  usage: python tools/kernel_resources.py pyflac_amd/csrc/flac_dec_wave.hip [more.hip ...]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only']


def resources(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', out, os.path.abspath(src)], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
        t = open(out).read()
    res = []
    for blk in t.split('  - .agpr_count:')[1:]:
        def g(k):
            m = re.search(r'\.' + k + r':\s+(\S+)', blk)
            return m.group(1) if m else '?'
        name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        res.append((name, g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
    return res


if __name__ == '__main__':
    for src in sys.argv[1:]:
        for r in resources(src):
            print('%-62s vgpr %3s sgpr %3s spilled %2s scratch %4s B  static lds %6s B' % r)
