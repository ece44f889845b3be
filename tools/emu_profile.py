"""Where a kernel's executed instructions are, without a GPU: the headline shape on the ISA-level emulator (tests/emu) with a count
per instruction, folded into basic blocks of the disassembly and listed by weight.
usage: python tools/emu_profile.py <substring of the kernel name> [--blocks 8] [--level 5] [--bps 16] [--top 25]"""
import argparse, collections, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu')); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('kernel'); ap.add_argument('--blocks', type=int, default=8); ap.add_argument('--level', type=int, default=5)
ap.add_argument('--bps', type=int, default=16); ap.add_argument('--top', type=int, default=25); ap.add_argument('--decode', action='store_true'); ap.add_argument('--lib', default=None, help='another build of the library (e.g. one with -gline-tables-only: gpurun_exp/libflacgpu_gl.so)'); ap.add_argument('--lines', action='store_true', help='fold the counts by source line (needs a library built with line tables)')
args = ap.parse_args()
os.environ['GFX950EMU_PROFILE'] = args.kernel
if args.lib:
    os.environ['FLACGPU_ALLOW_LIBRARY_OVERRIDE'] = '1'; os.environ['FLACGPU_LIBRARY'] = os.path.abspath(args.lib)
os.environ['GFX950EMU_CACHE'] = tempfile.mkdtemp(prefix='gfx950emu_prof_')      # (a disassembly cache of this run's own: kernel names repeat across builds, and the listing must be this build's)
import emurun
shim, L = emurun.load()
import numpy as np, torch
from pyflac_amd import batch, synth
sr = 48000 if args.bps == 16 else 96000
n = args.blocks * 4096
pcm = (synth.config2_stereo16(n / sr + 0.01, 0, sr) if args.bps == 16 else synth.config4_stereo24(n / sr + 0.01, 1, sr))[:n]
# (the headline launch's kernels on a short stream: a launch this small would take fg_pipe_autoc1_kernel -- unless that is what is asked for)
if 'autoc1' not in args.kernel and not args.lib:
    os.environ['FLACGPU_AUTOC1'] = '0'
ctx = batch.Context(0, testhooks='autoc1' not in args.kernel and not args.lib)
s = batch.settings(args.level, 2, args.bps, sr, 4096, True)
t = torch.from_numpy(pcm.astype(np.int32)).cuda()
out, offs, est = ctx.encode(s, t)
if args.decode:
    ctx.decode_stream(out[:est.total_bytes].clone(), 2, args.bps, n, nframes=est.nblocks)
torch.cuda.synchronize()
d = tempfile.mkdtemp()
shim.gfx950emu_write_profiles.argtypes = [__import__('ctypes').c_char_p]
k = shim.gfx950emu_write_profiles(d.encode())
cache = os.environ.get('GFX950EMU_CACHE', '/tmp/gfx950emu_cache')
for i in range(k):
    rows = open(os.path.join(d, 'profile_%d.txt' % i)).read().splitlines()
    name = rows[0][2:]
    cnt = {int(r.split()[1]): int(r.split()[0]) for r in rows[1:]}
    # the disassembly that holds this kernel
    src = None
    for f in os.listdir(cache):
        if f.endswith('.s') and ('<%s>:' % name) in open(os.path.join(cache, f)).read():
            src = open(os.path.join(cache, f)).read().splitlines()
            break
    total = sum(cnt.values())
    if args.lines:
        import subprocess
        elf = None
        for f in os.listdir(cache):
            if f.endswith('.s') and ('<%s>:' % name) in open(os.path.join(cache, f)).read():
                elf = os.path.join(cache, f[:-2] + '.elf')
        txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '-l', '--mcpu=gfx950', elf], capture_output=True, text=True).stdout.splitlines()
        # address -> source line from the -l listing; line of the plain listing -> address
        addr_src, cur = {}, '?'
        for ln in txt:
            if ln.startswith('; ') and re.search(r':\d+$', ln.strip()):
                cur = ln[2:].strip()
            elif ln.startswith('\t') and '//' in ln:
                m = re.search(r'// ([0-9A-F]+):', ln)
                if m:
                    addr_src[int(m.group(1), 16)] = cur
        by = collections.Counter()
        for l, c in cnt.items():
            m = re.search(r'// ([0-9A-F]+):', src[l - 1])
            by[addr_src.get(int(m.group(1), 16), '?') if m else '?'] += c
        print('== %s\n   %d wave-instructions executed (%d a block), by source line:' % (name[:150], total, total // args.blocks))
        for k2, v in by.most_common(args.top):
            print('%6.2f%%  %9d  %s' % (100.0 * v / total, v, k2.replace('/tmp/fgvar_gl/', '')))
        continue
    print('== %s\n   %d wave-instructions executed (%d a block)' % (name[:150], total, total // args.blocks))
    # basic blocks: runs of consecutive lines with the same count
    lines = sorted(cnt)
    runs, cur = [], [lines[0]]
    for a, b in zip(lines, lines[1:]):
        if b == a + 1 and cnt[b] == cnt[a] and not re.search(r's_cbranch|s_branch|s_endpgm|s_barrier', src[a - 1]):
            cur.append(b)
        else:
            runs.append(cur); cur = [b]
    runs.append(cur)
    runs.sort(key=lambda r: -cnt[r[0]] * len(r))
    for r in runs[:args.top]:
        w = cnt[r[0]] * len(r)
        mix = collections.Counter(src[l - 1].split()[0].split('_e32')[0].split('_e64')[0] for l in r)
        print('%6.2f%%  lines %6d-%-6d  %4d instructions x %8d   %s' % (100.0 * w / total, r[0], r[-1], len(r), cnt[r[0]], ' '.join('%s:%d' % kv for kv in mix.most_common(7))))
