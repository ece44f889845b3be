"""profiles/<round>_resource_usage.txt: registers, scratch, spills and occupancy of every kernel instantiation, from
hipcc -Rpass-analysis=kernel-resource-usage with the Makefile's flags.  usage: python tools/resource_usage.py r03"""
import glob, os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'pyflac_amd', 'csrc')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-I' + os.path.join(root, 'include'), '-I' + src,
         '--cuda-device-only', '-c', '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = []
for f in sorted(glob.glob(os.path.join(src, '*.hip'))):
    txt = subprocess.run(['/opt/rocm/bin/hipcc'] + flags + [f], capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in txt.splitlines():
        m = re.search(r'remark: Function Name: (\S+)', line)
        if m:
            name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            rows[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    for k, v in rows.items():
        out.append('%-20s %-60s VGPR %3s AGPR %2s SGPR %3s  scratch %4s B/lane  waves/SIMD %s  VGPR spills %3s  SGPR spills %3s' %
                   (os.path.basename(f)[:-4], k[:60], v.get('VGPRs'), v.get('AGPRs'), v.get('TotalSGPRs'), v.get('ScratchSize [bytes/lane]'),
                    v.get('Occupancy [waves/SIMD]'), v.get('VGPRs Spill'), v.get('SGPRs Spill')))
with open(os.path.join(root, 'profiles', tag + '_resource_usage.txt'), 'w') as fh:
    fh.write('# hipcc -Rpass-analysis=kernel-resource-usage (gfx950, the Makefile\'s flags), one line per kernel instantiation.\n'
             '# Occupancy is the register-limited figure; dynamic LDS (set at launch) can lower it.\n' + '\n'.join(out) + '\n')
print('\n'.join(out))
