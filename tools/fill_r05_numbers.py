"""Fill the @TOKENS@ of DESIGN.md / README.md from a collection's profiles: python tools/fill_r05_numbers.py <letter>"""
import json, sys, re
L = sys.argv[1]
b = json.load(open('profiles/r05_%s_bench.json' % L))
p = json.load(open('profiles/r05_pmc.json'))
alg = b['roofline']['algorithmic_bytes_per_launch']
c = b['configs']
tok = {
    'L': L,
    'ENC': 'encode %.3f ms' % b['encode_gpu_ms'], 'DEC': 'decode %.3f ms' % b['decode_gpu_ms'],
    'ETR': '%.2f' % (p['encode_traffic_bytes_per_launch'] / 1e9), 'ETX': '%.1f' % (p['encode_traffic_bytes_per_launch'] / alg),
    'EVA': '%.0f' % (p['encode_valu_insts_per_launch'] / 1e6), 'DVA': '%.0f' % (p['decode_valu_insts_per_launch'] / 1e6),
    'DTR': '%.2f' % (p['decode_traffic_bytes_per_launch'] / 1e9),
    'STEP': '%.2f' % b['ms_per_step'], 'VAL': '%.1f' % (b['value'] / 1e3),
    'S24E': '%.2f' % c['stream24']['encode_gpu_ms'], 'S24V': '%.1f' % (c['stream24']['value'] / 1e3),
    'BE': '%.2f' % c['batch']['encode_gpu_ms'], 'BD': '%.2f' % c['batch']['decode_gpu_ms'], 'BV': '%.0f' % (c['batch']['value'] / 1e3),
    'MD5': '%.0f' % c['batch']['md5_on']['md5_kernel_ms'],
    'S32': '%.1f' % (c['stream32']['value'] / 1e3), 'S32W': '%.1f' % (c['stream32w']['value'] / 1e3), 'SUR': '%.1f' % (c['surround6']['value'] / 1e3),
}
for f in ('DESIGN.md', 'README.md'):
    s = open(f).read()
    left = set(re.findall(r'@([A-Z0-9]+)@', s))
    for k, v in tok.items():
        s = s.replace('@%s@' % k, v)
    rest = set(re.findall(r'@([A-Z0-9]+)@', s))
    open(f, 'w').write(s)
    print(f, 'filled', sorted(left - rest), 'left', sorted(rest))
