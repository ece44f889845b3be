#!/bin/bash
# the shapes outside the headline: a stream with a 777-sample tail, 32-bit input, six channels (value / encode / decode ms)
run() { timeout 900 python3 bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$*: value %.1f ms/step %.3f enc_gpu %.3f dec_gpu %.3f' % (d['value'], d['ms_per_step'], d.get('encode_gpu_ms',0), d.get('decode_gpu_ms',0)))"; }
run --workload stream16
run --workload stream16 --seconds 600.0162
run --workload stream32 --seconds 120
run --workload stream32w --seconds 300
run --workload surround6 --seconds 120
