#!/bin/bash
# the three secondary workloads with the wave decoder and with the fused kernel of round 2 (decode_gpu_ms / value)
for wl in stream24 batch wasted; do
  for wv in 1 0; do
    FLACGPU_DEC_WAVE=$wv timeout 600 python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$wl wave=$wv value %.1f ms/step %.3f enc_gpu %.3f dec_gpu %.3f' % (d['value'], d['ms_per_step'], d.get('encode_gpu_ms',0), d.get('decode_gpu_ms',0)))"
  done
done
