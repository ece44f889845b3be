#!/bin/bash
# several settings of one environment variable on the headline bench, same box, interleaved: bash tools/ab_multi.sh VAR v1 v2 v3 ... [-- extra bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for rep in 1 2; do for v in "${VALS[@]}"; do
  env $VAR=$v python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-passes "$@" 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=r['timed_region_gpu_ms']; print('%-24s value %8.1f  step %.3f ms  encode %.3f  decode %.3f' % (sys.argv[1], r['value'], r['ms_per_step'], t['encode'], t['decode']))" "$VAR=$v"
done; done
