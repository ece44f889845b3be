#!/bin/bash
# A/B of one bench.py flag on the headline bench, same box, alternating runs: bash tools/ab_flag.sh --no-direct [runs] [extra bench args]
F=$1; N=${2:-3}; shift; shift
one() { python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-passes "$@" 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=r['timed_region_gpu_ms']; print('%-28s value %8.1f  step %.3f ms  encode %.3f  decode %.3f' % (sys.argv[1], r['value'], r['ms_per_step'], t['encode'], t['decode']))" "${LABEL}"; }
for i in $(seq $N); do LABEL=base one "$@"; LABEL="$F" one $F "$@"; done
