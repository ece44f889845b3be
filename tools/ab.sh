#!/bin/bash
# A/B of one environment switch on the headline bench, same box, alternating runs: bash tools/ab.sh VAR=value [runs]
# prints value, ms per step and the device-stamp times of the encode and the decode launch
V=$1; N=${2:-3}
one() { python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-passes 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=r['timed_region_gpu_ms']; print('%-28s value %8.1f  step %.3f ms  encode %.3f  decode %.3f' % (sys.argv[1], r['value'], r['ms_per_step'], t['encode'], t['decode']))" "$1"; }
for i in $(seq $N); do one "base"; env $V bash -c "$(declare -f one); one '$V'"; done
