one() { python3 bench.py --no-cpu-baseline --no-e2e --no-configs --no-passes 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=r['timed_region_gpu_ms']; print('%-6s decode %.3f encode %.3f value %.0f' % (sys.argv[1], t['decode'], t['encode'], r['value']))" $1; }
for i in 1 2 3; do
  one r5
  FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_r3.so one r3
  FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_r4.so one r4
done
