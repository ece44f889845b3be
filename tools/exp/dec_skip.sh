#!/bin/bash
# what the restore kernel's time consists of: the decode launch with its output (1), its recurrence (2) or both (3) left out
# (tuning build: bash tools/build_variant2.sh tun "" TUNING=1), then the kernel timeline of one launch.  bash tools/exp/dec_skip.sh
export FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_tun.so
for i in 1 2; do
  for s in 0 1 2 3; do echo -n "skip$s "; FLACGPU_DEC_SKIP=$s python3 tools/exp/dec_time.py 600 30 2>/dev/null; done
done
export TMPDIR=/tmp
for s in 0 1 2 3; do
  export FLACGPU_DEC_SKIP=$s
  OUT=$PWD/gpurun_out/kstat_decskip$s; mkdir -p $OUT
  (cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/exp/dec_time.py 600 20 > $OUT/log.txt 2>&1)
  echo "== skip $s"
  python3 - "$OUT" <<'P'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_stats.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name']
        if 'fg_dec' not in n: continue
        short = n[:n.index('>') + 1] if '<' in n else n.split('(')[0]
        print('%-60s calls %4s  avg %9.1f us' % (short.replace('void ', '').replace('(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1000))
P
done
unset FLACGPU_DEC_SKIP
