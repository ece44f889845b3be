#!/bin/bash
# Round-6 evidence run (GPU box), one gpurun call, most important first; every step under its own `timeout`, every log kept.
#   bash tools/r06_collect.sh <tag> [phases]      phases: any of t (tests) b (bench line) f (fuzzers) p (profiles), default "tbfp"
# Everything lands under gpurun_out/r06<tag>/ (+ gpurun_out/{prof_r06<tag>,cfg6_*,pmc6_*}); tools/r06_profiles.sh copies the
# summaries into profiles/ (run it on the build box).  The ids of the library that ran are in build_id.txt ("build kernel host").
export TMPDIR=/tmp
TAG=${1:-a}
PH=${2:-tbfp}
FUZZ=${FUZZ_SEEDS:-300000}
OUT=gpurun_out/r06$TAG
mkdir -p $OUT
python3 -c "from pyflac_amd import _lib; L = _lib.lib(); print(L.flacgpu_build_id().decode(), L.flacgpu_kernel_id().decode(), L.flacgpu_host_id().decode())" > $OUT/build_id.txt 2>/dev/null
if [[ $PH == *t* ]]; then
  # (1) the whole GPU suite on this tree, log kept; then smoke()
  timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; echo "pytest -m gpu: exit $?" >> $OUT/gpu_tests.log
  tail -3 $OUT/gpu_tests.log
  timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke: exit $?" >> $OUT/smoke.log
  tail -2 $OUT/smoke.log
fi
if [[ $PH == *b* ]]; then
  # (2) the driver-shaped bench line (defaults: N = 1, the other configs in the same process)
  timeout 1500 python3 bench.py > $OUT/bench_headline.log 2> $OUT/bench_headline.err; grep -v "^[WEI]2026" $OUT/bench_headline.log | tail -1 > $OUT/bench_headline.json
  tail -c 600 $OUT/bench_headline.json; echo
  # two ranks on the one GPU (gloo): the N > 1 line with its cpu_baseline
  timeout 900 python3 bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --streams 8 --seconds 20 > $OUT/bench_2ranks.log 2> $OUT/bench_2ranks.err
  grep -v "^[WEI]2026" $OUT/bench_2ranks.log | tail -1 > $OUT/bench_2ranks.json
  # configs[3], A/B of round 6's direct packing of 24-bit frames against the chunk form (written with no GPU at hand: this run decides)
  for rep in 1 2; do
    for ab in direct chunk; do
      flag=""; [[ $ab == chunk ]] && flag="--no-direct"
      timeout 600 python3 bench.py --workload stream24 --steps 100 --no-cpu-baseline --no-e2e --no-passes $flag 2> /dev/null | grep -v "^[WEI]2026" | tail -1 > $OUT/bench_stream24_${ab}_$rep.json
      python3 -c "import json,sys; d=json.load(open('$OUT/bench_stream24_${ab}_$rep.json')); print('stream24 $ab', d.get('ms_per_step'), d.get('encode_gpu_ms'), d.get('value'))" 2>/dev/null
    done
  done
fi
if [[ $PH == *f* ]]; then
  # (3) the differential fuzzers against the oracle
  timeout 3000 python3 tests/tools/gpu_fuzz.py 6000000 $FUZZ > $OUT/fuzz_gpu.log 2>&1; tail -1 $OUT/fuzz_gpu.log
  timeout 600 python3 tests/tools/gpu_api_fuzz.py 6000000 3000 > $OUT/fuzz_api.log 2>&1; tail -1 $OUT/fuzz_api.log
  timeout 600 python3 tests/tools/gpu_damage_fuzz.py 6000000 8000 > $OUT/fuzz_damage.log 2>&1; tail -1 $OUT/fuzz_damage.log
  timeout 600 python3 tests/tools/dec_stream_fuzz.py gpu 6000000 3000 > $OUT/fuzz_dec_stream.log 2>&1; tail -1 $OUT/fuzz_dec_stream.log
  timeout 600 python3 tests/tools/gpu_batch_fuzz.py 6000000 4000 > $OUT/fuzz_batch.log 2>&1; tail -1 $OUT/fuzz_batch.log
  timeout 900 python3 tests/tools/fuzz32.py 6000000 150000 > $OUT/fuzz32.log 2>&1; tail -1 $OUT/fuzz32.log
fi
if [[ $PH == *p* ]]; then
  # (4) rocprofv3 kernel stats + the counter passes (separate runs, --kernel-trace only beside --pmc) of the headline command, of
  # configs[3] and of configs[4]'s share; the small-call probe
  bash tools/rocprof_run.sh r06$TAG 600 5 stream16 --no-passes > $OUT/rocprof_run.log 2>&1      # (no event passes: they run the stages one by one, i.e. unfused)
  for spec in "stream24 300 8" "batch 60 5"; do
    set -- $spec
    O2=$PWD/gpurun_out/cfg6_$1
    mkdir -p $O2
    rocprofv3 --output-format csv --kernel-trace --stats -d $O2 -o t -- python3 bench.py --workload $1 --seconds $2 --level $3 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-configs --no-passes > $O2/bench.log 2>&1
    grep -v "^[WEI]2026" $O2/bench.log | tail -1 > $O2/bench.json
    O3=$PWD/gpurun_out/pmc6_$1
    mkdir -p $O3
    CMD="python3 bench.py --workload $1 --seconds $2 --level $3 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-configs --no-passes"
    rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O3/pmc1 -o pmc1 -- $CMD > $O3/bench_pmc1.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O3/pmc3 -o pmc3 -- $CMD > $O3/bench_pmc3.log 2>&1
    rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $O3/pmc4 -o pmc4 -- $CMD > $O3/bench_pmc4.log 2>&1
  done
  (python3 tools/exp/process_call_probe.py 4096; python3 tools/exp/process_call_probe.py 1024; python3 tools/exp/process_call_probe.py 16384; python3 tools/exp/small_call_probe.py) > $OUT/small_calls.txt 2>&1
  # (round 6's fg_pipe_autoc1_kernel against the wave-a-block kernel on the same one-block calls: written with no GPU at hand, this decides)
  (echo "== FLACGPU_AUTOC1=0: fg_pipe_autoc_kernel on the same calls (test-hooks library)"; PYFLAC_AMD_TESTHOOKS=1 FLACGPU_AUTOC1=0 python3 tools/exp/process_call_probe.py 4096;
   echo "== release library once more"; python3 tools/exp/process_call_probe.py 4096) >> $OUT/small_calls.txt 2>&1
fi
ls -la $OUT
