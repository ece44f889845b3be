for stop in 0 1 2 3; do
  if [ $stop = 0 ]; then unset FLACGPU_STOP; else export FLACGPU_STOP=$stop; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/x$stop -o x -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "STOP=$stop"; grep -h "autoc" gpurun_out/x$stop/*kernel_stats.csv | cut -d, -f2-4
done
