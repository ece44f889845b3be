for fbw in 1280 880 760; do
  export FLACGPU_FBW=$fbw
  rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/fbw$fbw -o x -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  echo "FBW=$fbw"; grep -h "pack_kernel<true, 2, 8, false, 2>" gpurun_out/fbw$fbw/*kernel_stats.csv | awk -F'",' '{print $2,$3,$4}'
done
