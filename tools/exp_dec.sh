for skip in 0 1 2 3; do
  export FLACGPU_DEC_SKIP=$skip
  rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/ds$skip -o x -- python3 tools/dec_only.py > /dev/null 2>&1
  echo "SKIP=$skip"; grep -h "fused" gpurun_out/ds$skip/*kernel_stats.csv | awk -F'",' '{print $2,$3,$4}'
done
