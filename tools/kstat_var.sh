#!/bin/bash
# per-kernel averages of the headline loop for several library variants, one group (kernels do not overlap): bash tools/kstat_var.sh v1 v2 ...
# FLACGPU_GROUPS is a kernel selector (fg_types.h fg_sel): only a test-hooks or a tuning build reads it.  "head" therefore runs the
# tree's test-hooks library (PYFLAC_AMD_TESTHOOKS=1: same kernel objects as the release library); a variant must have been built
# with tools/build_variant2.sh ... TUNING=1 -- pyflac_amd/_lib.py says so on stderr when a release build is handed a selector.
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "head" ]; then unset FLACGPU_LIBRARY FLACGPU_ALLOW_LIBRARY_OVERRIDE; export PYFLAC_AMD_TESTHOOKS=1
  else unset PYFLAC_AMD_TESTHOOKS; export FLACGPU_ALLOW_LIBRARY_OVERRIDE=1 FLACGPU_LIBRARY=$PWD/gpurun_exp/libflacgpu_$v.so; fi
  FLACGPU_GROUPS=1 bash tools/kstat.sh $v 2>&1 | grep -i "pyflac_amd:\|pack\|eval\|autoc\|levin\|assem\|scan_sizes"
  FLACGPU_GROUPS=1 python3 tools/exp/direct_time.py 600 40 1 | grep -v amdgpu
  python3 tools/exp/direct_time.py 600 40 1 | grep -v amdgpu
done
