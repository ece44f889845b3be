#!/bin/bash
# kernel-trace stats of tools/dec_only.py: average duration of the decode kernels (ns); optional grep pattern as $1
export TMPDIR=/tmp
rm -rf /tmp/dks; timeout 150 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/dks -o x -- python3 tools/dec_only.py > /dev/null 2>&1
grep -h "${1:-fg_dec_fused\|fg_dec_rice}" /tmp/dks/*kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2, $4}' | tr -d '"'
