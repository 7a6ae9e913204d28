#!/bin/bash
# Runs ON THE GPU BOX: average duration of k_pcg_update_dir (rocprofv3 --stats) for the default library and a variant
# tools/ab_update_kernel.sh <variant .so> [bench args]
set -u
V=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for rnd in 1 2 3; do for lib in base var; do
  if [ $lib = var ]; then export PFB_HIP_LIB=$V; else unset PFB_HIP_LIB; fi
  rm -rf /tmp/abu; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abu -- python3 $ROOT/bench.py --no-cpu --configs none --steps 20 --warmup 3 --repeats 2 "$@" > /dev/null 2>&1
  f=$(find /tmp/abu -name '*kernel_stats.csv' | head -1)
  echo "$lib: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_pcg_update_dir' in r['Name']: print('calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3, 2)); break
")"
done; done
