#!/bin/bash
# Runs ON THE GPU BOX: kernel trace (start/end timestamps) of one small PCG bench configuration, reduced to the
# per-iteration timeline tools/trace_gaps.py reads.   tools/trace_gaps.sh <tag> <bench.py args...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; W=/tmp/trace_$TAG
rm -rf "$W"; mkdir -p "$OUT" "$W"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$W" -- python3 "$ROOT/bench.py" --no-cpu --configs none "$@" > "$OUT/run.out" 2> "$OUT/run.err"
echo "rc=$?"
f=$(find "$W" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$OUT/trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
with open(sys.argv[2], 'w') as o:
    o.write('name,start_ns,end_ns\n')
    t0 = int(rows[0]['Start_Timestamp'])
    for r in rows:
        o.write(f"{r['Kernel_Name'].split('(')[0][:70].replace(',', ';')},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0}\n")
print(len(rows), 'kernels')
PY
grep '^{' "$OUT/run.out" | cut -c1-300
