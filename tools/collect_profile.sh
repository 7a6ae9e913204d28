#!/bin/bash
# Runs ON THE GPU BOX: the three rocprofv3 passes behind a profiles/<tag>_* summary (kernel trace + stats, and the
# FETCH_SIZE / WRITE_SIZE counter passes in runs of their own), reduced to the few small CSVs
# tools/make_profile_summary.py reads (raw rocprofv3 output is tens of MB; gpurun merges at most 64 MiB back).
#   tools/collect_profile.sh <tag> [bench.py args...]        -> gpurun_out/<tag>/{stats,fetch,write}/...csv
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
W=/tmp/prof_$TAG
rm -rf "$W"; mkdir -p "$OUT/stats" "$OUT/fetch" "$OUT/write" "$W"
cd /tmp && export TMPDIR=/tmp
run() {   # name, seconds, command...
    local name=$1 secs=$2; shift 2
    timeout -k 10 "$secs" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    echo "[$(date +%H:%M:%S)] rc=$rc $name" | tee -a "$OUT/steps.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
}
run stats 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/stats" -- python3 "$ROOT/bench.py" --no-cpu --configs none --steps 20 --warmup 5 --repeats 2 "$@"
run fetch 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$W/fetch" -- python3 "$ROOT/bench.py" --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 "$@"
run write 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$W/write" -- python3 "$ROOT/bench.py" --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 "$@"
find "$W/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/stats/bench_kernel_stats.csv" \;
for p in fetch write; do
    f=$(find "$W/$p" -name '*counter_collection.csv' | head -1)
    if [ -n "$f" ]; then (head -1 "$f"; grep 'pfb::' "$f") > "$OUT/$p/bench_counter_collection.csv"; fi
done
ls -la "$OUT" "$OUT/stats" "$OUT/fetch" "$OUT/write"
du -sh "$W"
