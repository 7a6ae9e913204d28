#!/bin/bash
# Runs ON THE GPU BOX: the end-of-round record for the three profiled workloads (headline C3, C4 primal-dual, C5 shard) --
# kernel stats + the two PMC passes each -- and one default bench.py run.   tools/collect_final.sh <tag>
# Then, in the build container:  tools/finish_final.sh <tag>   (summaries under profiles/, stamped with commit + source hash)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/$TAG"
bash "$ROOT/tools/collect_profile.sh" ${TAG}_c3 > "$ROOT/gpurun_out/$TAG/collect_c3.log" 2>&1
bash "$ROOT/tools/collect_profile.sh" ${TAG}_pd --workload pd > "$ROOT/gpurun_out/$TAG/collect_pd.log" 2>&1
bash "$ROOT/tools/collect_profile.sh" ${TAG}_c5 --size 8192 --bands 2 --dtype f64 > "$ROOT/gpurun_out/$TAG/collect_c5.log" 2>&1
cd "$ROOT" && timeout -k 10 300 python bench.py > "gpurun_out/$TAG/bench_default.json" 2> "gpurun_out/$TAG/bench_default.err"
tail -c 300 "gpurun_out/$TAG/bench_default.json"
