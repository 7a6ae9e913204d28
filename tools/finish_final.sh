#!/bin/bash
# tools/finish_final.sh <gpurun tag> <profiles tag>: summaries of tools/collect_final.sh's output under profiles/
set -e
cd "$(dirname "$0")/.."
G=$1; P=$2
python3 tools/make_profile_summary.py $P gpurun_out/${G}_c3/stats gpurun_out/${G}_c3/fetch gpurun_out/${G}_c3/write gpurun_out/$G/bench_default.json \
  "Headline workload (4096 x 4096 x 8 bands fp32), one box. Kernel times from the stats pass (20-step regions under the profiler), HBM traffic from the two PMC passes." > /dev/null
python3 tools/make_profile_summary.py ${P}_c5 gpurun_out/${G}_c5/stats gpurun_out/${G}_c5/fetch gpurun_out/${G}_c5/write gpurun_out/${G}_c5/stats.out \
  "The C5 per-GPU shard (8192 x 8192 x 2 bands fp64, nx_psf = 16384), one box; the bench line quoted below is the PROFILED run's own (20-step regions under rocprofv3)." > /dev/null
python3 tools/make_pd_traffic.py $P gpurun_out/${G}_pd/stats gpurun_out/${G}_pd/fetch gpurun_out/${G}_pd/write | tail -1
python3 tools/make_configs_table.py gpurun_out/$G/bench_default.json "$P: the BASELINE configs from ONE default \`python bench.py\` run (the \"configs\" object of the JSON line; one box)" > profiles/${P}_configs.md
python3 -c "import bench; print(bench.pmc_traffic(4096,8,'f32'), bench.pmc_traffic(8192,2,'f64'), bench.pmc_traffic_pd(2048,4,'f32'))"
