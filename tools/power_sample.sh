#!/bin/bash
# Runs ON THE GPU BOX: samples rocm-smi (power, sclk, mclk) while the headline bench loops (read-only queries)
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu --configs none --steps 600 --warmup 5 --repeats 6 > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level|fclk" | tr '\n' ';' ; echo
  sleep 0.5
done
wait $BP
tail -c 300 gpurun_out/power_bench.json
