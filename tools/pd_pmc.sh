#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the primal-dual iteration's kernels (two passes)
set -u
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pdpmc; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d /tmp/pdp1 -- python3 $ROOT/bench.py --workload pd --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 > $OUT/p1.out 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pdp2 -- python3 $ROOT/bench.py --workload pd --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 > $OUT/p2.out 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/pdp3 -- python3 $ROOT/bench.py --workload pd --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 > $OUT/p3.out 2>&1
for p in pdp1 pdp2 pdp3; do f=$(find /tmp/$p -name '*counter_collection.csv' | head -1); if [ -n "$f" ]; then (head -1 "$f"; grep 'pfb::k_dwt\|pfb::k_idwt\|pfb::k_dual\|pfb::k_pd_' "$f") > $OUT/$p.csv; fi; done
ls -la $OUT
