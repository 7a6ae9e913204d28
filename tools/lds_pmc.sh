set -u
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
mkdir -p $ROOT/gpurun_out/ldspmc
rocprofv3 -L 2>/dev/null | grep -i -E "LDS|SQ_WAIT_INST|SQ_INSTS_VALU\b|SQ_ACTIVE_INST_VALU|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INST_CYCLES_VMEM|SQ_WAIT_ANY" | cut -c1-200 > $ROOT/gpurun_out/ldspmc/avail.txt
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/ldsp1 -- python3 $ROOT/bench.py --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 > $ROOT/gpurun_out/ldspmc/p1.out 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/ldsp2 -- python3 $ROOT/bench.py --no-cpu --configs none --steps 4 --warmup 1 --repeats 1 > $ROOT/gpurun_out/ldspmc/p2.out 2>&1
for p in ldsp1 ldsp2; do f=$(find /tmp/$p -name '*counter_collection.csv' | head -1); if [ -n "$f" ]; then (head -1 "$f"; grep 'pfb::k_row\|pfb::k_col' "$f") > $ROOT/gpurun_out/ldspmc/$p.csv; fi; done
ls -la $ROOT/gpurun_out/ldspmc
