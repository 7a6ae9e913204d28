cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdprof -- python3 $GRAFT_REPO_ROOT/bench.py --workload pd --no-cpu --steps 20 --warmup 3 --repeats 2 > /tmp/pd.out 2> /tmp/pd.err; python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pdprof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:72].ljust(72), r['Calls'], r['AverageNs'], r['Percentage'])
PY
