#!/bin/bash
# usage: tools/gpu_step.sh <seconds> <logfile> <command...>
# Runs one GPU step under its own timeout; a step that TIMES OUT or is KILLED stops the whole batch
# (exit 99: callers chain steps with &&), any other failure is logged and the batch goes on.
secs=$1; log=$2; shift 2
mkdir -p "$(dirname "$log")"
timeout -k 10 "$secs" "$@" > "$log" 2> "${log%.*}.err"
rc=$?
echo "[$(date +%H:%M:%S)] rc=$rc : $*" | tee -a "$(dirname "$log")/steps.log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 99; fi
exit 0
