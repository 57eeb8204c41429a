#!/bin/bash
# Runs the GPU steps of one gpurun call in sequence: tools/gpu_seq.sh OUTDIR "name|timeout_s|command" ...
# Each step's stdout+stderr goes to OUTDIR/name.log.  An ordinary failure (a test that fails) does not stop the sequence, a step
# that had to be killed at its time limit (rc 124 / 137) does: nothing more is started on a GPU that may be hung.
out=$1; shift
mkdir -p "$out"
for step in "$@"; do
  name=${step%%|*}; rest=${step#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "$out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start ))s"; tail -n 3 "$out/$name.log"
  echo "$name rc=$rc" >> "$out/_steps.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== $name hit its time limit: stopping here"; exit 1; fi
done
exit 0
