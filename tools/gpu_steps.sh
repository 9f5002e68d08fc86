#!/bin/bash
# Run GPU steps one after another, each under its own timeout; stop at the first step that was KILLED (timeout / signal),
# keep going after ordinary failures.  usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name rc=$rc"; tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "step $name was killed: stopping"; exit $rc; fi
done
exit 0
