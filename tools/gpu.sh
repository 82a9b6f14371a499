#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3 = nothing ran, nothing charged).  usage: tools/gpu.sh TIMEOUT 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
