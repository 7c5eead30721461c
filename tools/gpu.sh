#!/bin/bash
# gpurun with a wait-and-retry when no box / slot is free (exit code 3: nothing was charged).  usage: tools/gpu.sh TIMEOUT 'command'
t=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
