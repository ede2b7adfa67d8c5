#!/bin/bash
# tools/gpurun_retry.sh <timeout-s> <command...> — gpurun, retried while the pool says "no slot / no box" (exit code 3)
T=$1; shift
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
