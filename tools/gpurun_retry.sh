#!/bin/bash
# tools/gpurun_retry.sh <timeout seconds> '<command>': gpurun, retried while the pool has no free slot (exit code 3: nothing was
# charged).  Any other outcome is returned as is -- a command that ran is never run again.
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
