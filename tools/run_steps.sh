#!/bin/bash
# Runs the given commands (one per line on stdin: "<seconds> <tag> <command...>") one after the other on the GPU box.
# Every step gets its own `timeout -k 10`; output goes to gpurun_out/<tag>.log; a step that is killed by its timeout or
# by a signal (rc >= 124, SIGABRT's 134 included: that is how a GPU memory fault ends a process), or whose log holds a
# GPU fault, ends the whole run (no further GPU step after a hang or fault).  An ordinary non-zero exit (a failing test) does not.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
while read -r secs tag cmd; do
  [ -z "$tag" ] && continue
  echo "=== $tag: $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$tag.log" 2>&1
  rc=$?
  echo "=== $tag rc=$rc ($(( $(date +%s) - start )) s)"
  tail -n 6 "gpurun_out/$tag.log" | cut -c1-600
  if [ $rc -ge 124 ]; then echo "step $tag was killed (rc=$rc): stopping"; exit $rc; fi
  if grep -qE "Memory access fault|HSA_STATUS_ERROR" "gpurun_out/$tag.log"; then echo "step $tag left a GPU fault in its log: stopping"; exit 70; fi
done
exit 0
