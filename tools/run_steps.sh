#!/bin/bash
# Runs the given commands (one per line on stdin: "<seconds> <tag> <command...>") one after the other on the GPU box.
# Every step gets its own `timeout -k 10`; output goes to gpurun_out/<tag>.log; a step that is killed by its timeout or
# by a signal ends the whole run (no further GPU step after a hang).  An ordinary non-zero exit (a failing test) does not.
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
  if [ $rc -ge 124 ] && [ $rc -ne 134 ]; then echo "step $tag was killed (rc=$rc): stopping"; exit $rc; fi
done
exit 0
