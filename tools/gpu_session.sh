#!/bin/bash
# Runs a list of GPU steps on the box, each under its own timeout; stops the whole session after a step that was
# killed at its limit (never start another GPU step after a hang).  Usage: tools/gpu_session.sh <outdir> <<< "name|seconds|command"
out=$1; mkdir -p "$out"; export TMPDIR=/tmp
while IFS='|' read -r name secs cmd; do
  [ -z "$name" ] && continue
  echo "=== $name (limit ${secs}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$secs" bash -c "$cmd" > "$out/$name.out" 2> "$out/$name.err"
  rc=$?
  echo "=== $name rc=$rc after $(( $(date +%s) - start ))s"
  tail -c 1500 "$out/$name.out"; tail -c 600 "$out/$name.err"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 3 ]; then echo "step $name was killed at its limit or reported a hung exchange (rc $rc): stopping the session"; exit $rc; fi
done
exit 0
