#!/bin/bash
# Runs the given commands one after the other on the GPU box, each under its own `timeout -k 10 LIMIT`; stops at the first one that
# was killed at its limit (a hung kernel: nothing further may touch the GPU), but continues past ordinary failures.
#   tools/gpu_step.sh OUTDIR LIMIT 'cmd 1' 'cmd 2' ...      (stdout+stderr of command i -> OUTDIR/step<i>.log)
out="$1"; limit="$2"; shift 2
mkdir -p "$out"
i=0
worst=0
for c in "$@"; do
  i=$((i + 1))
  echo "== step $i: $c"
  timeout -k 10 "$limit" bash -c "$c" > "$out/step$i.log" 2>&1
  rc=$?
  echo "   rc=$rc"
  tail -n 25 "$out/step$i.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
  [ $rc -ne 0 ] && worst=$rc
done
exit $worst
