#!/bin/bash
# gpurun wrapper: waits for a free box (gpurun exit code 3 = none free, nothing ran), then runs the command once.
# usage: tools/gr.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
