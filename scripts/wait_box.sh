#!/bin/bash
# Poll `gpurun --status` until the in-flight call's elapsed_s RESETS (the call left the queue: the box has started and
# the repository snapshot is being pushed), then wait 45 s more.  Edits to the tree are safe after this returns.
prev=-1
for i in $(seq 1 400); do
  e=$(/usr/local/graft/bin/gpurun --status 2>&1 | grep -o '"elapsed_s": [0-9.]*' | grep -o '[0-9.]*$')
  if [ -z "$e" ]; then echo "no call in flight"; exit 0; fi
  ei=${e%.*}
  if [ "$prev" -ge 0 ] && [ "$ei" -lt "$prev" ]; then echo "box started (elapsed reset $prev -> $ei)"; sleep 45; exit 0; fi
  prev=$ei
  sleep 15
done
echo "gave up"
