#!/bin/bash
# gpurun with retries while no GPU slot / box is free (exit code 3: nothing charged).
# usage: scripts/gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry $i] no slot free, sleeping 60 s"
  sleep 60
done
exit 3
