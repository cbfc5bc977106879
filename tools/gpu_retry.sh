#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> '<command>'  -- retries gpurun while the pod's GPU slots are busy (exit 3 / transient)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  st=$(python3 -c "import json;print(json.load(open('/root/repo/gpurun_out/.last_call.json')).get('status'))" 2>/dev/null)
  if [ "$st" != "transient" ]; then exit $rc; fi
  sleep 45
done
exit 3
