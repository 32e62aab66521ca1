#!/bin/bash
# HIP-graph replay of the step under the runtime's graph switches
mkdir -p gpurun_out
for v in "" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=1"; do
  echo "=== [$v]"
  env $v timeout 300 python tools/exp_capture.py 2>&1 | grep -E "ms/step|captured|Error|error" 
done
