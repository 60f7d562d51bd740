#!/bin/bash
# Round 6, last closing call: PMC / kernel-stat passes of the step, the bench line that quotes them, the one-rank point of the N > 1 workload, then smoke() and the whole GPU tier.
#   gpurun --timeout 3600 -- 'bash tools/gpu_calls_r06/r6_final4.sh'
bash tools/gpu_calls_r06/r6_final3.sh
bash tools/gpu_calls_r06/r6_call19.sh
