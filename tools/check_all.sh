#!/bin/bash
# build, CPU suite here, then GPU suite + smoke on a gpurun box (what the driver runs at round end)
set -e
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build(); print('built')" | tail -1
timeout 900 python -m pytest tests -x -q -m "not gpu" 2>&1 | tail -2
timeout 3000 /usr/local/graft/bin/gpurun --timeout 1100 -- 'timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/t_gpu.log 2>&1; tail -2 gpurun_out/t_gpu.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1' 2>&1 | tail -3
