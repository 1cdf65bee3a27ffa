#!/bin/bash
# whole GPU suite + smoke on the current library, then the evidence set (usage: gpu_final_check.sh [tag])
TAG=${1:-r07j}; cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "suite: $?" > gpurun_out/${TAG}_rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/${TAG}_rc.txt 2>&1
bash tools/gpu_evidence.sh ${TAG} > gpurun_out/${TAG}_evidence.txt 2>&1
cat gpurun_out/${TAG}_rc.txt | tail -3; tail -2 gpurun_out/${TAG}_pytest_gpu.txt; head -3 gpurun_out/${TAG}_evidence.txt
