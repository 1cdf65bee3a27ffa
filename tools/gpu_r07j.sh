#!/bin/bash
# whole GPU suite + smoke on the current library, then the evidence set
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r07j_pytest_gpu.txt 2>&1; echo "suite: $?" > gpurun_out/r07j_rc.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r07j_rc.txt 2>&1
bash tools/gpu_evidence.sh r07j > gpurun_out/r07j_evidence.txt 2>&1
cat gpurun_out/r07j_rc.txt | tail -3; tail -2 gpurun_out/r07j_pytest_gpu.txt; head -3 gpurun_out/r07j_evidence.txt
