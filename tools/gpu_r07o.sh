#!/bin/bash
# network-level tests under the other settings of this round's switches: exact arithmetic, every layer row-normalised, unpaired boundary passes
cd /root/repo; mkdir -p gpurun_out; : > gpurun_out/r07o_rc.txt
MORIG_PRECISION=f32 timeout 1500 python -m pytest tests/test_gpu_networks.py -m gpu -q > gpurun_out/r07o_f32.txt 2>&1; echo "f32: $? $(tail -1 gpurun_out/r07o_f32.txt)" >> gpurun_out/r07o_rc.txt
MORIG_PACK_NORMALISE=all timeout 1500 python -m pytest tests/test_gpu_networks.py -m gpu -q > gpurun_out/r07o_norm_all.txt 2>&1; echo "normalise all: $? $(tail -1 gpurun_out/r07o_norm_all.txt)" >> gpurun_out/r07o_rc.txt
MORIG_EDGE_PAIR=0 MORIG_DMA_SMALL_TILES=0 MORIG_NO_FEW_ROWS=1 MORIG_ATTN_LDS=1 MORIG_X3_TILE=1 timeout 1500 python -m pytest tests/test_gpu_networks.py -m gpu -q > gpurun_out/r07o_old_paths.txt 2>&1; echo "old paths: $? $(tail -1 gpurun_out/r07o_old_paths.txt)" >> gpurun_out/r07o_rc.txt
cat gpurun_out/r07o_rc.txt
