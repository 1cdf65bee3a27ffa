#!/bin/bash
cd /root/repo
bash tools/gpu_timeline.sh r07d corrnet fps_bkt_kernel\<8 > /dev/null 2>&1
bash tools/gpu_timeline.sh r07d mask_skin cls_attention > /dev/null 2>&1
head -2 gpurun_out/timeline_corrnet_r07d.txt gpurun_out/timeline_mask_skin_r07d.txt
