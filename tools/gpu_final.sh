#!/bin/bash
# end-of-round validation in ONE call: full -m gpu suite + smoke + default bench (gpu_full.sh), the evidence set (gpu_evidence.sh), the glue launch counts
TAG=${1:-final}
bash tools/gpu_full.sh $TAG
bash tools/gpu_evidence.sh $TAG
bash tools/gpu_glue_counts.sh $TAG
