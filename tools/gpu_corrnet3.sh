#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "corrnet or deformnet" --timeout=600 2>&1 | tail -2
for rep in 1 2 3; do for v in 1 0; do MORIG_CORRNET_ONE_CSR=$v timeout 600 python bench.py --workload corrnet --steps 40 --warmup 5 --cpu-seconds 0 --secondary 0 --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one_csr=$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done; done
