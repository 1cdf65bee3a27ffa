#!/bin/bash
# r03 second call: the new suites (geo graph, serving loop), then bench sync vs deferred guard, then small-batch with the captured graph
mkdir -p gpurun_out
TAG=${1:-r03b}
timeout 900 python -m pytest tests/test_geo_graph.py -q -m gpu --timeout=600 2>&1 | tail -25 > gpurun_out/pytest_geo_$TAG.txt; tail -25 gpurun_out/pytest_geo_$TAG.txt
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu --timeout=600 -k "async or captured or out_of_range or batched_equals" 2>&1 | tail -25 > gpurun_out/pytest_serving_$TAG.txt; tail -25 gpurun_out/pytest_serving_$TAG.txt
for g in sync deferred sync deferred; do
  MORIG_BENCH_GUARD=$g python bench.py --secondary 0 --cpu-seconds 0 --prof-steps 0 --steps 40 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$g', r['value'], r['ms_per_step'], r['ms_per_step_median'])"
done | tee gpurun_out/guard_ab_$TAG.txt
timeout 600 python tools/small_batch.py 1 2 4 8 graph > gpurun_out/small_batch_$TAG.txt 2>&1; tail -30 gpurun_out/small_batch_$TAG.txt
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python -c "
import json; r=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step']); print(json.dumps(r['secondary'], indent=0)[:3000])"; tail -3 gpurun_out/bench_$TAG.err
