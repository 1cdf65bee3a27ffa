timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fps or point or ball or knn" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q -x -k "corrnet or point_modules or deformnet" 2>&1 | tail -4
for i in 1 2; do for v in 0 1; do
MORIG_FPS_BKT=$v python bench.py --workload corrnet --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('BKT=$v', r['value'], r.get('ms_per_step_median'), ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:7]))"
done; done
