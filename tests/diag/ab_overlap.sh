for v in 1 0 1 0; do NERF_OVERLAP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('overlap', os.environ.get('NERF_OVERLAP'), round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})" || exit 1; done
