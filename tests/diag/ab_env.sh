# usage: ab_env.sh VAR "bench args": runs bench.py alternately with VAR unset and VAR=1
VAR=$1; shift
for v in 0 1 0 1; do env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 $@ 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value']), round(d['ms_per_step'],2), {k:round(v['ms_per_step'],3) for k,v in (d.get('kernels_serialized') or d['kernels']).items()})" || exit 1; done
