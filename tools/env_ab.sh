# usage: bash tools/env_ab.sh ENVVAR "vals..." ; alternating bench runs on one box
V=$1; shift
F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
for rep in 1 2 3; do for val in $@; do
  env $V=$val python bench.py $F > gpurun_out/_ab.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/_ab.json').read().strip().splitlines()[-1]);print('$V=$val', d['ms_per_step'], d['value'])"
done; done
