# usage: bash tools/env_ab_tuning.sh ENVVAR "vals..." ; alternating bench runs on one box, TUNING build (library knobs honoured)
V=$1; shift
F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
export CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so
for rep in 1 2 3 4 5; do for val in $@; do
  env $V=$val python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$V=$val', d['ms_per_step'], d['value'])"
done; done
