F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
L=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so
run() { env CONTOUR_HIP_LIB=$L $2 python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3 4 5 6; do
  run "tuning lib" "X=1"
  run "tuning lib, CU_NORM_NO_SMALL_RES" "CU_NORM_NO_SMALL_RES=1"
done
