F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
run() { (cd $1 && env $3 python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$2', d['ms_per_step'], d['value'])"); }
for rep in 1 2 3; do
  run _r03 "r03 python + r03 lib" "X=1"
  run _r03 "r03 python + r04 lib" "CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip.so"
  run . "r04 python + r04 lib" "X=1"
  run . "r04, PARAM_PARTS=0 PREP_OVERLAP=0" "CONTOUR_PARAM_PARTS=0 CONTOUR_PREP_OVERLAP=0"
done
