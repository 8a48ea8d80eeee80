F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
run() { (cd _r03 && env $2 python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['value'])"); }
L=$PWD/contouring-uncertainty_amd
for rep in 1 2 3; do
  run "r03 lib" "X=1"
  run "r04 lib" "CONTOUR_HIP_LIB=$L/libcontour_hip.so"
  for v in igemm_conv norm misc igemm_wgrad; do run "r04 lib with r03 $v" "CONTOUR_HIP_LIB=$L/libv_$v.so"; done
done
