# stream priorities inside the training step (bench.py --steps 200 --warmup 30, alternating): main stream / weight-gradient stream
F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { env $2 python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$1', d['ms_per_step'], d['value'])"; }
for rep in 1 2 3 4; do
  run "default (both 0)" "X=1"
  run "main -1 (high)" "CONTOUR_MAIN_PRIORITY=-1"
  run "side -1 (high)" "CONTOUR_SIDE_PRIORITY=-1"
done
