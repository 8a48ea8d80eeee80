# same-box A/B of two builds of the repository: bash tools/build_ab.sh <other checkout dir>   (bench.py of each, alternating)
O=$1
F="--no-cpu-baseline --no-parity --no-roofline --steps 200 --warmup 30"
for rep in 1 2 3; do
  for d in ${ORDER:-$O .}; do
    (cd $d && python bench.py $F 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$d', d['ms_per_step'], d['value'])")
  done
done
