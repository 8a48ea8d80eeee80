set -e
F="--no-cpu-baseline --no-parity --no-roofline --graph off --steps 300 --warmup 30"
python bench.py --batch 64 $F > gpurun_out/tp_b64.json 2>/dev/null
python bench.py --batch 32 $F > gpurun_out/tp_b32.json 2>/dev/null
python bench.py --batch 32 $F > gpurun_out/tp_b32_a.json 2>/dev/null &
P1=$!
python bench.py --batch 32 $F > gpurun_out/tp_b32_b.json 2>/dev/null &
P2=$!
wait $P1; wait $P2
python - <<'PY'
import json
for n in ("tp_b64","tp_b32","tp_b32_a","tp_b32_b"):
    d=json.loads(open(f"gpurun_out/{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"])
PY
