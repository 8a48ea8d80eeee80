for v in 0 1 0 1; do
  CONTOUR_HIP_LIB=$PWD/contouring-uncertainty_amd/libcontour_hip_tuning.so CU_PSM_MERGED_WINDOW=$v python bench.py --workload c5 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/_c5.json 2>gpurun_out/_c5.err
  python -c "
import json;d=json.loads(open('gpurun_out/_c5.json').read().strip().splitlines()[-1]);print('merged_window=$v', d['value'], d['ms_per_step'], d['config'].get('skew_psm_plus_masks_plus_entropy_frames_per_s'))"
done
