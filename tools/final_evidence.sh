# Round evidence in one GPU call: kernel-trace statistics (two streams / one stream), PMC HBM traffic (separate passes), per-launch
# table.  Usage on the GPU box: bash tools/final_evidence.sh r04   -> files under gpurun_out/, to be copied into profiles/
set -e
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev_two -- python3 $B > gpurun_out/ev_two.log 2>&1
cp $(ls gpurun_out/ev_two/*/*kernel_stats.csv | head -1) gpurun_out/${R}_kernel_stats_bench_b64_two_streams.csv
python tools/trace_gaps.py $(ls gpurun_out/ev_two/*/*kernel_trace.csv | head -1) > gpurun_out/${R}_timeline_gaps.txt 2>&1 || true
echo "two-stream trace done"
export CONTOUR_SIDE_WGRAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ev_one -- python3 $B > gpurun_out/ev_one.log 2>&1
unset CONTOUR_SIDE_WGRAD
cp $(ls gpurun_out/ev_one/*/*kernel_stats.csv | head -1) gpurun_out/${R}_kernel_stats_bench_b64_one_stream.csv
echo "one-stream trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ev_pf -- python3 $B --no-roofline > gpurun_out/ev_pf.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/ev_pw -- python3 $B --no-roofline > gpurun_out/ev_pw.log 2>&1
echo "write pass done"
python tools/pmc_summary.py gpurun_out/ev_pf gpurun_out/ev_pw 10 gpurun_out/${R}_pmc_hbm_traffic.json
CONTOUR_SIDE_WGRAD=0 python tools/layer_profile.py 64 > gpurun_out/${R}_layers_one_stream.txt 2>&1
python tools/layer_profile.py 64 > gpurun_out/${R}_layers_two_streams.txt 2>&1
echo "layer tables done"
