# Evidence batch of round 3 (run on the MI355X box through gpurun): files land in gpurun_out/, the ones to keep are copied to profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
echo "default bench done"
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_r03 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_bench_under_rocprof.err)
python3 tools/kernel_trace_by_shape.py $(ls gpurun_out/prof_r03/*/*_kernel_trace.csv | head -1) regrid_cols_ell_direct_kernel > gpurun_out/r03_bench_kernel_by_launch_shape.csv
cp $(ls gpurun_out/prof_r03/*/*_kernel_stats.csv | head -1) gpurun_out/r03_bench_kernel_stats.csv
echo "rocprof done"
python3 tools/kernel_bench.py --out gpurun_out/r03_kernel_bench.json > gpurun_out/r03_kernel_bench.log 2>&1
echo "kernel bench done"
python3 tools/level_sweep.py > gpurun_out/r03_level_sweep.log 2>&1
echo "level sweep done"
python3 tools/api_bench.py > gpurun_out/r03_api_bench.log 2>&1 || true
echo "api bench done"
