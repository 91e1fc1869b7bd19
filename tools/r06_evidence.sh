# Evidence batch of round 6 (run on the MI355X box through gpurun): files land in gpurun_out/, the ones to keep are copied to profiles/.
#   bash tools/r06_evidence.sh [bench|pmc]      (round 6 changed no kernel's memory access: the PMC part re-measures only if asked)
set -e
R=$GRAFT_REPO_ROOT
cd $R
WHAT=${1:-all}
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  python3 tools/kernel_bench.py --out gpurun_out/r06_kernel_bench.json > gpurun_out/r06_kernel_bench.log 2>&1
  echo "kernel bench done"
  python3 bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
  echo "default bench done"
  (cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_r06 && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r06 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r06_bench_under_rocprof.json 2> $R/gpurun_out/r06_bench_under_rocprof.err)
  python3 tools/kernel_trace_by_shape.py $(ls -t gpurun_out/prof_r06/*/*_kernel_trace.csv | head -1) regrid_cols_ell_direct_kernel > gpurun_out/r06_bench_kernel_by_launch_shape.csv
  cp $(ls -t gpurun_out/prof_r06/*/*_kernel_stats.csv | head -1) gpurun_out/r06_bench_kernel_stats.csv
  rm -rf gpurun_out/prof_r06
  echo "rocprof done"
  # the N > 1 sections at world size 1 on the real collective library (the record a first real N = 8 line is read against)
  python3 bench.py --rehearse-multi > gpurun_out/r06_bench_rehearse_multi_world1.json 2> gpurun_out/r06_bench_rehearse_multi_world1.err
  echo "rehearsal (world 1, RCCL) done"
  # two self-launched ranks sharing the one GPU over gloo: the N = 2 line's shape
  python3 bench.py --gpus 2 --backend gloo --share-device --steps 20 --warmup 5 > gpurun_out/r06_bench_n2_rehearsal_shared_gpu.json 2> gpurun_out/r06_bench_n2_rehearsal_shared_gpu.err
  echo "rehearsal (2 ranks, shared GPU, gloo) done"
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  # counter traffic of BASELINE configs 4 and 5 (FETCH_SIZE / WRITE_SIZE in separate passes, calibrated on atx_stream_copy)
  export ROUND=r06
  bash tools/pmc_run.sh "--case config4 --dtype f32 --shard 7" "--case config4 --dtype f32 --shard 7 --tall" "--case config4 --dtype f32 --tall" \
                        "--case config5 --dtype f64" "--case config5 --dtype f64 --tall" "--case config5 --dtype f64 --plain"
  echo "pmc done"
fi
