# Evidence batch of round 2 (run on the MI355X box through gpurun): every file lands in gpurun_out/, the ones to keep are copied to profiles/.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/level_sweep.py > gpurun_out/r02_level_sweep.log 2>&1
echo "level sweep done"
python3 tools/kernel_bench.py --out gpurun_out/r02_kernel_bench.json > gpurun_out/r02_kernel_bench.log 2>&1
echo "kernel bench done"
python3 bench.py --src-grid o2560 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_o2560.json 2> gpurun_out/r02_bench_o2560.err
echo "o2560 done"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/r02_bench_under_rocprof.err)
echo "rocprof done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-device > gpurun_out/r02_bench_n2_rehearsal.json 2> gpurun_out/r02_bench_n2_rehearsal.err
echo "n2 rehearsal done"
