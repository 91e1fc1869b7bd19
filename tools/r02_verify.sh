set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all3.log 2>&1 || { tail -30 gpurun_out/r02_gpu_all3.log; exit 1; }
tail -2 gpurun_out/r02_gpu_all3.log
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1; tail -1 gpurun_out/r02_smoke.log
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
echo "bench default done"
python3 bench.py --src-grid o2560 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_o2560.json 2> gpurun_out/r02_bench_o2560.err
echo "o2560 done"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --share-device > gpurun_out/r02_bench_n2_rehearsal.json 2> gpurun_out/r02_bench_n2_rehearsal.err
echo "n2 rehearsal done"
