# round 5, first GPU call: the whole -m gpu suite, the per-kernel table, the default bench line
set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r05_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r05_gpu_tests.log
python3 tools/kernel_bench.py --out gpurun_out/r05_kernel_bench.json > gpurun_out/r05_kernel_bench.log 2>&1 || { tail -30 gpurun_out/r05_kernel_bench.log; exit 1; }
grep -i "log then exp\|snow_cover\|regrid_ell k=4 columns\|affine out" gpurun_out/r05_kernel_bench.log || true
python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err || { tail -30 gpurun_out/r05_bench_default.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r05_bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','scaling')}, d['roofline']['frac'])
print(json.dumps(d['extras'].get('strong_scaling_shards_on_one_gpu'), indent=None)[:1500])
"
