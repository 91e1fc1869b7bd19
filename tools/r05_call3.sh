set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests -m gpu -x -q > gpurun_out/r05_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r05_gpu_tests.log; exit 1; }
tail -2 gpurun_out/r05_gpu_tests.log
bash tools/r05_evidence.sh bench
python3 -c "
import json
d=json.load(open('gpurun_out/r05_bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','scaling')}, d['roofline']['frac'])
print(json.dumps(d['extras']['config4'].get('traffic'))[:400])
"
