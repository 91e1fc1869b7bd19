set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 -m pytest tests/test_gpu_bench_contract.py tests/test_multi_filters.py tests/test_gpu_kernels.py tests/test_gpu_random_shapes.py tests/test_domain_filters.py -m gpu -x -q > gpurun_out/r05_gpu_tests2.log 2>&1 || { tail -40 gpurun_out/r05_gpu_tests2.log; exit 1; }
tail -2 gpurun_out/r05_gpu_tests2.log
bash tools/r05_evidence.sh bench
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05_bench_default.json'))
print({k:d[k] for k in ('value','ms_per_step','scaling')}, d['roofline']['frac'])
s=d['extras']['strong_scaling_shards_on_one_gpu']
print({k:(round(v['slowest_ms'],4), round(v['speedup_bound'],2)) for k,v in s.items() if isinstance(v,dict)})
for f in ('r05_bench_rehearse_multi_world1','r05_bench_n2_rehearsal_shared_gpu'):
    m=json.load(open(f'gpurun_out/{f}.json'))
    print(f, m['scaling'], m['value'], json.dumps(m['config']['multi_gpu'])[:600])
PY
