# round 6: the records around the bench line refreshed — the per-kernel table's own rocprofv3 summary, the host-fed job, Filter.forward() wall times
set -e
R=$GRAFT_REPO_ROOT
cd $R
(cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_kb && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kb -- python3 $R/tools/kernel_bench.py > $R/gpurun_out/r06_kernel_bench_under_rocprof.log 2>&1)
cp $(ls -t gpurun_out/prof_kb/*/*_kernel_stats.csv | head -1) gpurun_out/r06_kernel_bench_kernel_stats_full.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_kernel_bench_kernel_stats_full.csv')))
keep=[r for r in rows if 'atx::' in r['Name']]
with open('gpurun_out/r06_kernel_bench_kernel_stats.csv','w',newline='') as f:
    w=csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
    for r in keep:
        r=dict(r); r['Name']=r['Name'][:160]; w.writerow(r)
print(len(keep), 'library kernels')
PY
rm -rf gpurun_out/prof_kb
echo "kernel_bench under rocprofv3 done"
python3 tools/host_path_bench.py > gpurun_out/r06_host_fed_path.json 2> gpurun_out/r06_host_fed_path.err
tail -c 600 gpurun_out/r06_host_fed_path.json; echo
python3 tools/api_bench.py > gpurun_out/r06_api_bench.log 2>&1
tail -6 gpurun_out/r06_api_bench.log | cut -c1-200
