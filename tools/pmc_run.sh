set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in 4 16; do
  rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/pmc_probe.py --k $K > $R/gpurun_out/r02_pmc_fetch_k$K.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/pmc_probe.py --k $K > $R/gpurun_out/r02_pmc_write_k$K.log 2>&1
  python3 $R/tools/pmc_summarize.py --round r02 --out $R/gpurun_out/r02_traffic.json > $R/gpurun_out/r02_pmc_summary_k$K.log 2>&1
  tail -12 $R/gpurun_out/r02_pmc_summary_k$K.log
done
