# rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as the microarchitecture guide prescribes) over tools/pmc_probe.py.
#   bash tools/pmc_run.sh "<probe args>" ["<probe args>" ...]     e.g.  bash tools/pmc_run.sh "--k 4" "--k 16" "--k 4 --layout fields"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ARGS in "$@"; do
  TAG=$(echo "$ARGS" | tr -d ' -')
  rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/pmc_probe.py $ARGS > $R/gpurun_out/${ROUND:-r03}_pmc_fetch_$TAG.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/pmc_probe.py $ARGS > $R/gpurun_out/${ROUND:-r03}_pmc_write_$TAG.log 2>&1
  python3 $R/tools/pmc_summarize.py --round ${ROUND:-r03} --out $R/gpurun_out/${ROUND:-r03}_traffic.json > $R/gpurun_out/${ROUND:-r03}_pmc_summary_$TAG.log 2>&1
  tail -12 $R/gpurun_out/${ROUND:-r03}_pmc_summary_$TAG.log
done
