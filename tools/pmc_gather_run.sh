# counter passes over tools/pmc_gather_probe.py (run on the GPU box through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_gather
i=0
for SET in "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_GATE_EN1" \
           "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TCR_TCP_STALL_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES"; do
  # (round 3 tried a fifth pass with FOUR TA_* counters + GRBM_GUI_ACTIVE: rocprofiler_create_counter_config error 38, "Request exceeds the capabilities of
  #  the hardware to collect" — the TA block has two slots per pass — SIGABRT, and rocprofv3's signal handler never returned.  An oversubscribed pass, not the
  #  pool: tools/pmc_ta_run.sh collects them two at a time, profiles/r04_pmc_gather_counters.txt)
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gather/$i -- python3 $R/tools/pmc_gather_probe.py > $R/gpurun_out/pmc_gather_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_gather_$i.log; exit 1; }
  echo "pass $i done"
done
python3 $R/tools/pmc_gather_probe.py --summarize $R/gpurun_out/pmc_gather > $R/gpurun_out/r03_pmc_gather_counters.txt
cat $R/gpurun_out/r03_pmc_gather_counters.txt
