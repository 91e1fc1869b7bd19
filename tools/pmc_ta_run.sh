# TA / GRBM counter passes over tools/pmc_gather_probe.py (run on the GPU box through gpurun).
# Round 3's single pass asked for four TA_* counters plus GRBM_GUI_ACTIVE at once and rocprofv3 aborted with
# "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of the hardware to collect"
# (gpurun_out/pmc_gather_5.log): the TA block has fewer counter slots than that.  Two TA counters per pass fit.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_ta${ATX_PROBE_DTYPE:+_$ATX_PROBE_DTYPE}
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 120 rocprofv3 -L > $R/gpurun_out/r04_rocprofv3_list_avail.txt 2>&1 || echo "rocprofv3 -L failed (list kept)"
i=0
FAILED=0
# (GRBM_GUI_ACTIVE rides along in every pass so that each counter is normalised by the active cycles of its own run)
for SET in "TA_TA_BUSY GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_BUSY_max TA_BUSY_min GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE" \
           "TA_FLAT_READ_WAVEFRONTS TA_FLAT_WAVEFRONTS GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TD_CYCLES TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE" \
           "TD_TD_BUSY TD_TC_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$i -- python3 $R/tools/pmc_gather_probe.py > $R/gpurun_out/pmc_ta${ATX_PROBE_DTYPE:+_$ATX_PROBE_DTYPE}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_ta${ATX_PROBE_DTYPE:+_$ATX_PROBE_DTYPE}_$i.log; echo "pass $i FAILED: $SET"; FAILED=1; break; }
  echo "pass $i done: $SET"
done
python3 $R/tools/pmc_gather_probe.py --summarize $OUT > $R/gpurun_out/r04_pmc_ta_counters${ATX_PROBE_DTYPE:+_$ATX_PROBE_DTYPE}.txt
cat $R/gpurun_out/r04_pmc_ta_counters${ATX_PROBE_DTYPE:+_$ATX_PROBE_DTYPE}.txt
exit $FAILED
