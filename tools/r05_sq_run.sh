# SQ counter pass over the float64 per-point kernels that evaluate library functions (round 5): share of wave time parked on memory /
# stalled at issue / issuing, VALU instructions per wave — the evidence behind DESIGN.md §3 "The float64 library functions, round 5".
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
    --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/tools/pmc_sq_probe.py > $R/gpurun_out/r05_pmc_sq.log 2>&1 || { tail -20 $R/gpurun_out/r05_pmc_sq.log; exit 1; }
python3 $R/tools/pmc_sq_probe.py --summarize $R/gpurun_out/pmc_sq > $R/gpurun_out/r05_pmc_sq_transcendentals.txt
cat $R/gpurun_out/r05_pmc_sq_transcendentals.txt
