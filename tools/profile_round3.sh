#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round2.sh <tag> [workload]
# Round-3 evidence set for one workload (default c3 = the default bench line).  Every GPU step runs under its own time
# limit; the script stops at the first step that fails (set -e), so nothing follows a fault or a time-out.
#  1. bench.py (JSON line incl. cpu_baseline)                               -> gpurun_out/<tag>_bench.json
#  2. rocprofv3 --kernel-trace --stats of the same bench command              -> gpurun_out/<tag>_stats/
#  3. PMC passes, one run each, counters only (no trace domains), per-block slot limits respected:
#       fetch  FETCH_SIZE                         write  WRITE_SIZE
#       sq     issue / occupancy                  mix    VALU instruction classes
#       cache  L2 hit/miss, L1 accesses, L1->L2   tatd   TA / TD / TCP busy and stall cycles
#  4. (c3 only) the calibration streams under the same SQ counters            -> gpurun_out/<tag>_calib_pmc/
set -e
tag=${1:-r3}
wl=${2:-c3}
R=/root/repo
O=$R/gpurun_out
rm -rf $O/${tag}_pmc_* $O/${tag}_stats $O/${tag}_calib_pmc
cd /tmp && export TMPDIR=/tmp
WL="--workload $wl"
STEPS=""
[ "$wl" = "c4" ] && STEPS="--steps 10 --warmup 3"
timeout -k 10 500 python3 $R/bench.py $WL $STEPS > $O/${tag}_bench.json 2> $O/${tag}_bench.err
tail -c 1500 $O/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -- \
    python3 $R/bench.py $WL $STEPS --cpu-seconds 0 > $O/${tag}_stats.log 2>&1
BENCH="python3 $R/bench.py $WL --steps 3 --warmup 1 --cpu-seconds 0 --no-extra"
RX='k_trace|k_atrous|k_prep|k_shadow|k_tile_cut|k_pix'
pass() {   # pass <name> <counters...>
    local name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-include-regex "$RX" --output-format csv \
        -d $O/${tag}_pmc_$name -- $BENCH > $O/${tag}_pmc_$name.log 2>&1
    echo "pmc pass $name done"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY
pass mix SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD
pass cache TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass tatd GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
python3 $R/tools/pmc_summary.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_sq $O/${tag}_pmc_mix $O/${tag}_pmc_cache $O/${tag}_pmc_tatd \
    > $O/${tag}_pmc_summary.txt
if [ "$wl" = "c3" ]; then
    timeout -k 10 120 $R/tools/calib/issue_rates2 > $O/${tag}_issue_rates2.jsonl
    timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv \
        -d $O/${tag}_calib_pmc -- $R/tools/calib/issue_rates2 > $O/${tag}_calib_pmc.log 2>&1
    python3 $R/tools/pmc_summary.py $O/${tag}_calib_pmc > $O/${tag}_calib_pmc_summary.txt
fi
cat $O/${tag}_pmc_summary.txt
