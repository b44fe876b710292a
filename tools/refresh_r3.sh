#!/bin/bash
# usage (here, not on the GPU box): tools/refresh_r3.sh
# Regenerates the round-3 evidence set in profiles/ for the current build: two gpurun calls of tools/profile_round3.sh (C3, C4),
# PMC summaries, kernel stats, ISA census of k_trace_q / k_atrous_tile, profiles/r3_k1_valu_model.json, and a final default
# bench.py run that reads that model (profiles/r3_final_bench.json).
set -e
R=/root/repo
cd $R
make -s -C tools/calib
export VKR_ROUND=r3
tools/grun.sh 1200 'tools/profile_round3.sh r3_final c3 > gpurun_out/profile_round3.log 2>&1 && tools/profile_round3.sh r3_final_c4 c4 > gpurun_out/profile_round3_c4.log 2>&1; tail -1 gpurun_out/profile_round3_c4.log' 2>&1 | tail -2
cd $R/gpurun_out
for d in r3_final_pmc_* r3_final_c4_pmc_* r3_final_stats r3_final_c4_stats r3_final_calib_pmc; do
  [ -d "$d/vm" ] || continue
  for suf in counter_collection agent_info kernel_stats kernel_trace domain_stats; do ls -t $d/vm/*_${suf}.csv 2>/dev/null | tail -n +2 | xargs -r rm -f; done
done
for t in r3_final r3_final_c4; do
  python3 ../tools/pmc_summary.py ${t}_pmc_fetch ${t}_pmc_write ${t}_pmc_sq ${t}_pmc_mix ${t}_pmc_cache ${t}_pmc_tatd > ${t}_pmc_summary.txt
  cp ${t}_pmc_summary.txt ${t}_bench.json ../profiles/
  cp ${t}_stats/vm/*_kernel_stats.csv ../profiles/${t}_kernel_stats.csv
done
python3 ../tools/pmc_summary.py r3_final_calib_pmc > r3_final_calib_pmc_summary.txt
cp r3_final_calib_pmc_summary.txt r3_final_issue_rates2.jsonl ../profiles/
mkdir -p /tmp/isa_r3 && cd /tmp/isa_r3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --save-temps -c $R/vulkanrenderer_amd/csrc/rtao_kernels.hip -o rtao.o 2>&1 | grep " error" || true
S=rtao_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
python3 $R/tools/isa_census.py $S _ZN3vkr9k_trace_qILb0ELi7EE depth3 > $R/profiles/r3_k1_isa_census.json
python3 $R/tools/isa_census.py $S _ZN3vkr13k_atrous_tileILb0EE > $R/profiles/r3_k2_isa_census.json
grep -n "k_trace_qILb0ELi7EE" -A40 $S | grep -E "vgpr_count|vgpr_spill|private_segment_fixed|sgpr_spill" | tail -4
cd $R
python3 tools/make_valu_model.py r3_final:c3 r3_final_c4:c4 > /dev/null
tools/grun.sh 600 'timeout -k 10 400 python3 bench.py > gpurun_out/r3_final_bench.json 2> gpurun_out/r3_final_bench.err; echo done' | tail -1
cp gpurun_out/r3_final_bench.json profiles/r3_final_bench.json
python3 tools/docs_tables.py --write          # the result tables of DESIGN.md / BASELINE.md / README.md follow the new evidence
python3 - <<'PY'
import json,csv
d=json.loads(open('/root/repo/profiles/r3_final_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('bench', d['ms_per_step'], d['value'], d['kernels'], r['achieved'], r['peak'], r['frac'], r['frac_estimate'], r['traffic'], r['lane_util_node_loop'], r['lane_util_tri_loop'], r['nodes_per_ray'], r['tris_per_ray'], r['cache_side_bytes_per_launch'], d['parity']['mismatching_pixels'], d['cpu_baseline']['value'], d['cpu_baseline']['single_thread']['value'], d['roofline_k2']['frac'], d['config']['lbvh_build_ms'])
for f in ('r3_final','r3_final_c4'):
    for r in list(csv.DictReader(open('/root/repo/profiles/%s_kernel_stats.csv'%f))):
        if any(k in r['Name'] for k in ('k_trace_q<false','k_pix','k_atrous_tile','k_prep','k_shadow_sun_pkt')): print('   ', r['Name'][:40], r['Calls'], r['AverageNs'])
d=json.loads(open('/root/repo/profiles/r3_final_c4_bench.json').read().strip().splitlines()[-1]); print('c4', d['ms_per_step'], d['value'], d['kernels'])
m=json.load(open('/root/repo/profiles/r3_k1_valu_model.json'))
for w in ('c3','c4'):
    print(w,{k:m[w][k] for k in ('valu_insts_per_ray','valu_insts_per_launch','salu_insts_per_launch','lds_insts_per_launch','vmem_rd_insts_per_launch','td_busy_frac','ta_busy_frac','td_tc_stall_frac','l1_hit_rate','l2_hit_rate','hbm_bytes_per_launch','mean_issue_cycles_lower','mean_issue_cycles_estimate','wave_cycles_issue_stalled_frac')})
PY
