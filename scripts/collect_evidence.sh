#!/bin/bash
# Copy what one evidence run (scripts/gpu_profile.sh <tag>, gpu_pmc_icache.sh <tag>ic, gpu_prof_1m.sh <tag>m1, gpu_final.sh <tag>f, the GPU suite under <tag>t, the timelines under <tag>tl)
# merged back into gpurun_out/ into profiles/ under the round's names.  usage: scripts/collect_evidence.sh r06
T=${1:-r06}; G=gpurun_out
cp $G/$T/profiles/${T}_* profiles/
sed "s#/tmp/code/[^ ]*/gpurun_out/$T/profiles/##" $G/$T/phase_stats.txt > profiles/${T}_phase_stats.txt
sed "s#/tmp/code/[^ ]*/repo/##g" $G/${T}ic/log.txt | grep -v "^[WEI]2026" > profiles/${T}_pmc_issue_counters.txt
cp $G/${T}m1/pmc_256_1M_water.txt profiles/${T}_pmc_256_1M_water.txt; cp $G/${T}m1/kernel_stats_256_1M_water.csv profiles/${T}_kernel_stats_256_1M_water.csv
cp $G/${T}t/pytest_gpu_measured.txt profiles/${T}_pytest_gpu_measured.txt
cp $G/${T}f/device.txt profiles/${T}_device.txt
cp $G/${T}f/bench_driver.json profiles/${T}_bench_driver_flags.json; cp $G/${T}f/bench_default.json profiles/${T}_bench_default_no_extras.json
cp $G/${T}f/replica_config3.json profiles/${T}_replica_config3_one_rank_rccl.json
for B in 1 2 3; do cp $G/${T}f/replica_as_shipped_B$B.json profiles/${T}_replica_as_shipped_B${B}_per_gpu.json; done
cp $G/${T}tl/timeline.txt profiles/${T}_timeline_final_build.txt; cp $G/${T}tl/timeline_tail.txt profiles/${T}_timeline_tail.txt
python scripts/design_tables.py $T
