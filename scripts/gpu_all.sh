#!/bin/bash
# One box, everything the round's evidence needs, in the order the files depend on each other: GPU test suite (measured lines kept), rocprofv3
# phases + PMC of the driver's command -> profiles/ (bench.py reads them beside the same build), issue counters, bench lines + replicas.
TAG=${1:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
echo "== pytest -m gpu -s"; timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "MEASURED|passed|failed|Error|error" | cut -c1-1500 > gpurun_out/$TAG/pytest_gpu_measured.txt; tail -1 gpurun_out/$TAG/pytest_gpu_measured.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_profile.sh $TAG 20 5 2>&1 | tail -30
cp gpurun_out/$TAG/profiles/* profiles/
bash scripts/gpu_pmc_icache.sh ${TAG}ic > gpurun_out/$TAG/pmc_issue_counters.txt 2>&1; grep -E "k_p2g|k_g2p|k_grid|k_pgg" gpurun_out/$TAG/pmc_issue_counters.txt | grep VALU | cut -c1-250
echo "== work list by window"; timeout 300 python scripts/work_probe.py 35 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/work_list_by_window.txt; tail -3 gpurun_out/$TAG/work_list_by_window.txt
bash scripts/gpu_final.sh ${TAG}f
bash scripts/gpu_prof_1m.sh ${TAG}m1 2>&1 | tail -12
