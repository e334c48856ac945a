#!/bin/bash
# end-of-round evidence: kernel stats + HBM traffic of the headline run, SQ/MFMA counters, training-step PMC passes, smoke
cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof.sh > gpurun_out/prof_summary.txt 2>&1
bash scripts/gpu_pmc_mfma.sh > gpurun_out/pmcm_summary.txt 2>&1
bash scripts/gpu_pmc_train.sh 128 bf16 > gpurun_out/pmct_summary.txt 2>&1
bash scripts/gpu_prof_train.sh 128 bf16 > gpurun_out/proft2_summary.txt 2>&1
cp gpurun_out/proft2/t_kernel_stats.csv gpurun_out/proft2_b128_bf16_kernel_stats.csv
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.txt 2>&1
tail -3 gpurun_out/prof_summary.txt; tail -12 gpurun_out/pmcm_summary.txt; tail -4 gpurun_out/pmct_summary.txt; head -6 gpurun_out/proft2_summary.txt; tail -2 gpurun_out/smoke.txt
