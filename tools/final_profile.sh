#!/bin/bash
# the round's profile set: bench line, rocprofv3 kernel trace of the same command, three PMC passes (GPU box).
# Output: gpurun_out/final/{bench.json, bench_under_rocprof.json, kernel_stats.csv, hbm_traffic_pmc.json, mfma_pmc.json}
# - copy them to profiles/rNN_<tag>_* afterwards.  Counter passes carry --kernel-trace only (gpurun refuses --pmc
# together with the other trace domains).
set -o pipefail
R=/root/repo
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 5 --warmup 2 --no-cpu-baseline --no-torch-baseline"
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline > $O/bench_under_rocprof.json 2> $O/trace.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o f -- python3 $R/bench.py $SHORT > $O/pmc_f.json 2> $O/pmc_f.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o w -- python3 $R/bench.py $SHORT > $O/pmc_w.json 2> $O/pmc_w.err || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_m -o m -- python3 $R/bench.py $SHORT > $O/pmc_m.json 2> $O/pmc_m.err || exit 7
python3 $R/tools/kernel_stats.py $O/trace/t_results.db > $O/kernel_stats.csv || exit 5
python3 $R/tools/pmc_traffic.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db > $O/hbm_traffic_pmc.json || exit 6
python3 $R/tools/pmc_mfma.py $O/pmc_m/m_results.db > $O/mfma_pmc.json || exit 8
rm -rf $O/trace $O/pmc_f $O/pmc_w $O/pmc_m
tail -c 600 $O/bench.json
