#!/bin/bash
# the round's profile set (GPU box).  Output under gpurun_out/final/ - copy to profiles/rNN_<tag>_* afterwards:
#   bench.json                      the default bench line (BASELINE configs[1])
#   bench_under_rocprof.json, kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (short form)
#   hbm_traffic_pmc.json            FETCH_SIZE / WRITE_SIZE passes
#   mfma_pmc.json                   SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE pass
#   fp8_*.json                      the same three for --weight-format fp8_mfma, plus the configs[4] line
# Counter passes carry --kernel-trace only (gpurun refuses --pmc together with the other trace domains).
# Each counter pass keeps the bench.py line it ran (pmc_*.json): the tools check the pass against exact quantities of that
# run (tools/pmc_common.py: coverage) and store the kernel-source fingerprint; RAJNI_GIT_HEAD=<commit> in the environment
# is recorded as git_commit (the box has no .git).
set -o pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 5 --warmup 2 --no-cpu-baseline --no-torch-baseline"
F8="--weight-format fp8_mfma"
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline > $O/bench_under_rocprof.json 2> $O/trace.err || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_WRREQ_64B_sum --kernel-trace -d $O/pmc_f -o f -- python3 $R/bench.py $SHORT > $O/pmc_f.json 2> $O/pmc_f.err || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o w -- python3 $R/bench.py $SHORT > $O/pmc_w.json 2> $O/pmc_w.err || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_m -o m -- python3 $R/bench.py $SHORT > $O/pmc_m.json 2> $O/pmc_m.err || exit 7
python3 $R/tools/kernel_stats.py $O/trace/t_results.db > $O/kernel_stats.csv || exit 5
python3 $R/tools/pmc_traffic.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db $O/pmc_f.json > $O/hbm_traffic_pmc.json || exit 6
python3 $R/tools/pmc_mfma.py $O/pmc_m/m_results.db $O/pmc_m.json > $O/mfma_pmc.json || exit 8
rm -rf $O/trace $O/pmc_f $O/pmc_w $O/pmc_m
# ---- the fp8 matrix-pipe format
timeout -k 10 300 python3 $R/bench.py $F8 --no-cpu-baseline > $O/fp8_bench.json 2> $O/fp8_bench.err || exit 11
timeout -k 10 300 python3 $R/bench.py --model deit3_base_patch16_224 --batch 512 $F8 --no-cpu-baseline > $O/fp8_bench_deit3_b512.json 2> $O/fp8_deit3.err || exit 12
timeout -k 10 300 python3 $R/bench.py --model deit3_base_patch16_224 --batch 512 $F8 --residual bf16 --no-cpu-baseline > $O/fp8_bench_deit3_b512_bf16stream.json 2> $O/fp8_deit3b.err || exit 13
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace8 -o t -- python3 $R/bench.py $F8 --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline > $O/fp8_bench_under_rocprof.json 2> $O/trace8.err || exit 14
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F8 GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_m8 -o m -- python3 $R/bench.py $F8 $SHORT > $O/pmc_m8.json 2> $O/pmc_m8.err || exit 15
python3 $R/tools/kernel_stats.py $O/trace8/t_results.db > $O/fp8_kernel_stats.csv || exit 16
python3 $R/tools/pmc_mfma.py $O/pmc_m8/m_results.db $O/pmc_m8.json > $O/fp8_mfma_pmc.json || exit 17
rm -rf $O/trace8 $O/pmc_m8
# ---- the other single-GPU BASELINE configs
timeout -k 10 300 python3 $R/bench.py --model vit_large_patch16_384 --batch 64 --schedule '{"4":{"keep_ratio":0.7},"12":{"keep_ratio":0.5},"20":{"keep_ratio":0.3}}' --no-cpu-baseline > $O/bench_l384.json 2> $O/l384.err || exit 21
timeout -k 10 300 python3 $R/bench.py --model vit_tiny_patch16_224 --no-cpu-baseline > $O/bench_tiny.json 2> $O/tiny.err || exit 22
tail -c 400 $O/bench.json
