#!/bin/bash
# Per-round profile collection on the GPU box (developer tool):  bash tools/profile_round.sh r03
# 1) kernel trace + stats of the bench command, 2) three PMC passes (separate runs, --kernel-trace only).
# Outputs under gpurun_out/<tag>/; tools/pmc_traffic.py / pmc_mfma.py / trace_timeline.py fold them into profiles/.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# The per-kernel passes time the step WITHOUT the in-step decode (bench.py --no-wer): the first three decode launches of a process
# (a freshly initialised model emits max_symbols labels per frame: 136 / 87 / 11 ms) would carry a third of the process's kernel
# time and distort every share; one extra stats pass of the default command (with the decode) is kept beside it.
CMD="python3 $ROOT/bench.py --steps 25 --warmup 5 --no-wer --no-cpu-baseline --no-wer-leg --no-peaks"
CMDW="python3 $ROOT/bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-wer-leg --no-peaks"
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_wer -- $CMDW > $OUT/trace_wer.log 2>&1 || { echo trace_wer failed; tail -5 $OUT/trace_wer.log; exit 1; }
echo trace_wer done
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
echo trace done
PM="python3 $ROOT/bench.py --steps 2 --warmup 2 --no-wer --no-cpu-baseline --no-wer-leg --no-peaks"
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $PM > $OUT/pmc_fetch.log 2>&1 || { echo fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo fetch done
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- $PM > $OUT/pmc_write.log 2>&1 || { echo write failed; tail -5 $OUT/pmc_write.log; exit 1; }
echo write done
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $OUT/pmc_mfma -- $PM > $OUT/pmc_mfma.log 2>&1 || { echo mfma failed; tail -5 $OUT/pmc_mfma.log; exit 1; }
echo mfma done
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $T --skip-frac 0.5 > $OUT/timeline.txt 2>&1
KS="joint_grad_h_db_kernel joint_dh_fused_kernel joint_dw_fused_kernel joint_fwd_kernel ffn_fused_kernel relpos_flash_fwd_kernel relpos_flash_bwd_q_kernel relpos_flash_bwd_kv_kernel gemm_bf16_nt_kernel<64 gemm_bf16_nt_dma_kernel greedy_decode_mfma_kernel gemm_bf16_nt_kernel<96 gemm_bf16_nt_kernel<128 gemm_bnsilu_kernel gemm_tn_grouped_kernel gemm_tn_kernel dwconv_fwd_kernel lstm_fwd_kernel lstm_bwd_kernel adamw_seg_kernel cl_penalty_kernel layernorm_kernel layernorm_bwd_kernel conv1_relu_cl_kernel ctc_alpha_beta rnnt_alpha_beta"
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json $KS > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma.py $OUT/pmc_mfma $OUT/pmc_mfma.json $KS > $OUT/pmc_mfma.txt 2>&1
# keep the returned payload small: the raw trace / counter CSVs are large
python3 - <<PY
import glob, os
for f in glob.glob("$OUT/**/*", recursive=True):
    if os.path.isfile(f) and os.path.getsize(f) > 12e6:
        os.remove(f)
PY
du -sh $OUT
