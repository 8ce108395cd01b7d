#!/bin/bash
# rocprofv3 evidence for the two kernels beside the filters -- filter1d_grad_kernel (forward-mode NLL gradient) and
# cf1d_fast_kernel (characteristic function) -- on `bench.py --extra-kernels` (the same calls the bench line's
# other_workloads.gradient_N7_P2 / characteristic_fn_N15 time).  Run on the GPU box:
#     bash tools/profile_extra.sh NAME
# Writes gpurun_out/NAME/{kernel_stats.csv, extra_under_rocprof.json, pmc_grad.json, pmc_cf.json}.
set -euo pipefail
OUT=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RAW=$ROOT/gpurun_out/prof_$OUT
OUT=gpurun_out/$OUT
mkdir -p "$ROOT/$OUT" "$RAW"
BENCH="python3 $ROOT/bench.py --extra-kernels"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/kt" -- $BENCH > "$ROOT/$OUT/extra_under_rocprof.json"
cp "$(ls "$RAW"/kt/*/*kernel_stats.csv | head -1)" "$ROOT/$OUT/kernel_stats.csv"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i + 1))
    rocprofv3 --pmc $PMC --output-format csv -d "$RAW/pmc$i" -- $BENCH > /dev/null
done
python3 "$ROOT/tools/pmc_summary.py" filter1d_grad_kernel "well_poisson_N7_T1000_B16384 gradient P=2" "$ROOT/$OUT/pmc_grad.json" "$RAW"/pmc1 "$RAW"/pmc2 "$RAW"/pmc3 "$RAW"/pmc4 > /dev/null
python3 "$ROOT/tools/pmc_summary.py" cf1d_fast_kernel "characteristic_fn N15 8192 x 2000" "$ROOT/$OUT/pmc_cf.json" "$RAW"/pmc1 "$RAW"/pmc2 "$RAW"/pmc3 "$RAW"/pmc4 > /dev/null
head -8 "$ROOT/$OUT/kernel_stats.csv"
