#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/: a kernel trace + stats of `bench.py` and the PMC passes the
# bench's `roofline` object reads (MI355X_MICROARCH.md, rocprofv3 PMC slots: FETCH_SIZE and WRITE_SIZE in separate
# passes; counters never together with a trace).  Run on the GPU box:
#     bash tools/profile.sh NAME KERNEL_SUBSTRING WORKLOAD [extra bench.py flags]
# Writes gpurun_out/NAME/{kernel_stats.csv, bench_under_rocprof.json, pmc.json} (gpurun_out/ is what travels back from the
# GPU box; copy the directory to profiles/NAME to keep it); raw outputs stay under gpurun_out/prof_NAME.
set -euo pipefail
OUT=$1; KERNEL=$2; WORKLOAD=$3; shift 3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RAW=$ROOT/gpurun_out/prof_$OUT
OUT=gpurun_out/$OUT
mkdir -p "$ROOT/$OUT" "$RAW"
BENCH="python3 $ROOT/bench.py --workload $WORKLOAD --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-workloads $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$RAW/kt" -- $BENCH > "$ROOT/$OUT/bench_under_rocprof.json"
cp "$(ls "$RAW"/kt/*/*kernel_stats.csv | head -1)" "$ROOT/$OUT/kernel_stats.csv"
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i + 1))
    rocprofv3 --pmc $PMC --output-format csv -d "$RAW/pmc$i" -- $BENCH > /dev/null
done
python3 "$ROOT/tools/pmc_summary.py" "$KERNEL" "$WORKLOAD" "$ROOT/$OUT/pmc.json" "$RAW"/pmc1 "$RAW"/pmc2 "$RAW"/pmc3 "$RAW"/pmc4 > /dev/null
head -4 "$ROOT/$OUT/kernel_stats.csv"
cat "$ROOT/$OUT/pmc.json"
