# counter passes of the bf16x3 training kernels (single-pass workload, 1,048,576 samples per launch)
set -u
OUT=$PWD/gpurun_out/r3/x3pmc
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
    local name=$1; shift
    rm -rf "$OUT/$name"
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 bench.py --sampling single --train-precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/$name.log" || return 1
    python3 profiles/summarize_pmc.py "$OUT/$name" > "$OUT/x3_pmc_$name.csv"
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE &&
run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE &&
run mem FETCH_SIZE GRBM_GUI_ACTIVE && run wr WRITE_SIZE
grep -h "x3\|kernel," "$OUT"/x3_pmc_*.csv | cut -c1-220
