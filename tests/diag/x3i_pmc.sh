# counter passes of the three-term inference kernels (tests/diag/x3i_variants.py: 1,048,576 samples per launch)
set -u
OUT=$PWD/gpurun_out/s3/x3ipmc
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {
    local name=$1; shift
    rm -rf "$OUT/$name"
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$OUT/$name" -o p --output-format csv -- python3 tests/diag/bf16x3_speed.py ${MODEL_ARG:-} > /dev/null 2> "$OUT/$name.log" || return 1
    python3 profiles/summarize_pmc.py "$OUT/$name" > "$OUT/x3i_pmc_$name.csv"
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE &&
run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
grep -h "x3i\|bf16_kernel\|kernel," "$OUT"/x3i_pmc_*.csv | grep -v pack | cut -c1-40,140-260
