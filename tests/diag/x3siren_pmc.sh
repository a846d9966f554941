set -u
OUT=$PWD/gpurun_out/r3/prof; mkdir -p "$OUT"; export TMPDIR=/tmp
rm -rf "$OUT/pmc_x3siren_sq"
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_x3siren_sq" -o p --output-format csv -- python3 bench.py --sampling single --model siren --train-precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_x3siren_sq.log" || exit 1
python3 profiles/summarize_pmc.py "$OUT/pmc_x3siren_sq" > "$OUT/r3_pmc_x3siren_sq.csv"
grep -E "x3i_kernel" "$OUT/r3_pmc_x3siren_sq.csv" | awk -F, '{print $(NF-3)" "$(NF-2)" "$(NF-1)" "$NF}'
