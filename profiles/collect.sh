#!/bin/bash
# Collects the profiles of one round on the GPU box (run from the repository root through gpurun):
#   bash profiles/collect.sh r3            (everything)
#   bash profiles/collect.sh r3 modes      (only the bench lines of the other modes and the HBM-bound kernels)
#   bash profiles/collect.sh r3 pmc        (only the counter passes)
#   bash profiles/collect.sh r3 stats      (only the rocprofv3 --kernel-trace --stats runs)
# writes gpurun_out/<tag>/prof/<tag>_*.csv|json|jsonl; copy what should be judged into profiles/.
#   * rocprofv3 --kernel-trace --stats of the default bench command and of its other modes
#   * separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) of the single-pass workload, where every
#     fused launch covers exactly 1,048,576 samples (the figure <tag>_pmc_meta.json records for bench.py)
#   * bench.py lines of the other modes (builder-run, kept next to the profiles)
set -u
TAG=${1:-r3}
ONLY=${2:-all}
OUT=$PWD/gpurun_out/$TAG/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
stats() {   # name, bench arguments...
    local name=$1; shift
    rm -rf "$OUT/kt_$name"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$name" -o p --output-format csv -- python3 bench.py "$@" --no-cpu-baseline \
        > "$OUT/${TAG}_bench_${name}_profiled.json" 2> "$OUT/kt_$name.log" || return 1
    cp "$(find "$OUT/kt_$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_${name}_kernel_stats.csv"
}
pmc() {     # name, counters...   (NeRF single-pass workload; name "siren_*" profiles the Siren network instead,
            #                      name "x3_*" the opt-in bf16x3 training kernels of the NeRF network, name "x3i_*" the
            #                      bf16x3 inference kernel: 32768 rays x 256 samples = 8,388,608 samples per launch)
    local name=$1; shift
    local model=nerf
    local work="--train-precision fp32"
    case "$name" in siren_*) model=siren;; x3siren_*) model=siren; work="--train-precision bf16x3";; x3_*) work="--train-precision bf16x3";; x3i_*) work="--mode infer --precision bf16x3 --rays 32768";; esac
    rm -rf "$OUT/pmc_$name"
    timeout -s KILL 240 rocprofv3 --pmc "$@" -d "$OUT/pmc_$name" -o p --output-format csv -- python3 bench.py --sampling single --model $model $work --steps 3 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/pmc_$name.log" || return 1
    python3 profiles/summarize_pmc.py "$OUT/pmc_$name" > "$OUT/${TAG}_pmc_$name.csv"
}
pmc_all() {
    pmc fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
    pmc write WRITE_SIZE &&
    pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pmc insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
    pmc siren_fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
    pmc siren_write WRITE_SIZE &&
    pmc siren_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pmc siren_insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
    pmc x3_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pmc x3_insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
    pmc x3_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE &&
    pmc x3_fetch FETCH_SIZE GRBM_GUI_ACTIVE &&
    pmc x3_write WRITE_SIZE &&
    pmc x3siren_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pmc x3i_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
    pmc x3i_insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE &&
    pmc x3i_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
    echo '{"samples_per_dispatch": 1048576, "command": "rocprofv3 --pmc <counters> -- python3 bench.py --sampling single [--model siren] --steps 3 --warmup 1 --no-cpu-baseline", "unit": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (profiles/summarize_pmc.py means); files named *_pmc_siren_* are the Siren network, *_pmc_x3_* the opt-in bf16x3 training kernels (--train-precision bf16x3; *_pmc_x3siren_* the Siren network in that mode), *_pmc_x3i_* the bf16x3 inference kernel (--mode infer --precision bf16x3 --rays 32768: 8,388,608 samples per dispatch)"}' > "$OUT/${TAG}_pmc_meta.json"
}
modes() {
    : > "$OUT/${TAG}_bench_modes.jsonl"
    for args in "--sampling single" "--sampling single --model siren" "--model siren" "--train-precision bf16x3" "--train-precision bf16x3 --sampling single" "--train-precision bf16x3 --model siren" "--train-precision bf16x3 --model siren --sampling single" "--train-precision bf16x3 --rays 16384" "--mode infer" "--mode infer --sampling single" \
                "--mode infer --precision bf16" "--mode infer --precision bf16 --rays 32768" "--mode infer --precision bf16x3" "--mode infer --precision bf16x3 --rays 32768" "--mode infer --model siren" "--mode infer --model siren --precision bf16x3" "--mode infer --precision bf16x3 --rays 640000 --steps 3 --warmup 1" "--mode infer --rays 640000 --steps 3 --warmup 1" \
                "--rays 1024" "--rays 16384"; do
        timeout -k 10 300 python3 bench.py $args --no-cpu-baseline >> "$OUT/${TAG}_bench_modes.jsonl" 2>> "$OUT/modes.err" || return 1
    done
    timeout -k 10 300 python3 bench_kernels.py > "$OUT/${TAG}_hbm_kernels.jsonl" 2> "$OUT/hbm.err"
}
if [ "$ONLY" = modes ]; then
    modes && echo "modes collected in $OUT"
    exit $?
fi
if [ "$ONLY" = stats ]; then
    stats hier && stats single --steps 6 --warmup 2 --sampling single && stats siren --steps 6 --warmup 2 --sampling single --model siren &&
    stats x3 --steps 6 --warmup 2 --train-precision bf16x3 && stats x3_single --steps 6 --warmup 2 --sampling single --train-precision bf16x3 &&
    stats x3_siren --steps 6 --warmup 2 --sampling single --model siren --train-precision bf16x3 &&
    stats infer_bf16x3 --steps 6 --warmup 2 --mode infer --precision bf16x3 --rays 32768 && echo "kernel statistics collected in $OUT"
    exit $?
fi
if [ "$ONLY" = pmc ]; then
    pmc_all && echo "counter passes collected in $OUT"
    exit $?
fi
stats hier &&    # the default bench command itself (10 timed steps after 3 warm-ups)
stats single --steps 6 --warmup 2 --sampling single &&
stats siren --steps 6 --warmup 2 --sampling single --model siren &&
stats x3 --steps 6 --warmup 2 --train-precision bf16x3 &&
stats x3_single --steps 6 --warmup 2 --sampling single --train-precision bf16x3 &&
stats x3_siren --steps 6 --warmup 2 --sampling single --model siren --train-precision bf16x3 &&
stats infer_bf16x3 --steps 6 --warmup 2 --mode infer --precision bf16x3 --rays 32768 &&
pmc_all &&
modes &&
echo "profiles collected in $OUT"
