set -u
OUT=$PWD/gpurun_out/s3/tcp
mkdir -p "$OUT"; export TMPDIR=/tmp
rm -rf "$OUT/p"
timeout -s KILL 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE -d "$OUT/p" -o p --output-format csv -- python3 bench.py --sampling single --train-precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/p.log" || { tail -5 "$OUT/p.log"; exit 1; }
python3 profiles/summarize_pmc.py "$OUT/p" | grep -E "x3_kernel|kernel," | cut -d, -f2-
