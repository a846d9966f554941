# cycles (not time) of the three-term inference kernel for several library builds: GRBM_GUI_ACTIVE / 8 = cycles per XCD
set -u
OUT=$PWD/gpurun_out/s3/x3icyc
mkdir -p "$OUT"
export TMPDIR=/tmp
for n in "$@"; do
    export NERF_HIP_LIB=$PWD/nerf-pytorch_amd/lib/x3i_$n.so
    rm -rf "$OUT/$n"
    VARIANTS=0 timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/$n" -o p --output-format csv -- python3 tests/diag/x3i_variants.py > "$OUT/$n.out" 2> "$OUT/$n.log" || exit 1
    python3 profiles/summarize_pmc.py "$OUT/$n" | grep "fwd_x3i_kernel<0,false>" | cut -d, -f2- | sed "s/^/$n: /"
    grep "x3i v0 again" "$OUT/$n.out" | tail -1
done
