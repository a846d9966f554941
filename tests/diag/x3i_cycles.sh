# cycles (not time) of the three-term inference kernel for several library builds (make -C nerf-pytorch_amd/csrc variant NAME=x DEFS=...
# -> lib/variant_x.so; "." = the production library): GRBM_GUI_ACTIVE / 8 = cycles per XCD.  usage: x3i_cycles.sh . x y
set -u
OUT=$PWD/gpurun_out/s3/x3icyc
mkdir -p "$OUT"
export TMPDIR=/tmp
for n in "$@"; do
    if [ "$n" = . ]; then unset NERF_HIP_LIB; else export NERF_HIP_LIB=$PWD/nerf-pytorch_amd/lib/variant_$n.so; fi
    rm -rf "$OUT/$n"
    timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d "$OUT/$n" -o p --output-format csv -- python3 tests/diag/bf16x3_speed.py > "$OUT/$n.out" 2> "$OUT/$n.log" || exit 1
    python3 profiles/summarize_pmc.py "$OUT/$n" | grep "fwd_x3i_kernel<0,false>" | cut -d, -f2- | sed "s/^/$n: /"
    grep "^bf16x3  " "$OUT/$n.out" | head -1
done
