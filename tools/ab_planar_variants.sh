#!/usr/bin/env bash
# Same-box A/B of generator variants of the planar 3x3 assembly (tools/build_gen_variant.py NAME AQ_GEN_...=1 ...):
#   tools/ab_planar_variants.sh OUTDIR NAME1 NAME2 ...     ("base" = the in-tree library)
# per variant: the planar parity tests through that library, then tools/time_conv3x3.py (cold operands, stamped phases), twice, interleaved
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
    lib=build/variant_$v/libaqengine.so
    [ "$v" = base ] && lib=aquaculture_amd/csrc/libaqengine.so
    AQ_ENGINE_LIB=$PWD/$lib timeout -k 10 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu \
        -k "(planar_conv3x3_matches_reference and 13pm) or planar_builds_are_bit_identical or planar_conv3x3_in_engine" > "$out/parity_$v.log" 2>&1
    echo "parity $v: $(tail -1 "$out/parity_$v.log")"
done
for rep in 1 2; do
    for v in "$@"; do
        lib=build/variant_$v/libaqengine.so
        [ "$v" = base ] && lib=aquaculture_amd/csrc/libaqengine.so
        timeout -k 10 300 python tools/time_conv3x3.py --nbs 13 --reps 40 --stamp --old "" --lib "$PWD/$lib" 2>/dev/null | grep -E "NB=13 asm:|stamped asm NB=13 ABL=0|planar auto" | sed "s/^/$v rep $rep: /" | tee -a "$out/timing.txt"
    done
done
