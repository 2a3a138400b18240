#!/bin/bash
# Builds alternative libschnorr_sig_amd.so variants (compile-time switches of the kernels) under build/variants/
# for A/B measurements on the GPU box:  SSA_LIB=build/variants/<name>.so python bench.py --skip-torsion-leg ...
# usage: tools/build_variants.sh name1:"-DFLAG1 -DFLAG2" name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
    name=${spec%%:*}
    flags=${spec#*:}
    echo "building $name ($flags)"
    (
        for u in ssa_api ssa_msm ssa_sign; do
            hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c -cuid=$u $flags -o build/variants/$name.$u.o \
                schnorr-sig_amd/csrc/$u.hip &
        done
        wait
        hipcc --offload-arch=gfx950 -fPIC -shared -o build/variants/$name.so build/variants/$name.ssa_api.o \
            build/variants/$name.ssa_msm.o build/variants/$name.ssa_sign.o && rm -f build/variants/$name.ssa_*.o
    ) &
done
wait
ls -la build/variants
