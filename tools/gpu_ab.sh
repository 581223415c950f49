#!/bin/bash
# On the GPU box: interleaved A/B of the shipped library against a variant build + do all traced tensors and
# gradients hash identically?   usage: tools/gpu_ab.sh <variant-name> [families] [AB_ARGS...]
# (variant = pointvs_amd/libpvs_egnn_<name>.so from tools/variant_obj.sh or make variant)
name=$1; fam=${2:-default}; shift; shift
out=gpurun_out/ab_$name
mkdir -p $out
AB_ROUNDS=${AB_ROUNDS:-3} AB_ARGS="$*" python3 tools/ab.py base=pointvs_amd/libpvs_egnn.so $name=pointvs_amd/libpvs_egnn_$name.so > $out/ab.txt 2>&1
cat $out/ab.txt
python3 tools/hash_outputs.py --families $fam > $out/hash_base.json 2> $out/hash_base.err
PVS_EGNN_LIB=$PWD/pointvs_amd/libpvs_egnn_$name.so python3 tools/hash_outputs.py --families $fam > $out/hash_var.json 2> $out/hash_var.err
cmp -s $out/hash_base.json $out/hash_var.json && echo "HASHES_IDENTICAL ($fam)" || echo "HASHES_DIFFER ($fam)"
