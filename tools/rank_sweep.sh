#!/bin/bash
# Diagnostics: per-layer times of one config with every layer's r-th best tile by the chooser's model (TREBI_TILE_RANK=r).
# usage: tools/rank_sweep.sh <config 1|2|3> <max rank> <out dir>
cfg=$1; maxr=$2; out=$3
mkdir -p $out
for r in $(seq 0 $maxr); do
  TREBI_TILE_RANK=$r timeout -k 10 300 python -u tools/layer_profile.py $cfg $out/rank$r.tsv > $out/rank$r.txt 2>&1 || exit 1
done
