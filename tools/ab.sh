#!/bin/bash
# Diagnostics: A/B timing of two builds of the library on ONE box, interleaved (run-to-run spread between boxes and between
# processes is ~0.2 ms at the 64-candidate headline): tools/ab.sh <libA.so> <libB.so> [candidates] [rounds]
A=$1; B=$2; N=${3:-64}; R=${4:-3}
for r in $(seq $R); do
  TREBI_LIB=$A timeout -k 10 100 python tools/abl_bench.py $N || exit 1
  TREBI_LIB=$B timeout -k 10 100 python tools/abl_bench.py $N || exit 1
done
