#!/bin/bash
# Evidence of one build, per workload alone (run on the GPU box through gpurun; everything lands under gpurun_out/ev/<tag>/):
#   kernel_stats : rocprofv3 --kernel-trace --stats of the workload
#   pmc_mfma     : SQ_INSTS_VALU_MFMA_F32 / busy / wave cycles (tools/analyze_mfma.py)
#   pmc_fetch / pmc_write : FETCH_SIZE / WRITE_SIZE in passes of their own (tools/analyze_traffic.py)
#   layers       : tools/layer_profile.py table (planning configs)
# usage: tools/collect_evidence.sh <tag> <workload: headline|cfg2|cfg3|unet8192|train> ...
tag=$1; shift
out=gpurun_out/ev/$tag
mkdir -p $out
export TMPDIR=/tmp
( while true; do date >> $out/heartbeat.txt; sleep 30; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
PMC_MFMA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32 SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
for w in "$@"; do
  case $w in
    headline) prog="bench.py --steps 20 --no-cpu-baseline --no-unet-util --no-train"; layers=1 ;;
    cfg2)     prog="tools/bench_other.py 2 3"; layers=2 ;;
    cfg3)     prog="tools/bench_other.py 3 2"; layers=3 ;;
    unet8192) prog="tools/unet_util.py 8192 1,2,4,8 10"; layers= ;;
    train)    prog="tools/bench_train.py --steps 3"; layers= ;;
    *) echo "unknown workload $w"; exit 1 ;;
  esac
  echo "== $w: $prog"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${w}_stats -- python3 $prog > $out/${w}_stats.log 2>&1 || { echo "stats run of $w failed"; exit 1; }
  timeout -s KILL 400 rocprofv3 --pmc $PMC_MFMA --output-format csv -d $out/${w}_pmc_mfma -- python3 $prog > $out/${w}_pmc_mfma.log 2>&1 || { echo "mfma pass of $w failed"; exit 1; }
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${w}_pmc_fetch -- python3 $prog > $out/${w}_pmc_fetch.log 2>&1 || { echo "fetch pass of $w failed"; exit 1; }
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${w}_pmc_write -- python3 $prog > $out/${w}_pmc_write.log 2>&1 || { echo "write pass of $w failed"; exit 1; }
  python3 tools/analyze_mfma.py $out/${w}_pmc_mfma $w > $out/${w}_pmc_mfma.json || exit 1
  python3 tools/analyze_traffic.py $out/${w}_pmc_fetch $out/${w}_pmc_write > $out/${w}_hbm_traffic.json || exit 1
  cp $(find $out/${w}_stats -name "*kernel_stats.csv" | head -1) $out/${w}_kernel_stats.csv || exit 1
  if [ -n "$layers" ]; then
    timeout -k 10 300 python3 -u tools/layer_profile.py $layers $out/${w}_layers.tsv > $out/${w}_layers.txt 2>&1 || { echo "layer profile of $w failed"; exit 1; }
  fi
  # the raw traces are large: keep the summaries only
  rm -rf $out/${w}_stats $out/${w}_pmc_mfma $out/${w}_pmc_fetch $out/${w}_pmc_write
done
echo done
