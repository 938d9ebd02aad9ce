#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun): kernel trace + stats of the default bench.py
# command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE - they do not fit one pass, MI355X_MICROARCH.md §rocprofv3
# PMC slots - and one pass with the MFMA counters).  The raw counter CSVs are large, so only per-kernel summaries are
# kept under gpurun_out/prof_<tag>/ (copy that directory to profiles/<tag>/ to commit it).
#   tools/prof.sh <tag> [workload]
set -o pipefail
export TMPDIR=/tmp
tag=${1:-r02}
wl=${2:-cfg3_b128_l4}
out=gpurun_out/prof_${tag}
raw=/tmp/prof_raw_${tag}
mkdir -p $out $raw
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/trace -- python3 bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --proxy-steps 0 --other-workloads "" --ingraph-steps 0 > $out/trace.log 2>&1 || echo "trace run failed"
grep "^{\"metric" $out/trace.log > $out/bench_under_rocprof.json
echo "trace done" > $out/progress.txt
if [ ! -s $out/bench_under_rocprof.json ]; then echo "the profiled bench run printed no result line:"; tail -30 $out/trace.log; exit 1; fi
if [ "$TRACE_ONLY" = "1" ]; then python3 tools/prof_summarize.py $raw $out $wl; ls -la $out; exit 0; fi
for c in FETCH_SIZE WRITE_SIZE; do
  if ! SINGA_CALIB=1 timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $raw/pmc_$c -- python3 bench.py --workload $wl --eager --steps 2 --warmup 1 --roofline-steps 1 --no-cpu-baseline --proxy-steps 0 --other-workloads "" --ingraph-steps 0 > $out/pmc_$c.log 2>&1; then
    echo "pmc $c run failed:"; tail -30 $out/pmc_$c.log; exit 1
  fi
  echo "pmc $c done" >> $out/progress.txt
done
if ! timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $raw/pmc_MFMA -- python3 bench.py --workload $wl --eager --steps 2 --warmup 1 --roofline-steps 0 --no-cpu-baseline --proxy-steps 0 --other-workloads "" --ingraph-steps 0 > $out/pmc_MFMA.log 2>&1; then
  echo "pmc MFMA run failed:"; tail -30 $out/pmc_MFMA.log; exit 1
fi
echo "pmc MFMA done" >> $out/progress.txt
python3 tools/prof_summarize.py $raw $out $wl
ls -la $out
