#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box through gpurun): kernel trace + stats, then separate PMC
# passes (FETCH_SIZE, WRITE_SIZE - they do not fit one pass, MI355X_MICROARCH.md §rocprofv3 PMC slots).  The raw
# counter CSVs are large, so only per-kernel summaries are kept under gpurun_out/prof_<tag>/.
set -o pipefail
export TMPDIR=/tmp
tag=${1:-r01}
out=gpurun_out/prof_${tag}
raw=/tmp/prof_raw_${tag}
mkdir -p $out $raw
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1 || echo "trace run failed"
for c in FETCH_SIZE WRITE_SIZE; do
  SINGA_CALIB=1 timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $raw/pmc_$c -- python3 bench.py --eager --steps 2 --warmup 1 --roofline-steps 0 --no-cpu-baseline > $out/pmc_$c.log 2>&1 || echo "pmc $c run failed"
done
python3 tools/prof_summarize.py $raw $out
ls -la $out
