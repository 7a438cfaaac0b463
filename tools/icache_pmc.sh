#!/bin/bash
# instruction-cache counters of the step kernels (own PMC pass)
export TMPDIR=/tmp
rm -rf /tmp/icp; rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d /tmp/icp/pmc1 -- python3 bench.py --steps 30 --warmup 5 --no-c2 --no-cpu-baseline > /dev/null 2>/tmp/icp.err
tail -5 /tmp/icp.err
python3 tools/prof_summary.py /tmp/icp
