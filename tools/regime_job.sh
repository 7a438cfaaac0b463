#!/bin/bash
# on the GPU box: three counter passes of tools/regime_probe.py (program directly after `--`), then the two-column table
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/tests
OUT=$PWD/gpurun_out/regime
rm -rf $OUT; mkdir -p $OUT
P1="TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum"
P2="TCP_PENDING_STALL_CYCLES_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY"
P3="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 tools/regime_probe.py ${R:-6} > $OUT/pass$i.txt 2> $OUT/pass$i.err
  cat $OUT/pass$i.txt
  i=$((i+1))
done
python3 tools/regime_diff.py $OUT > $OUT/regime_table.txt 2>&1
cat $OUT/regime_table.txt
# keep what is small enough to come back: the tables, not the per-dispatch csv
find $OUT -name "*.csv" -size +2M -delete
