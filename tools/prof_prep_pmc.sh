#!/bin/bash
# SQ / TA / TCP counters of level0_prep at 4K (separate --pmc passes): bash tools/prof_prep_pmc.sh <tag>
tag=${1:-prep_pmc}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters.txt 2>&1
grep -o -E "\b(SQ_WAIT[A-Z_]*|SQ_ACTIVE_INST[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|SQ_BUSY[A-Z_]*|SQ_WAVE[A-Z_]*|TA_[A-Z_]*|TD_[A-Z_]*|TCP_[A-Z_]*)\b" $out/counters.txt | sort -u > $out/names.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $out/a -- python3 $root/tools/one_prep.py > $out/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VALU --output-format csv -d $out/b -- python3 $root/tools/one_prep.py > $out/b.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $out/c -- python3 $root/tools/one_prep.py > $out/c.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $out/d -- python3 $root/tools/one_prep.py > $out/d.log 2>&1
for s in a b c d; do echo "== $s"; python3 $root/tools/pmc_kernel.py $out/$s level0_prep 2>&1 | tail -12; tail -2 $out/$s.log; done
