#!/bin/bash
# Runs on the GPU box: SQ / cache counters of the decode probe (48 images), one counter set per run.  usage: tools/dc_pmc.sh TAG [env...]
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PB=${PB:-48}
rocprofv3 --kernel-trace --stats -d $O/st -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/st.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $O/sq1 -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/sq1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/sq2 -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/sq2.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/tcc.txt 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum -d $O/tcp -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/tcp.txt 2>&1
cd $R
for d in sq1 sq2 tcc tcp; do
  f=$O/$d/p_counter_collection.csv
  [ -f $f ] && python3 tools/pmc_agg.py $f cconv > $O/$d.agg.txt 2>&1
done
head -12 $O/st/p_kernel_stats.csv > $O/kernel_stats_head.csv 2>/dev/null
rm -rf $O/st $O/sq1 $O/sq2 $O/tcc $O/tcp
cat $O/*.agg.txt $O/kernel_stats_head.csv
