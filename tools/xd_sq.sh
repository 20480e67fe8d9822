#!/bin/bash
# GPU box: SQ counters of the decode probe for the current build.  usage: tools/xd_sq.sh TAG
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/xdsq_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PB=${PB:-48}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $O/a -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/a.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM -d $O/b -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/b.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/c -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/c.txt 2>&1
cd $R
for d in a b c; do python3 tools/pmc_agg.py $O/$d/p_counter_collection.csv cconv1 > $O/$d.agg.txt 2>&1; done
rm -rf $O/a $O/b $O/c
cat $O/*.agg.txt
