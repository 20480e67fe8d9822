#!/bin/bash
# GPU box: FETCH_SIZE / TCC hit counters of the decode probe for the current build.  usage: tools/xd_pmc.sh TAG
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/xdpmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PB=${PB:-48}
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/f.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/t -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/t.txt 2>&1
cd $R
for d in f t; do python3 tools/pmc_agg.py $O/$d/p_counter_collection.csv cconv1 > $O/$d.agg.txt 2>&1; done
rm -rf $O/f $O/t
cat $O/*.agg.txt
