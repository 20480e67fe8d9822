#!/bin/bash
# GPU box: instruction-cache counters of the conv kernels (one encode + optional decode probe).  usage: tools/icache_pmc.sh [ec|dc]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/icpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PB=${PB:-48}
W=${1:-ec}
if [ "$W" = ec ]; then PROBE=$R/tools/ec_probe.py; PAT=cconv16; else PROBE=$R/tools/dc_plane_probe.py; PAT=cconv; fi
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  T=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -d $O/$T -o p --output-format csv -- python3 $PROBE > $O/$T.txt 2>&1
  python3 $R/tools/pmc_agg.py $O/$T/p_counter_collection.csv $PAT > $O/$W.$T.agg.txt; rm -rf $O/$T
done
cat $O/$W.*.agg.txt
