#!/bin/bash
# GPU box: FETCH_SIZE / WRITE_SIZE of the encode probe's kernels (one counter per run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/ecpmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp PB=48
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $C -d $O/$C -o p --output-format csv -- python3 $R/tools/ec_probe.py > $O/$C.txt 2>&1; python3 $R/tools/pmc_agg.py $O/$C/p_counter_collection.csv cconv16 > $O/$C.agg.txt; rm -rf $O/$C; done
cat $O/*.agg.txt
