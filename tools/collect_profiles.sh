#!/bin/bash
# Runs on the GPU box (through gpurun): the rocprofv3 evidence behind profiles/rNN_* -- kernel stats of the default bench,
# kernel stats of the encode / decode / importance-map probes alone on the GPU, and the FETCH_SIZE / WRITE_SIZE passes (each counter in its own run).
# usage: tools/collect_profiles.sh r06      -> gpurun_out/prof_r06/...   (progress lines on stdout: the run takes ~12 minutes)
set -e
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
echo bench done
export PB=64
rocprofv3 --kernel-trace --stats -d $O/dc -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/dc_probe.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/ec -o p --output-format csv -- python3 $R/tools/ec_probe.py > $O/ec_probe.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/imp -o p --output-format csv -- python3 $R/tools/imp_probe.py > $O/imp_probe.txt 2>&1
# the streaming ops at 32 images: kernel-only durations (the event timings of stream_ops.json include the shim's per-call overhead)
SOB_BATCHES=32 rocprofv3 --kernel-trace --stats -d $O/sops -o p --output-format csv -- python3 $R/tools/stream_ops_bench.py > $O/stream_ops_32.json 2> $O/sops.err
echo probes done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d $O/dc_$C -o p --output-format csv -- python3 $R/tools/dc_probe.py > $O/dc_$C.txt 2>&1
  rocprofv3 --kernel-trace --pmc $C -d $O/ec_$C -o p --output-format csv -- python3 $R/tools/ec_probe.py > $O/ec_$C.txt 2>&1
  rocprofv3 --kernel-trace --pmc $C -d $O/imp_$C -o p --output-format csv -- python3 $R/tools/imp_probe.py > $O/imp_$C.txt 2>&1
  echo pmc $C done
done
echo pmc done
cd $R
python3 tools/pmc_traffic.py $O/pmc_traffic.json 64 $O/dc_FETCH_SIZE/p_counter_collection.csv $O/dc_WRITE_SIZE/p_counter_collection.csv \
    $O/ec_FETCH_SIZE/p_counter_collection.csv $O/ec_WRITE_SIZE/p_counter_collection.csv \
    $O/imp_FETCH_SIZE/p_counter_collection.csv $O/imp_WRITE_SIZE/p_counter_collection.csv
python3 tools/stream_ops_bench.py > $O/stream_ops.json
# keep only the summaries (the traces are hundreds of MB)
for d in dc ec imp; do cp $O/$d/p_kernel_stats.csv $O/${d}_kernel_stats.csv; done
# the bench pass also runs the transforms once: MIOpen's find-mode search (naive_conv_* reference kernels) is not part of any timed region
python3 tools/stats_without_find.py $O/bench/p_kernel_stats.csv $O/bench_kernel_stats.csv
cp $O/sops/p_kernel_stats.csv $O/stream_ops_kernel_stats.csv
rm -rf $O/bench $O/dc $O/ec $O/imp $O/sops $O/*_FETCH_SIZE $O/*_WRITE_SIZE
ls -la $O
