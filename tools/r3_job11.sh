#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$PWD/gpurun_out/r3k; mkdir -p $O; export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o t -- python $R/tools/train_bench.py --steps 3 --warmup 2 > $O/train_line.json 2> $O/train.err
cd $R; head -40 $O/train/t_kernel_stats.csv | cut -c1-200; cat $O/train_line.json | cut -c1-600; rm -f $O/train/t_kernel_trace.csv
