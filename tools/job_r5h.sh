#!/bin/bash
# stability: the complete GPU suite twice more (fresh box)
O=gpurun_out/r5h; mkdir -p $O
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/tests_$i.log 2>&1; tail -2 $O/tests_$i.log; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
