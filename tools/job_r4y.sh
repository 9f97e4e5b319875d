#!/bin/bash
# full GPU suite + smoke on the current build
O=gpurun_out/r4y; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/tests_full.log 2>&1; tail -5 $O/tests_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
