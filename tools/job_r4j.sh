#!/bin/bash
# full GPU suite + smoke on the current build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4j; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/tests_all.txt 2>&1; tail -15 $O/tests_all.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
