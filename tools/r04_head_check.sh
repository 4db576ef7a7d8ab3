#!/bin/bash
# HEAD check after the r04z evidence call (commits since: tests, the handshake fix, the launcher's attribute call): GPU suite, smoke, default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q > $out/gputests.log 2>&1; echo "pytest rc=$?" >> $out/gputests.log; tail -4 $out/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py > $out/bench_14B_720p.json 2> $out/bench.err; tail -c 700 $out/bench_14B_720p.json
