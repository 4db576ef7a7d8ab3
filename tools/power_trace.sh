#!/bin/bash
# dev: sample the GPU's power draw and shader clock (rocm-smi) while a command runs; output under gpurun_out/<tag>.
#   bash tools/power_trace.sh <tag> <command...>
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $(dirname $out)
"$@" > $out.cmd.log 2>&1 &
pid=$!
: > $out.smi.log
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showpower --showclocks --showuse --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|GPU use|Temperature \(Sensor (junction|edge)"; } >> $out.smi.log
  sleep 0.4
done
wait $pid; echo "rc=$?" >> $out.cmd.log
python3 - $out.smi.log <<'P'
import re, sys
pw, ck = [], []
for line in open(sys.argv[1]):
    m = re.search(r"Power \(W\):\s*([\d.]+)", line)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", line)
    if m: ck.append(int(m.group(1)))
def q(v, f): v = sorted(v); return v[int(f * (len(v) - 1))] if v else None
print("samples", len(pw), "power W p10/p50/p90/max", q(pw, .1), q(pw, .5), q(pw, .9), max(pw or [0]), " sclk MHz p10/p50/p90/max", q(ck, .1), q(ck, .5), q(ck, .9), max(ck or [0]))
P
tail -3 $out.cmd.log
