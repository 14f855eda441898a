#!/bin/bash
# On the GPU box: every diagnostic build through run.py, one process each; output to gpurun_out/alpha_dual_probe.jsonl
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $ROOT/gpurun_out
: > $ROOT/gpurun_out/alpha_dual_probe.jsonl
for v in inline call auto byval fence; do
  MO_LIB_PATH=$ROOT/tools/alpha_dual_probe/lib/libminiopt_$v.so timeout -k 10 300 python3 $ROOT/tools/alpha_dual_probe/run.py $v >> $ROOT/gpurun_out/alpha_dual_probe.jsonl 2>> $ROOT/gpurun_out/alpha_dual_probe.err || echo "{\"build\": \"$v\", \"failed\": true}" >> $ROOT/gpurun_out/alpha_dual_probe.jsonl
done
cat $ROOT/gpurun_out/alpha_dual_probe.jsonl
