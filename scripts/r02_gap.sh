#!/bin/bash
# host phases of a step (FALCON_AMD_TIMING) in pipelined and unpipelined bench steps
tag=${1:-gap}; out=gpurun_out/$tag; mkdir -p $out
FALCON_AMD_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 > $out/pipe.json 2> $out/pipe.err
FALCON_AMD_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 --no-pipeline > $out/serial.json 2> $out/serial.err
grep "align + msa plan" $out/pipe.err | tail -4; grep "align + msa plan" $out/serial.err | tail -3
python -c "
import json
for f in ('pipe','serial'):
    d=json.loads(open('$out/'+f+'.json').readline()); print(f, d['value'], d['ms_per_step'], d['kernel_ms'])"
