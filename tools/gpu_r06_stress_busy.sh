#!/bin/bash
# the stress runs of gpu_r06_stress.sh beside busy host cores (what a step gets next to other ranks or a parent process)
B=${BUSY:-14}
pids=""
for i in $(seq 1 $B); do (while :; do :; done) & pids="$pids $!"; done
tools/gpu_r06_stress.sh "$@"
kill $pids
