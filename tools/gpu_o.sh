#!/bin/bash
# training runs on the round-2 kernels: the reference's coupled loop (overlapped actor) and the vectorised rollout
set -u
mkdir -p gpurun_out/o
cd gpurun_out/o
timeout 600 python ../../run.py --disable-wandb --num-steps 1500000 --in-embed 64 --overlap --sampler device --eval-frequency 50000 --eval-episodes 20 --seed 1 --project-name r02_overlap > run_overlap.log 2>&1
echo "overlap rc=$? $(tail -1 run_overlap.log)"
timeout 600 python ../../run.py --disable-wandb --num-steps 1500000 --in-embed 64 --num-envs 8 --sampler device --eval-frequency 50000 --eval-episodes 20 --seed 1 --project-name r02_vector8 > run_vector8.log 2>&1
echo "vector rc=$? $(tail -1 run_vector8.log)"
find policies -name "*_results.csv" | while read f; do echo $f; tail -8 "$f"; done
find policies -name "*.csv" -exec sh -c 'cp "$1" "$(echo $1 | cut -d/ -f2)_$(basename "$1" | sed "s/.*_seed=1_//")"' _ {} \;
rm -rf policies
ls
