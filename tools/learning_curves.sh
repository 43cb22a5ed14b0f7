#!/bin/bash
# round 3: steps-to-success of the rollout modes on live DiscreteCarFlag-v0: run.py to 1.5 M steps per (mode, seed),
# two runs at a time (the loops are host-bound; one run keeps the GPU ~2/3 busy).  CSVs -> gpurun_out/curves/<mode>_seed<k>_{results,losses}.csv
#   [NUM_STEPS=2000000 PAR=3] bash tools/learning_curves.sh "overlap:2 vector8:2 overlap:3 vector8:3"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/curves
run_one() {
  mode=${1%%:*}; seed=${1##*:}
  case $mode in
    serial) flags="" ;;
    overlap) flags="--overlap" ;;
    vector8) flags="--num-envs 8" ;;
    refq) flags="--overlap --sampler reference --ref-quirks" ;;     # the reference's data stream: Python `random` window draws, int-truncated actor context
  esac
  d=$(mktemp -d)
  ( cd $d && cp -r "$GRAFT_REPO_ROOT"/{run.py,dtqn_amd,include} . 2>/dev/null
    t0=$(date +%s)
    timeout ${RUN_TIMEOUT:-900} python run.py --disable-wandb --num-steps ${NUM_STEPS:-1500000} --in-embed 64 $( [[ "$flags" == *--sampler* ]] || echo --sampler device ) $flags --eval-frequency 50000 --eval-episodes 20 --seed $seed > log.txt 2>&1
    echo "$mode seed $seed rc=$? wall=$(( $(date +%s) - t0 ))s" )
  for f in $(find $d -name '*results.csv' -o -name '*losses.csv'); do
    k=results; [[ $f == *losses* ]] && k=losses
    cp $f "$GRAFT_REPO_ROOT/gpurun_out/curves/${mode}_seed${seed}_$k.csv"
  done
  tail -2 $d/log.txt
}
set -- $1
PAR=${PAR:-2}                      # runs at a time
while [ $# -gt 0 ]; do
  pids=""
  for k in $(seq 1 $PAR); do
    [ $# -gt 0 ] || break
    run_one $1 & pids="$pids $!"
    shift
  done
  wait $pids
done
ls -la gpurun_out/curves
