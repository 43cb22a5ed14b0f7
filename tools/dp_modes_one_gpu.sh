#!/bin/bash
# run.py under two ranks on a ONE-GPU box (DTQN_DIST_SAME_DEVICE=1: both ranks on cuda:0, gloo collectives) in every loop mode:
# a rank that enters a collective alone hangs, so every run sits under a timeout.  Prints one OK / FAIL line per mode.
cd "$GRAFT_REPO_ROOT" || exit 1
export DTQN_DIST_SAME_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1"
R="$PWD/run.py --disable-wandb --in-embed 64 --num-steps 2500 --prepopulate 4000 --eval-frequency 1000 --eval-episodes 2 --sampler device --verbose"
mode() {   # name, port, pattern, extra args...
  name=$1; port=$2; pat=$3; shift 3
  d=gpurun_out/dpmodes/$name; rm -rf $d; mkdir -p $d
  (cd $d && timeout 150 $T --master-port $port $R "$@" > log.txt 2>&1; echo $? > rc.txt)
  n=$(grep -c "$pat" $d/log.txt); rc=$(cat $d/rc.txt)
  if [ "$rc" = 0 ] && [ "$n" -ge 1 ]; then echo "OK   $name"; else echo "FAIL $name rc=$rc matches=$n"; tail -5 $d/log.txt; fi
}
mode overlap 29701 "Training Steps: 2000" --overlap
mode vector4 29702 "Training Steps: 2000" --num-envs 4
DTQN_DP_EXCHANGE=p2p mode p2p_overlap 29703 "Training Steps: 2000" --overlap
mode reference_sampler 29704 "Training Steps: 2000" --sampler reference
mode time_limit 29705 "Reached time limit" --num-steps 2000000 --time-limit 0.003
