cd "$GRAFT_REPO_ROOT" || exit 1
export DTQN_DIST_SAME_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { name=$1; shift
  d=gpurun_out/s16/$name; rm -rf $d; mkdir -p $d
  (cd $d && env "$@" timeout 45 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $PORT $GRAFT_REPO_ROOT/run.py --disable-wandb --in-embed 64 --num-steps 2500 --prepopulate 4000 --eval-frequency 1000 --eval-episodes 2 --sampler device --verbose --overlap > log.txt 2>&1; echo "$name rc=$? steps2000=$(grep -c 'Training Steps: 2000' log.txt)")
}
PORT=29721 run base X=1
PORT=29722 run nopipe DTQN_PIPELINE=0
PORT=29723 run actor2 DTQN_ACTOR_SLICES=2
PORT=29724 run rccl DTQN_DP_EXCHANGE=rccl
PORT=29725 run inline DTQN_PIPELINE=inline
