#!/bin/bash
# round 5, GPU session 9: the multi-rank bench path on a one-GPU box (both ranks on cuda:0, gloo as the control plane): start-up diagnostics, the line
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/s9; mkdir -p $OUT
DTQN_DIST_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-env-rate > $OUT/bench2.json 2> $OUT/bench2.err
echo "rc=$?"; grep "^\[bench\]" $OUT/bench2.err; tail -3 $OUT/bench2.err | cut -c1-300
python -c "
import json
d=json.loads([l for l in open('$OUT/bench2.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['ms_per_step'], d.get('exchange',{}).get('kind'), d.get('exchange',{}).get('weak_scaling_efficiency'), d.get('pipeline'))"
DTQN_DP_INJECT=mapping:1 DTQN_DIST_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 50 --warmup 10 --no-other-configs --no-cpu-baseline --no-env-rate > $OUT/bench2i.json 2> $OUT/bench2i.err
echo "rc=$?"; grep "^\[bench\]\|falling back" $OUT/bench2i.err | cut -c1-300
timeout 200 python -m pytest tests/test_bench_contract.py -q -p no:cacheprovider 2>&1 | tail -2
