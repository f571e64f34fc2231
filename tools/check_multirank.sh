#!/bin/bash
# Functional check of the N > 1 path on a one-GPU box: 2 and 3 ranks share cuda:0 (gloo, host-staged payload).
# The sum of the per-rank valid k-mer counts must equal the single-GPU count over the same reads.
set -e
R=${1:-2000000}
one() { timeout 300 python bench.py --reads $1 --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['valid_kmers_all_ranks'])"; }
many() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + $1)) bench.py --gpus $1 --reads $R --steps 1 --warmup 0 --no-cpu-baseline --backend gloo --one-device 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['valid_kmers_all_ranks'], d['n_gpus'])"; }
for W in 2 3; do
  a=$(one $((R * W))); b=$(many $W)
  echo "world $W: single-GPU valid=$a  sharded valid,ranks=$b"
  [ "$a" == "$(echo $b | cut -d' ' -f1)" ] || { echo MISMATCH; exit 1; }
done
echo multirank ok
