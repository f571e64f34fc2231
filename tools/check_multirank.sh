#!/bin/bash
# Functional check of the N > 1 path on a one-GPU box: 2, 3 and 8 ranks share cuda:0 (gloo, host-staged payload), started the
# way a user would (`python bench.py --gpus N`, which re-executes itself under torch.distributed.run).  The per-rank tables
# must add up to the single-GPU table over the same reads: valid k-mer count AND the order-independent table digest
# (keys, Exts, label lists of every row; every k-mer lives on exactly one rank).
set -e
R=${1:-2000000}
one() { timeout 300 python bench.py --reads $1 --steps 1 --warmup 0 --no-cpu-baseline --compress-reads 0 --no-other-shapes --digest 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['valid_kmers_all_ranks'], d['table_digest'])"; }
many() { timeout 900 python bench.py --gpus $1 --reads $R --steps 1 --warmup 0 --no-cpu-baseline --backend gloo --one-device --digest 2>&1 | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); assert d['n_gpus'] == $1 and d['ranks_seen'] == $1, d; b=d['balance']; assert b and b['records_owned']['min'] > 0 and len(b['exchange_exposed_ms_by_round_max_rank']) == d['exchange']['rounds'], b; print(d['config']['valid_kmers_all_ranks'], d['table_digest'])"; }
for W in ${WORLDS:-2 3 8}; do          # 8 = the target node shape: eight ranks, here all on one GPU
  a=$(one $((R * W))); b=$(many $W)
  echo "world $W: single-GPU valid,digest=$a  sharded=$b"
  [ "$a" == "$b" ] || { echo MISMATCH; exit 1; }
  # the sender-side duplicate merge the other way round (default: on at 2 ranks, off above)
  m=$([ $W -le 2 ] && echo 0 || echo 1)
  b=$(DBG_SHARD_MERGE=$m many $W)
  echo "world $W, DBG_SHARD_MERGE=$m: sharded=$b"
  [ "$a" == "$b" ] || { echo MISMATCH; exit 1; }
done
echo multirank ok
