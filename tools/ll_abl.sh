for n in ${ABLS:-1 2 3 4 main}; do
  L=$PWD/rust-debruijn_amd/_exp/libllabl$n.so; [ $n = main ] && L=$PWD/rust-debruijn_amd/libdbg_mi355x.so
  echo "== $n"; DBG_LIB=$L timeout 200 python tools/bench_labels.py --reads 20000000 --iters 2 5000 2>&1 | grep labels
done
