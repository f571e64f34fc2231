# A/B of experimental builds on the dense path: tools/dense_abl.sh "k ..." name ...   (rust-debruijn_amd/_exp/lib<name>.so; main = the regular build)
KS=$1; shift
for n in "$@"; do for k in $KS; do
  L=$PWD/rust-debruijn_amd/_exp/lib$n.so; [ $n = main ] && L=$PWD/rust-debruijn_amd/libdbg_mi355x.so
  DBG_LIB=$L DBG_PATH=dense python bench.py --k $k --reads 20000000 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --summarizer ${SUMM:-count} 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$n k=$k', d['value'], d['roofline']['kernel_ms_per_step'])"
done; done
