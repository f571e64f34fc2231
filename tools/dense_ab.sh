for k in 8 9; do for m in 1 0; do
DBG_DENSE_RANGES=$m DBG_PATH=dense timeout 300 python bench.py --k $k --reads 20000000 --no-cpu-baseline --no-host-boundary --compress-reads 0 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('k=$k ranges=$m', d['value'], 'Gkmer/s', d['ms_per_step'], 'ms', d['roofline']['kernel_ms_per_step'], 'valid', d['config']['valid_kmers_rank0'])"
done; done
