for mode in 2048 0 256 2048 0 256 2048 0 256; do
  DBG_SLAB_VMM=$mode timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-boundary --compress-reads 0 --no-other-shapes --k 31 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('VMM=$mode k=31:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step']['sk_scan'])"
done
