for rep in 1 2; do
for mb in 64 128 256 512 1024 4096; do for p in 4 8; do
SPARTACUS_PIPES=$p SPARTACUS_SCRATCH_MB=$mb timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scratch_mb', $mb, 'pipes', $p, '%.3e' % d['value'], '%.2f' % d['ms_per_step'])"
done; done; done
