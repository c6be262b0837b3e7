# usage (GPU box): sweep_chunks.sh [bench args]: chunks per pipe x pipes
for rep in 1 2; do
for c in 0 1 2 4 8; do for p in 4 6; do
SPARTACUS_PIPES=$p SPARTACUS_CHUNKS_PER_PIPE=$c timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunks_per_pipe', $c, 'pipes', $p, '%.3e' % d['value'], '%.2f' % d['ms_per_step'])"
done; done; done
