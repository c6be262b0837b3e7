for n in 100000 200000 400000 1000000; do
  for p in 1 4; do
    echo "== columns $n pipes $p"
    SPARTACUS_PIPES=$p bash tools/quick_bench.sh --no-extras --workload c3ns4 --columns $n
  done
done
