for c in 8 16 32 64 128 256; do
  timeout -k 10 150 python bench.py --no-side --cpu-pairs 0 --pair-chunk $c > gpurun_out/chunk_$c.log 2>&1 || exit 1
done
