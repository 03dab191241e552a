set -e
for C in 0 1; do
  export PTM_COMPACT=$C
  python tools/kbench.py --walkers 16384 --tag "compact=$C" 2>&1 | tail -1
  python tools/kbench.py --walkers 4096 --tag "compact=$C W=4096" 2>&1 | tail -1
done
unset PTM_COMPACT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharding.py -x -q -m gpu -k "full_size or sharded or bench or 1024 or exchange_decide" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu 2>&1 | tail -1 | cut -c1-1400
