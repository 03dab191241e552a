set -e
for P in 0 -1 2 3 4; do
  if [ "$P" = "-1" ]; then unset PTM_PERSIST; else export PTM_PERSIST=$P; fi
  python tools/kbench.py --walkers 16384 --tag "persist=$P" 2>&1 | tail -1
done
unset PTM_PERSIST
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mfma or bit_exact or full_size" 2>&1 | tail -3
