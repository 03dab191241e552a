# bench.py --gpus 2 itself on a ONE-GPU box: both ranks on device 0, the messages over gloo (PTM_BENCH_REHEARSAL=1) -- the
# real engine shards, tensors, streams and message pattern, only the transport is not RCCL.  Three runs: the rung-sharded step;
# a pre-flight that fails on every rank; one whose messages "never complete" (the walker-split fallbacks, the second leaving
# through the host-side group and os._exit as it would with a stuck RCCL message).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PTM_BENCH_REHEARSAL=1
L=gpurun_out/rehearse2.log; : > $L
port=29611
for sab in "" fail stall; do
  echo "== PTM_PREFLIGHT_SABOTAGE='$sab'" >> $L
  PTM_PREFLIGHT_SABOTAGE=$sab timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --walkers 1024 --steps 10 --warmup 2 >> $L 2>&1
  echo rc=$? >> $L
  port=$((port + 1))
done
