cd ${GRAFT_REPO_ROOT:-/root/repo}
export PTM_BENCH_REHEARSAL=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --walkers 1024 --steps 10 --warmup 2 > gpurun_out/rehearse2.log 2>&1
echo rc=$? >> gpurun_out/rehearse2.log
