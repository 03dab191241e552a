# the LISA example with 64 replicas of a 20-temperature ladder: 1280 host likelihood calls per step -- the facade's thread pool
# (sized by the processors the process may really use, switched off by itself when it does not pay) against one thread
cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/lisa
cd /tmp
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)))"
t() { local a=$(date +%s.%N); "$@" > /dev/null 2>&1; local b=$(date +%s.%N); python3 -c "print('%.3f' % ($b - $a))"; }
per_step() {  # name, steps1, steps2, command... (--nsteps appended)
  local name=$1 n1=$2 n2=$3; shift 3
  local t1=$(t "$@" --nsteps=$n1 --nevery=$n2) t2=$(t "$@" --nsteps=$n2 --nevery=$n2)
  echo "$name: $(python3 -c "print('%.1f' % (($t2 - $t1) * 1e6 / ($n2 - $n1)))") us per step  ($n1 steps $t1 s, $n2 steps $t2 s)"
}
per_step "lisa, default recipe, 20 T x 64 replicas, pool as the facade sizes it"  1000 4000 ./lisa --outname=l4 --pt=20 --replicas=64
PTM_EVAL_THREADS=1 per_step "lisa, default recipe, 20 T x 64 replicas, one thread"  1000 4000 ./lisa --outname=l5 --pt=20 --replicas=64
PTM_EVAL_THREADS=8 per_step "lisa, default recipe, 20 T x 64 replicas, 8 threads"  1000 4000 ./lisa --outname=l6 --pt=20 --replicas=64
PTM_EVAL_THREADS=16 per_step "lisa, default recipe, 20 T x 64 replicas, 16 threads"  1000 4000 ./lisa --outname=l6 --pt=20 --replicas=64
per_step "lisa, default recipe, 128 T"   2000 6000 ./lisa --outname=l3 --pt=128
PTM_EVAL_THREADS=1 per_step "lisa, default recipe, 128 T, one thread"   2000 6000 ./lisa --outname=l3 --pt=128
PTM_EVAL_SPIN_US=200 per_step "lisa, default recipe, 128 T, workers that spin 200 us for their next batch"   2000 6000 ./lisa --outname=l3 --pt=128
per_step "lisa, default recipe, 20 T"   2000 10000 ./lisa --outname=l3 --pt=20
