# wall time of the facade's sampler on small problems (device proposals, host likelihood, the default host-proposal recipe);
# per-step cost = difference of two run lengths.  PTM_SHARED_HANDOVER=0 gives the copies back for comparison.
cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/sampler
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/lisa
cd /tmp
t() { local a=$(date +%s.%N); "$@" > /dev/null 2>&1; local b=$(date +%s.%N); python3 -c "print('%.3f' % ($b - $a))"; }
per_step() {  # name, steps1, steps2, command... (--nsteps appended)
  local name=$1 n1=$2 n2=$3; shift 3
  local t1=$(t "$@" --nsteps=$n1 --nevery=$n2) t2=$(t "$@" --nsteps=$n2 --nevery=$n2)
  echo "$name: $(python3 -c "print('%.1f' % (($t2 - $t1) * 1e6 / ($n2 - $n1)))") us per step  ($n1 steps $t1 s, $n2 steps $t2 s)"
}
for Z in 1 0; do
  export PTM_SHARED_HANDOVER=$Z; echo "PTM_SHARED_HANDOVER=$Z"
  per_step "sampler, device target"        5000 25000 ./sampler run1
  per_step "lisa, device proposals"        2000 10000 ./lisa --outname=l1 --pt=20 --gauss_draw_frac=1
  per_step "lisa, default recipe"          2000 10000 ./lisa --outname=l2 --pt=20
  per_step "lisa, default recipe, 128 T"   2000 6000 ./lisa --outname=l3 --pt=128
  # (the default recipe's differential evolution is drawn on the device since round 4; PTM_HOST_DE=1 keeps the host-proposal path)
  PTM_HOST_DE=1 per_step "lisa, default recipe, DE on the host"          2000 10000 ./lisa --outname=l4 --pt=20
  PTM_HOST_DE=1 per_step "lisa, default recipe, 128 T, DE on the host"   2000 6000 ./lisa --outname=l5 --pt=128
done
