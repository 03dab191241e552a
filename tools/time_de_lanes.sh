# differential evolution on the device, through the facade's sampler (per-step cost = difference of two run lengths):
#   default                          the persistent ladder kernel's builds with differential evolution (9..32 dimensions, FL = 11 / 15)
#   PTM_LADDER=0                     two launches per step, the lanes kernel (a lane per dimension)
#   PTM_LADDER=0 PTM_FORCE_VALU=1    two launches per step, the general kernel (a lane per chain: the only device form before)
#   PTM_HOST_DE=1                    drawn on the host (the host-proposal path)
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
./sampler s0 --dim=12 --default_recipe --pt=64 --nsteps=200 | tail -1
for V in "1 0" "0 0" "0 1"; do
  set -- $V; export PTM_LADDER=$1 PTM_FORCE_VALU=$2; echo "PTM_LADDER=$1 PTM_FORCE_VALU=$2"
  per_step "sampler, device target, 12 dimensions, 64 T, default recipe"   2000 10000 ./sampler s1 --dim=12 --default_recipe --pt=64
  per_step "sampler, device target, 17 dimensions, 128 T, default recipe"  2000 6000 ./sampler s2 --dim=17 --default_recipe --pt=128
  per_step "sampler, device target, 32 dimensions, 128 T, default recipe"  2000 6000 ./sampler s3 --dim=32 --default_recipe --pt=128
  per_step "lisa, default recipe, 20 T"    2000 10000 ./lisa --outname=l2 --pt=20
  per_step "lisa, default recipe, 128 T"   2000 6000 ./lisa --outname=l3 --pt=128
done
unset PTM_FORCE_VALU PTM_LADDER
PTM_HOST_DE=1 per_step "sampler, device target, 12 dimensions, 64 T, default recipe, DE on the host"   2000 6000 ./sampler s4 --dim=12 --default_recipe --pt=64
PTM_HOST_DE=1 per_step "sampler, device target, 32 dimensions, 128 T, default recipe, DE on the host"  1000 3000 ./sampler s5 --dim=32 --default_recipe --pt=128
