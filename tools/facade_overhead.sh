cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/sampler
cd /tmp
t() { local a=$(date +%s.%N); "$@" > /dev/null 2>&1; local b=$(date +%s.%N); python3 -c "print('%.3f' % ($b - $a))"; }
for ne in 500 2000 8000; do
  echo "nevery=$ne: 8000 steps $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=8000 --nevery=$ne) s;  2000 steps $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=2000 --nevery=$ne) s"
done
echo "pt_dump_n=0?"; echo "nskip=100: 8000 steps $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=8000 --nevery=8000 --nskip=100) s"
echo "save_every=10: 8000 steps $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=8000 --nevery=8000 --save_every=10) s"
