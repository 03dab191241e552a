cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/sampler
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/lisa
cd /tmp
( time ./sampler run1 --nsteps=20000 --nevery=5000 > /tmp/s1.out ) 2>&1 | grep real
( time ./sampler run2 --nsteps=20000 --nevery=5000 --replicas=64 > /tmp/s2.out ) 2>&1 | grep real
( time ./sampler run3 --nsteps=20000 --nevery=5000 --pt_evolve_rate=0 > /tmp/s3.out ) 2>&1 | grep real
( time ./lisa --outname=l1 --nsteps=5000 --nevery=2500 --pt=20 --gauss_draw_frac=1 > /tmp/l1.out ) 2>&1 | grep real
( time ./lisa --outname=l2 --nsteps=5000 --nevery=2500 --pt=20 > /tmp/l2.out ) 2>&1 | grep real
