cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/sampler
cat > /tmp/hipinit.cc <<'EOC'
#include <hip/hip_runtime.h>
#include <cstdio>
int main() { void* p; hipMalloc(&p, 1 << 20); hipDeviceSynchronize(); hipFree(p); return 0; }
EOC
/opt/rocm/bin/hipcc -O2 /tmp/hipinit.cc -o /tmp/hipinit 2>/dev/null
cd /tmp
t() { local a=$(date +%s.%N); "$@" > /dev/null 2>&1; local b=$(date +%s.%N); python3 -c "print('%.3f' % ($b - $a))"; }
echo "bare HIP program (hipMalloc, sync): $(t ./hipinit) $(t ./hipinit) s"
echo "sampler 3-D default example, 10 steps: $(t ./sampler s0 --nsteps=10) $(t ./sampler s0 --nsteps=10) s"
echo "sampler 32-D 128 T default recipe, 10 steps: $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=10) $(t ./sampler s1 --dim=32 --default_recipe --pt=128 --nsteps=10) s"
echo "sampler 32-D 128 T Gaussian recipe only, 10 steps: $(t ./sampler s2 --dim=32 --pt=128 --nsteps=10) s"
echo "sampler 12-D 64 T default recipe, 10 steps: $(t ./sampler s3 --dim=12 --default_recipe --pt=64 --nsteps=10) s"
