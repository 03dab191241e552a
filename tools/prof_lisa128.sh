# the LISA example at 128 temperatures with the default recipe: where a step's ~100 us go (kernels and the HIP calls between them)
cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -g -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/lisa
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_lisa128
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lisa128 -- /tmp/lisa --outname=l128 --pt=128 --nsteps=4000 --nevery=4000 > /dev/null 2>&1
for f in $(find $R/gpurun_out/prof_lisa128 -name "*_stats.csv"); do echo "== $(basename $f)"; head -14 $f | cut -c1-170; done
