cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/sampler
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_de; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s3 -- /tmp/sampler s3 --dim=32 --default_recipe --pt=128 --nsteps=4000 --nevery=4000 > $O/s3.out 2> $O/s3.err
f=$(find $O/s3 -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-200
