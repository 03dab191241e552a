cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -std=c++11 -O2 -g -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o /tmp/lisa
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in "a --gauss_draw_frac=1" "b "; do
  set -- $v; n=$1; shift
  rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lisa/$n -- /tmp/lisa --outname=l$n --pt=20 --nsteps=5000 --nevery=5000 $@ > /dev/null 2>&1
  for f in $(find $R/gpurun_out/prof_lisa/$n -name "*_stats.csv"); do echo "== $n $(basename $f)"; head -12 $f | cut -c1-160; done
done
