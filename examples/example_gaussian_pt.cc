// example_gaussian_pt.cc -- the reference's PT-on-a-correlated-Gaussian set-up (cython/exampleGaussian.py main(),
// BASELINE.md scratch driver) written against ptmcmc_gpu.hh: same classes, same call sequence, every step on the MI355X.
//   build: g++ -std=c++11 -O2 -Iinclude -Iptmcmc_amd/host examples/example_gaussian_pt.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -pthread
//   usage: example_gaussian_pt [mode] [D] [Ntemps] [nsteps] [chainfile]      mode = device | callback
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

struct target { int D; std::vector<double> P; double like0; long calls; };
// a user likelihood in the reference's function-pointer form (bayesian.hh:547)
static double my_loglike(void* object, const state& s) {
  target* t = (target*)object;
  __sync_fetch_and_add(&t->calls, 1);   // (evaluated from several host threads)
  double q = 0;
  for (int i = 0; i < t->D; i++)
    for (int j = 0; j < t->D; j++) q += s.get_param(i) * t->P[i * t->D + j] * s.get_param(j);
  return t->like0 - 0.5 * q;
}

int main(int argc, char** argv) {
  const bool callback = argc > 1 && !strcmp(argv[1], "callback");
  const int D = argc > 2 ? atoi(argv[2]) : 4, Nt = argc > 3 ? atoi(argv[3]) : 8, nsteps = argc > 4 ? atoi(argv[4]) : 2000;
  // tridiagonal precision: P = tridiag(-0.4, 1, -0.4) => known covariance, |rho| < 1
  target t; t.D = D; t.P.assign(D * D, 0.0); t.like0 = 0; t.calls = 0;
  for (int i = 0; i < D; i++) { t.P[i * D + i] = 1.0; if (i + 1 < D) t.P[i * D + i + 1] = t.P[(i + 1) * D + i] = -0.4; }

  stateSpace space(D);
  std::vector<std::string> names, types(D, "uni");
  std::vector<double> centers(D, 0.0), scales(D, 50.0);
  for (int i = 0; i < D; i++) names.push_back("x" + std::to_string(i));
  space.set_names(names);

  bayes_likelihood like_cb;
  gaussian_likelihood like_dev(t.P, t.like0);
  bayes_likelihood& like = callback ? like_cb : (bayes_likelihood&)like_dev;
  like.register_reference_object(&t);
  like.register_evaluate_log(my_loglike);
  like.basic_setup(&space, types, centers, scales);

  std::vector<double> sig(D, 2.38 / std::sqrt((double)D));
  gaussian_prop prop(sig, 0.0);

  parallel_tempering_chains ptc(Nt, 100.0, 0.2, 10);
  ptc.keep_history(1 + (nsteps + nsteps / 4) / 5);   // every 10th state, up to two adds per step
  ptc.track_exchanges(true);                          // instances / directions / ups / downs (chain.cc:1448-1451,1495-1498)
  ptc.initialize(&like, like.getObjectPrior().get(), 1);
  ptc.set_proposal(prop);

  // burn in, then accumulate second moments of the cold chain
  for (int k = 0; k < nsteps / 4; k++) ptc.step();
  std::vector<double> m2(D, 0.0);
  for (int k = 0; k < nsteps; k++) {
    ptc.step();
    state s = ptc.getState();
    for (int i = 0; i < D; i++) m2[i] += s.get_param(i) * s.get_param(i);
  }
  std::vector<int64_t> tries, acc;
  ptc.swap_counts(tries, acc);
  long st = 0, sa = 0;
  for (size_t i = 0; i < tries.size(); i++) { st += tries[i]; sa += acc[i]; }
  printf("mode=%s D=%d Ntemps=%d steps=%d  var(x0)=%.4f var(x%d)=%.4f  lpost0=%.6f  beta_top=%.6g  swaps %ld/%ld  likelihood_calls=%ld\n",
         callback ? "callback" : "device", D, Nt, ptc.getStep(), m2[0] / nsteps, D - 1, m2[D - 1] / nsteps, ptc.getLogPost(),
         ptc.subchain(Nt - 1)->invTemp(), sa, st, t.calls);
  printf("%s", ptc.status().c_str());
  printf("MAP lpost=%.6f at x0=%.4f (lpost now %.6f)\n", ptc.getMAPlpost(), ptc.getMAPstate().get_param(0), ptc.getLogPost());
  {   // every instance is still on exactly one rung, and hot states did travel down
    std::vector<int> seen(Nt, 0);
    int moved = 0;
    for (int i = 0; i < Nt; i++) { seen[ptc.getInstances()[i]]++; moved += ptc.getInstances()[i] != i; }
    bool perm = true;
    for (int i = 0; i < Nt; i++) perm = perm && seen[i] == 1;
    printf("instances %s a permutation of the rungs, %d displaced\n", perm ? "are" : "ARE NOT", moved);
    if (argc > 5) { std::ofstream ts(std::string(argv[5]) + ".tempstats"); ptc.dumpTempStats(ts); }
  }
  if (argc > 5) {   // the cold chain's file, as MH_chain::dumpChain writes it (chain.cc:1112-1135)
    std::ofstream os(argv[5]);
    os.precision(13);
    ptc.dumpChain(0, os, nsteps / 4, 10);
    printf("wrote %s\n", argv[5]);
  }
  return 0;
}
