// Host-side proposals through the facade (ptmcmc_gpu.hh): the proposal classes of proposal_distribution.hh that the device
// cannot draw itself run through the engine's host-proposal step.  Target: the correlated Gaussian P = tridiag(-0.4, 1, -0.4)
// (known covariance), evaluated on the device.  Checked by tests/test_cxx_facade.py.
//   usage: host_proposals <mode> [D] [Ntemps] [nsteps]
//     de          proposal_distribution_set{ differential_evolution (snooker 0.3) 0.7, gaussian_prop 0.3 }, Ninit = 20 D
//     usergauss   user_gaussian_prop on the sub-space {x1, x2} + gaussian_prop on all, with a check_update callback that
//                 hands a new covariance now and then and accept / reject callbacks that count
//     ugstatic    user_gaussian_prop without a callback: a fixed Gaussian, which goes to the device
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

struct parent { long updates, accepts, rejects, calls; };
static bool my_check_update(const void* parent_object, void* instance_object, const state& s, double invtemp, const std::vector<double>& randoms,
                            std::vector<double>& covarvec) {
  parent* p = (parent*)parent_object;
  (void)instance_object;
  p->calls++;
  if (randoms.size() != 2 || s.size() != 2 || !(invtemp > 0 && invtemp <= 1)) { printf("check_update: bad arguments\n"); exit(3); }
  if (randoms[0] > 0.05) return false;
  const double sc = (0.5 + randoms[1]) / invtemp;   // upper triangle, row by row: {c00, c01, c11}
  covarvec.assign({1.2 * sc, 0.5 * sc, 1.4 * sc});
  p->updates++;
  return true;
}
static void my_accept(const void* parent_object, void*) { ((parent*)parent_object)->accepts++; }
static void my_reject(const void* parent_object, void*) { ((parent*)parent_object)->rejects++; }

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "de";
  const int D = argc > 2 ? atoi(argv[2]) : 4, Nt = argc > 3 ? atoi(argv[3]) : 6, nsteps = argc > 4 ? atoi(argv[4]) : 4000;
  std::vector<double> P(D * D, 0.0);
  for (int i = 0; i < D; i++) { P[i * D + i] = 1.0; if (i + 1 < D) P[i * D + i + 1] = P[(i + 1) * D + i] = -0.4; }
  stateSpace space(D);
  std::vector<std::string> names, types(D, "uni");
  std::vector<double> centers(D, 0.0), scales(D, 30.0);
  for (int i = 0; i < D; i++) names.push_back("x" + std::to_string(i));
  space.set_names(names);
  gaussian_likelihood like(P, 0.0);
  like.basic_setup(&space, types, centers, scales);

  parent par = {0, 0, 0, 0};
  std::vector<double> sig(D, 2.38 / std::sqrt((double)D));
  proposal_distribution* prop = nullptr;
  int Ninit = 1;
  if (mode == "de") {
    differential_evolution* de = new differential_evolution(0.3, 0.3, 1e-4, 0.0, 0.0);
    de->reduce_gamma(1.0);
    prop = new proposal_distribution_set(std::vector<proposal_distribution*>{de, new gaussian_prop(sig, 0.2)}, std::vector<double>{0.7, 0.3});
    Ninit = 20 * D;
  } else if (mode == "usergauss" || mode == "ugstatic") {
    stateSpace sub(2);
    sub.set_names(std::vector<std::string>{"x1", "x2"});
    user_gaussian_prop* ug = new user_gaussian_prop(sub, std::vector<double>{1.0, 0.3, 1.5}, 2, "sub12", &par);
    if (mode == "usergauss") {
      ug->register_check_update(my_check_update);
      ug->register_accept_reject(my_accept, my_reject);
      prop = new proposal_distribution_set(std::vector<proposal_distribution*>{ug, new gaussian_prop(sig, 0.0)}, std::vector<double>{0.5, 0.5});
    } else {
      prop = ug;   // moves x1, x2 only
    }
  } else { printf("unknown mode\n"); return 2; }

  parallel_tempering_chains ptc(Nt, 50.0, 0.2, 2);
  ptc.initialize(&like, like.getObjectPrior().get(), Ninit);
  ptc.set_proposal(*prop);
  for (int k = 0; k < nsteps / 4; k++) ptc.step();
  std::vector<double> m2(D, 0.0), m11(D, 0.0);
  for (int k = 0; k < nsteps; k++) {
    ptc.step();
    state s = ptc.getState();
    for (int i = 0; i < D; i++) { m2[i] += s.get_param(i) * s.get_param(i); if (i + 1 < D) m11[i] += s.get_param(i) * s.get_param(i + 1); }
  }
  printf("mode=%s host=%d D=%d steps=%d size=%d Ninit_rows=%d", mode.c_str(), ptc.proposals_on_host() ? 1 : 0, D, ptc.getStep(), ptc.subchain(0)->size(), Ninit);
  for (int i = 0; i < D; i++) printf(" var%d=%.4f", i, m2[i] / nsteps);
  printf(" cov01=%.4f", m11[0] / nsteps);
  printf(" updates=%ld accepts=%ld rejects=%ld calls=%ld\n", par.updates, par.accepts, par.rejects, par.calls);
  printf("%s", ptc.report_prop(0).c_str());
  if (ptc.proposals_on_host()) {   // the chain file comes from the host mirror
    std::ostringstream os;
    ptc.dumpChain(0, os, nsteps, 8);
    int rows = 0;
    std::istringstream is(os.str());
    std::string line;
    while (std::getline(is, line)) if (!line.empty() && line[0] != '#') rows++;
    printf("dump_rows=%d\n", rows);
  }
  delete prop;
  return 0;
}
