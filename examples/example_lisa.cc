// example_lisa.cc -- BASELINE configs[4]: a USER plug-in likelihood through the reference's own interface
// (bayes_likelihood::register_evaluate_log, bayesian.hh:544-552), a mixed uniform / polar / co-polar prior with limit and
// wrap boundaries (the set-up of the reference's exampleLISA.cc:528-593), the ptmcmc_sampler driver loop
// (ptmcmc.cc:489-679) with its defaults (evolving ladder, chain files) -- all against ptmcmc_gpu.hh.
// The likelihood is the toy LISA extrinsic-parameter model of that example (antenna responses, exampleLISA.cc:59-72;
// log-likelihood :130-142), written here from its formulas; parameters: d, phi, inc, lambda, beta, psi.
//   build: g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd
//   usage: example_lisa <outbase> [--nsteps=N] [--pt=Ntemps] [--nchains=R] [--option=value ...]
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

typedef std::complex<double> cplx;
static const double FACTOR = 216147.866077;
static const cplx SA_INJ(0.33687296665053773, 0.087978055005482114), SE_INJ(-0.12737105239204741, 0.21820079314765678);

static cplx modes(double d, double phi, double inc, double psi, cplx plus, cplx cross) {
  const cplx I(0, 1);
  const double pref = 0.5 / d * std::sqrt(5 / M_PI);
  const cplx m22 = pref * std::pow(std::cos(inc / 2), 4) * std::exp(2.0 * I * (-phi - psi)) * 0.5 * (plus + I * cross);
  const cplx m2m2 = pref * std::pow(std::sin(inc / 2), 4) * std::exp(2.0 * I * (-phi + psi)) * 0.5 * (plus - I * cross);
  return m22 + m2m2;
}
static double lisa_loglike(void*, const state& s) {
  const std::vector<double> x = s.get_params_vector();
  const double d = x[0], phi = x[1], inc = x[2], lam = x[3], beta = x[4], psi = x[5];
  const cplx I(0, 1);
  const cplx a_plus = I * (0.75 * (3 - std::cos(2 * beta)) * std::cos(2 * lam - M_PI / 3));
  const cplx a_cross = I * (3.0 * std::sin(beta) * std::sin(2 * lam - M_PI / 3));
  const cplx e_plus = -I * (0.75 * (3 - std::cos(2 * beta)) * std::sin(2 * lam - M_PI / 3));
  const cplx e_cross = I * (3.0 * std::sin(beta) * std::cos(2 * lam - M_PI / 3));
  const cplx sa = modes(d, phi, inc, psi, a_plus, a_cross), se = modes(d, phi, inc, psi, e_plus, e_cross);
  return -0.5 * FACTOR * (std::norm(sa - SA_INJ) + std::norm(se - SE_INJ));
}

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s <outbase> [--option=value ...]\n", argv[0]); return 2; }
  const int D = 6;
  const double PI = M_PI;
  stateSpace space(D);
  space.set_names(std::vector<std::string>{"d", "phi", "inc", "lambda", "beta", "psi"});
  space.set_bound(0, boundary(boundary::limit, boundary::limit, 0, 30));
  space.set_bound(1, boundary(boundary::wrap, boundary::wrap, 0, 2 * PI));
  space.set_bound(2, boundary(boundary::limit, boundary::limit, 0, PI));
  space.set_bound(3, boundary(boundary::wrap, boundary::wrap, 0, 2 * PI));
  space.set_bound(4, boundary(boundary::limit, boundary::limit, -PI / 2, PI / 2));
  space.set_bound(5, boundary(boundary::wrap, boundary::wrap, 0, PI));
  bayes_likelihood like;
  like.register_evaluate_log(lisa_loglike);
  std::vector<std::string> types = {"uni", "uni", "pol", "uni", "cpol", "uni"};
  std::vector<double> centers = {1.667, PI, PI / 2, PI, 0, PI / 2}, scales = {1.333, PI, PI / 2, PI, PI / 2, PI / 2};
  like.basic_setup(&space, types, centers, scales);
  // the sampler's default Gaussian recipe (ptmcmc.cc:117-139), without its differential-evolution part
  std::vector<proposal_distribution*> gset;
  std::vector<double> gshares;
  double fac = 1.0, share = 1;
  for (int i = 0; i < 6; i++) {
    std::vector<double> sig(D);
    for (int d = 0; d < D; d++) sig[d] = scales[d] / fac;
    gset.push_back(new gaussian_prop(sig, 0.2));
    fac *= 4.0;
    share *= 2;
    gshares.push_back(share);
  }
  proposal_distribution_set prop(gset, gshares);
  for (auto g : gset) delete g;
  ptmcmc_sampler mcmc;
  mcmc.set("nsteps", "4000"); mcmc.set("pt", "20"); mcmc.set("pt_Tmax", "1e9"); mcmc.set("save_every", "4");
  mcmc.set("nevery", "1000"); mcmc.set("nskip", "4"); mcmc.set("pt_dump_n", "1");
  if (!mcmc.parse(argc - 1, argv + 1)) { printf("bad option\n"); return 2; }
  mcmc.setup(like);
  mcmc.select_proposal(prop);
  mcmc.initialize();   // prior draws on the device (uniform / polar / co-polar), the plug-in prices them on the host
  mcmc.run(argv[1]);
  printf("%s", mcmc.chains()->status().c_str());
  printf("MAP: lpost = %.6f at %s\n", mcmc.chains()->getMAPlpost(), mcmc.chains()->getMAPstate().get_string().c_str());
  return 0;
}
