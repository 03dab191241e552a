// Priors beyond the per-dimension product through the facade (ptmcmc_gpu.hh): independent_dist_product
// (probability_function.hh:181-215) and a user's own subclass of sampleable_probability_function (evaluate_log + drawSample,
// nothing else) -- the latter is evaluated on the host through the engine's prior callback.  Checked by tests/test_cxx_facade.py.
//   usage: host_prior <mode> [Ntemps] [nsteps]
//     indep       independent_dist_product{ uniform on (a, b), gaussian on (c, d) }: a per-dimension product after all -> device
//     user        a correlated 2-D Gaussian prior N(0, S) written as a subclass; likelihood N(0, P^-1): posterior cov (P + S^-1)^-1
//     indepuser   independent_dist_product{ that subclass on (a, b), uniform on (c, d) } -> host, through the product class
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

// N(0, S), S = [[s00, s01], [s01, s11]]: only what the reference asks of a prior
class corr_gauss_prior : public sampleable_probability_function {
  double s00, s01, s11, det;

 public:
  mutable long calls;
  corr_gauss_prior(const stateSpace* sp, double s00, double s01, double s11) : sampleable_probability_function(sp), s00(s00), s01(s01), s11(s11), calls(0) {
    dim = 2;
    det = s00 * s11 - s01 * s01;
  }
  double evaluate(state& s) const override {
    if (s.invalid()) return 0;
    calls++;
    const double x = s.get_param(0), y = s.get_param(1);
    const double q = (s11 * x * x - 2 * s01 * x * y + s00 * y * y) / det;
    return std::exp(-0.5 * q) / (2 * M_PI * std::sqrt(det));
  }
  state drawSample(Random& rng) const override {
    const double u1 = rng.Next(), u2 = rng.Next();
    const double r = std::sqrt(-2 * std::log(u1)), z0 = r * std::cos(2 * M_PI * u2), z1 = r * std::sin(2 * M_PI * u2);
    const double l00 = std::sqrt(s00), l10 = s01 / l00, l11 = std::sqrt(s11 - l10 * l10);
    return state(space, std::vector<double>{l00 * z0, l10 * z0 + l11 * z1});
  }
  void getScales(std::valarray<double>& out) const override { out = std::valarray<double>{std::sqrt(s00), std::sqrt(s11)}; }
  std::string show() const override { return "CorrelatedGaussianPrior()"; }
};

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "user";
  const int Nt = argc > 2 ? atoi(argv[2]) : 5, nsteps = argc > 3 ? atoi(argv[3]) : 4000;
  const int D = mode == "user" ? 2 : 4;
  stateSpace space(D), sA(2), sB(2);
  space.set_names(D == 2 ? std::vector<std::string>{"a", "b"} : std::vector<std::string>{"a", "b", "c", "d"});
  sA.set_names(std::vector<std::string>{"a", "b"});
  sB.set_names(std::vector<std::string>{"c", "d"});
  // likelihood: independent Gaussians of precision p_i
  std::vector<double> P(D * D, 0.0);
  const double prec[4] = {1.0, 0.5, 2.0, 1.5};
  for (int i = 0; i < D; i++) P[i * D + i] = prec[i];
  gaussian_likelihood like(P, 0.0);
  const double s00 = 1.5, s01 = 0.9, s11 = 2.0;
  corr_gauss_prior* user = new corr_gauss_prior(D == 2 ? &space : &sA, s00, s01, s11);
  sampleable_probability_function* prior = nullptr;
  if (mode == "user") prior = user;
  else if (mode == "indep")
    prior = new independent_dist_product(&space, new uniform_dist_product(&sA, std::valarray<double>{-20, -20}, std::valarray<double>{20, 20}),
                                         new gaussian_dist_product(&sB, std::valarray<double>{0.5, -0.5}, std::valarray<double>{1.0, 2.0}));
  else if (mode == "indepuser")
    prior = new independent_dist_product(&space, user, new uniform_dist_product(&sB, std::valarray<double>{-20, -20}, std::valarray<double>{20, 20}));
  else { printf("unknown mode\n"); return 2; }
  like.basic_setup(&space, prior);

  std::vector<double> sig(D, 1.0);
  gaussian_prop prop(sig, 0.0);
  parallel_tempering_chains ptc(Nt, 30.0, 0.2, 2);
  ptc.initialize(&like, like.getObjectPrior().get(), 1);
  ptc.set_proposal(prop);
  for (int k = 0; k < nsteps / 4; k++) ptc.step();
  std::vector<double> m1(D, 0.0), m2(D, 0.0);
  double m01 = 0;
  for (int k = 0; k < nsteps; k++) {
    ptc.step();
    state s = ptc.getState();
    for (int i = 0; i < D; i++) { m1[i] += s.get_param(i); m2[i] += s.get_param(i) * s.get_param(i); }
    m01 += s.get_param(0) * s.get_param(1);
  }
  printf("mode=%s hostprior=%d D=%d steps=%d calls=%ld", mode.c_str(), ptc.prior_evaluated_on_host() ? 1 : 0, D, ptc.getStep(), user->calls);
  for (int i = 0; i < D; i++) printf(" mean%d=%.4f var%d=%.4f", i, m1[i] / nsteps, i, m2[i] / nsteps - (m1[i] / nsteps) * (m1[i] / nsteps));
  printf(" cov01=%.4f\n", m01 / nsteps - (m1[0] / nsteps) * (m1[1] / nsteps));
  return 0;
}
