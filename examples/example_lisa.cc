// example_lisa.cc -- BASELINE configs[4] through the DRIVER, call for call as a program written against the reference:
// the sequence of its exampleLISA.cc main() (:698-822)
//
//     ptmcmc_sampler::Init(argc, argv);  Options opt;  ptmcmc_sampler mcmc;  bayes_sampler* s0 = &mcmc;
//     [likelihood]  s0->addOptions(opt);  like->addOptions(opt);  opt.add(Option("nchains" | "seed" | "precision" | "outname"));
//     opt.parse(argc, argv);  ProbabilityDist::setSeed(seed);  mcmc.setup(*like, precision);  mcmc.select_proposal();
//     for (ic < Nchain) { bayes_sampler* s = s0->clone();  s->initialize();  s->run(base, ic);  delete s; }
//
// against ptmcmc_gpu.hh ("one include and one using").  The likelihood is a USER plug-in registered through
// bayes_likelihood::register_evaluate_log (bayesian.hh:544-552): the toy LISA extrinsic-parameter model of that example
// (antenna responses exampleLISA.cc:59-72, log-likelihood :130-142), written here from its formulas; parameters d, phi, inc,
// lambda, beta, psi with the mixed uniform / polar / co-polar prior and limit / wrap boundaries of :528-593.
// select_proposal() builds the reference's default recipe (ptmcmc.cc:15-183): 80 % differential evolution + six diagonal
// Gaussians -- a host-side proposal (the engine's host-proposal step); --gauss_draw_frac=1 makes it all Gaussian, fused on
// the device.
//   build: g++ -std=c++11 -O2 -pthread -Iinclude -Iptmcmc_amd/host examples/example_lisa.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd
//   usage: example_lisa [--outname=base] [--nsteps=N] [--pt=Ntemps] [--nchains=R] [--replicas=W] [--seed=s] [--option=value ...]
#include <cmath>
#include <complex>
#include <cstdio>
#include <ctime>
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

// the likelihood's set-up (the reference's simple_likelihood_setup_nc, exampleLISA.cc:528-593, in this program's words)
static void setup_likelihood(bayes_likelihood* like) {
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
  like->register_evaluate_log(lisa_loglike);
  const std::vector<std::string> types = {"uni", "uni", "pol", "uni", "cpol", "uni"};
  const std::vector<double> centers = {1.667, PI, PI / 2, PI, 0, PI / 2}, scales = {1.333, PI, PI / 2, PI, PI / 2, PI / 2};
  like->basic_setup(&space, types, centers, scales);
}

int main(int argc, char* argv[]) {
  ptmcmc_sampler::Init(argc, argv);
  Options opt(true);
  // create the sampler
  ptmcmc_sampler mcmc;
  bayes_sampler* s0 = &mcmc;
  // create the likelihood
  bayes_likelihood* like = new bayes_likelihood();
  setup_likelihood(like);

  // prep command-line options
  s0->addOptions(opt);
  like->addOptions(opt);
  opt.add(Option("nchains", "How many chains to run, one after the other. [1]", "1"));
  opt.add(Option("seed", "Seed of the random streams, a number in [0,1). [-1: seed from the clock]", "-1"));
  opt.add(Option("precision", "Significant digits in the chain files. [13]", "13"));
  opt.add(Option("outname", "Stem of the output file names. [mcmc_output]", "mcmc_output"));
  const bool parseBAD = opt.parse(argc, argv);
  if (parseBAD) {
    std::cout << "Usage:\n example_lisa [--options=vals] " << std::endl;
    std::cout << opt.print_usage() << std::endl;
    return 1;
  }
  std::cout << "flags=\n" << opt.report() << std::endl;
  like->setup();

  double seed;
  int Nchain, output_precision;
  std::string outname;
  std::istringstream(opt.value("nchains")) >> Nchain;
  std::istringstream(opt.value("seed")) >> seed;
  if (seed < 0) seed = std::fmod(time(NULL) / 3.0e7, 1);   // seed from the clock
  std::istringstream(opt.value("precision")) >> output_precision;
  std::istringstream(opt.value("outname")) >> outname;
  if (argc > 1) outname = argv[1];   // (a bare first argument names the output too)
  std::cout.precision(output_precision);
  std::cout << "\noutname = '" << outname << "'" << std::endl;
  std::cout << "seed=" << seed << std::endl;
  ProbabilityDist::setSeed(seed);

  // the space / prior, for the report
  const stateSpace space = *like->getObjectStateSpace();
  std::cout << "like.nativeSpace=\n" << space.show() << std::endl;
  std::shared_ptr<const sampleable_probability_function> prior = like->getObjectPrior();
  std::cout << "Prior is:\n" << prior->show() << std::endl;
  std::cout << "Npar=" << space.size() << std::endl;

  // Bayesian sampling: set up the sampler and its proposal distribution
  mcmc.setup(*like, output_precision);
  mcmc.select_proposal();

  const std::string base = outname;
  for (int ic = 0; ic < Nchain; ic++) {
    bayes_sampler* s = s0->clone();
    s->initialize();
    s->run(base, ic);
    ptmcmc_sampler* ps = dynamic_cast<ptmcmc_sampler*>(s);
    std::cout << ps->chains()->status();
    std::cout << "MAP: lpost = " << ps->chains()->getMAPlpost() << " at " << ps->chains()->getMAPstate().get_string() << std::endl;
    std::cout << "proposals drawn on the " << (ps->chains()->proposals_on_host() ? "host" : "device")
              << (ps->chains()->draws_de_on_device() ? " (differential evolution from the device's own history)" : "") << std::endl;
    delete s;
  }
  // summary
  std::cout << "best_post " << like->bestPost() << ", state=" << like->bestState().get_string() << std::endl;
  delete like;
  return 0;
}
