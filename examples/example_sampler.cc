// example_sampler.cc -- the reference's driver loop (ptmcmc_sampler: setup, select_proposal, initialize, run;
// ptmcmc.cc:489-679) written against ptmcmc_gpu.hh: chain files <base>_t<k>.dat come out of the device's history ring.
//   build: g++ -std=c++11 -O2 -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -pthread
//   usage: example_sampler <outbase> [--nsteps=N] [--pt=Ntemps] [--save_every=S] [--nevery=E] [--nskip=K] [--pt_dump_n=M]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s <outbase> [--option=value ...]\n", argv[0]); return 2; }
  // --dim=D (first option, if given): a D-dimensional tridiagonal-precision target instead of the 3-dimensional one -- e.g. --dim=12
  // with --pt=64: a ladder long enough for the persistent ladder kernel (its build with everything the sampler switches on)
  // --default_recipe (next, if given): select_proposal() without an argument -- the reference's default set, 80 % differential evolution
  // from the chains' saved history + six Gaussians (ptmcmc.cc:15-183), drawn on the device -- instead of the Gaussians below
  int D = 3;
  bool default_recipe = false;
  if (argc > 2 && std::string(argv[2]).rfind("--dim=", 0) == 0) {
    D = atoi(argv[2] + 6);
    for (int k = 2; k + 1 < argc; k++) argv[k] = argv[k + 1];
    argc--;
  }
  if (argc > 2 && std::string(argv[2]) == "--default_recipe") {
    default_recipe = true;
    for (int k = 2; k + 1 < argc; k++) argv[k] = argv[k + 1];
    argc--;
  }
  std::vector<double> P = {2.0, 0.6, 0.0, 0.6, 1.0, -0.3, 0.0, -0.3, 1.5};   // precision of the target
  if (D != 3) {
    P.assign((size_t)D * D, 0.0);
    for (int i = 0; i < D; i++) { P[(size_t)i * D + i] = 1.0 + 0.1 * i; if (i + 1 < D) P[(size_t)i * D + i + 1] = P[(size_t)(i + 1) * D + i] = 0.3; }
  }
  stateSpace space(D);
  {
    std::vector<std::string> names;
    for (int i = 0; i < D; i++) names.push_back(D == 3 ? std::string(1, (char)('a' + i)) : "p" + std::to_string(i));
    space.set_names(names);
  }
  gaussian_likelihood like(P, 0.0);
  std::vector<std::string> types(D, "uni");
  std::vector<double> centers(D, 0.0), scales(D, 20.0);
  like.basic_setup(&space, types, centers, scales);
  // the sampler's default Gaussian recipe (ptmcmc.cc:117-139): six diagonal Gaussians a factor 4 apart in scale with
  // doubling shares, each with 20 % one-dimensional moves; here without the differential-evolution part
  std::vector<proposal_distribution*> gset;
  std::vector<double> gshares;
  double fac = std::pow(2.0 / 4.0, 4.0), share = 1;
  for (int i = 0; i < 6; i++) {
    fac *= 4.0;
    std::vector<double> sig(D);
    for (int d = 0; d < D; d++) sig[d] = scales[d] / 100.0 / fac * 400.0;
    gset.push_back(new gaussian_prop(sig, 0.2));
    share *= 2;
    gshares.push_back(share);
  }
  proposal_distribution_set prop(gset, gshares);
  ptmcmc_sampler mcmc;
  mcmc.set("nsteps", "2000"); mcmc.set("pt", "6"); mcmc.set("pt_Tmax", "50"); mcmc.set("save_every", "2");
  mcmc.set("nevery", "500"); mcmc.set("nskip", "4"); mcmc.set("pt_dump_n", "2"); mcmc.set("pt_swap_rate", "0.3");
  if (!mcmc.parse(argc - 1, argv + 1)) { printf("bad option\n"); return 2; }
  mcmc.setup(like);
  if (default_recipe) mcmc.select_proposal();
  else mcmc.select_proposal(prop);
  mcmc.initialize();
  mcmc.run(argv[1]);
  printf("%s", mcmc.chains()->status().c_str());
  printf("step kernel: %s%s\n", ptm_step_kernel_name(mcmc.chains()->engine()), mcmc.chains()->draws_de_on_device() ? "  (differential evolution drawn on the device)" : "");
  return 0;
}
