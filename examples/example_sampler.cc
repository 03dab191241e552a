// example_sampler.cc -- the reference's driver loop (ptmcmc_sampler: setup, select_proposal, initialize, run;
// ptmcmc.cc:489-679) written against ptmcmc_gpu.hh: chain files <base>_t<k>.dat come out of the device's history ring.
//   build: g++ -std=c++11 -O2 -Iinclude -Iptmcmc_amd/host examples/example_sampler.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -pthread
//   usage: example_sampler <outbase> [--nsteps=N] [--pt=Ntemps] [--save_every=S] [--nevery=E] [--nskip=K] [--pt_dump_n=M]
#include <cmath>
#include <cstdio>
#include <vector>

#include "ptmcmc_gpu.hh"
using namespace ptmgpu;

int main(int argc, char** argv) {
  if (argc < 2) { printf("usage: %s <outbase> [--option=value ...]\n", argv[0]); return 2; }
  const int D = 3;
  std::vector<double> P = {2.0, 0.6, 0.0, 0.6, 1.0, -0.3, 0.0, -0.3, 1.5};   // precision of the target
  stateSpace space(D);
  space.set_names(std::vector<std::string>{"a", "b", "c"});
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
  mcmc.select_proposal(prop);
  mcmc.initialize();
  mcmc.run(argv[1]);
  printf("%s", mcmc.chains()->status().c_str());
  return 0;
}
