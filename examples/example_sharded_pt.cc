// example_sharded_pt.cc -- a ladder sharded over the GPUs of one node by a C++ host through the C ABI alone:
// one process per GPU, contiguous rung blocks, the engine library's native RCCL step (ptm_shard_*: neighbour-only
// ncclSend / ncclRecv, no collective).  What replaces `mpirun -np N` + the reference's MPI layer (chain.cc:1199-1209,
// 1298-1309, 1879-1972) for this path; the rendezvous here is a file (any channel that carries 128 bytes will do).
//   build: g++ -std=c++11 -O2 -Iinclude examples/example_sharded_pt.cc -Lptmcmc_amd -lptm_engine -Wl,-rpath,$PWD/ptmcmc_amd -o sharded
//   usage: sharded <world> [nsteps=200] [walkers=256] [rungs=64]      world = 0: one engine, plain ptm_step (the comparison)
// Every rank prints a checksum of its block's states; the sum over ranks does not depend on the number of ranks.
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptm_engine.h"

#define CHK(x)                                                                      \
  do {                                                                              \
    if ((x) != PTM_OK) { printf("[rank %d] %s: %s\n", g_rank, #x, ptm_last_error()); fflush(stdout); _exit(3); } \
  } while (0)
static int g_rank = 0;

static void block_of(int Nt, int world, int rank, int& begin, int& count) {   // sizes differ by at most one
  const int base = Nt / world, rem = Nt % world;
  begin = rank * base + (rank < rem ? rank : rem);
  count = base + (rank < rem ? 1 : 0);
}

static int run_rank(int rank, int world, bool plain, int nsteps, int W, int Nt, const std::string& idfile) {
  g_rank = rank;
  const int D = 8;
  int begin = 0, count = Nt;
  if (!plain) block_of(Nt, world, rank, begin, count);
  const int ndev = ptm_device_count();
  if (ndev < 1) { printf("no gfx950 (MI355X) device visible\n"); return 4; }
  ptm_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.struct_size = sizeof cfg;
  cfg.dim = D; cfg.n_rungs = Nt; cfg.rung_begin = begin; cfg.rung_count = count; cfg.n_walkers = W; cfg.seed = 0x5EED0001ull;
  cfg.swap_rate = 0.2; cfg.add_every_n = 10; cfg.min_prior = -30; cfg.device = rank % ndev;
  ptm_engine* e = nullptr;
  CHK(ptm_engine_create(&cfg, &e));
  // problem: tridiag(-0.4, 1, -0.4) precision, uniform box prior, geometric ladder to Tmax = 100, diagonal steps
  std::vector<int32_t> open_(D, PTM_BOUND_OPEN), uni(D, PTM_PRIOR_UNIFORM);
  std::vector<double> zero(D, 0.0), half(D, 40.0), P(D * D, 0.0), beta(Nt), sig((size_t)count * D);
  for (int i = 0; i < D; i++) { P[i * D + i] = 1.0; if (i + 1 < D) P[i * D + i + 1] = P[(i + 1) * D + i] = -0.4; }
  for (int i = 0; i < Nt; i++) beta[i] = Nt > 1 ? std::pow(100.0, -(double)i / (Nt - 1)) : 1.0;
  for (int r = 0; r < count; r++) for (int d = 0; d < D; d++) sig[(size_t)r * D + d] = 2.38 / std::sqrt((double)D) / std::sqrt(beta[begin + r]);
  CHK(ptm_set_bounds(e, open_.data(), open_.data(), zero.data(), zero.data()));
  CHK(ptm_set_prior(e, uni.data(), zero.data(), half.data()));
  CHK(ptm_set_target_gaussian(e, nullptr, P.data(), 0.0));
  CHK(ptm_set_ladder(e, beta.data()));
  CHK(ptm_set_proposals(e, PTM_PROP_DIAG, sig.data(), nullptr));
  CHK(ptm_init_from_prior(e));   // keyed by global rung and walker: every rank draws its own block of the same population
  if (plain) {
    CHK(ptm_step(e, nsteps));
  } else {
    unsigned char id[PTM_SHARD_ID_BYTES];
    if (rank == 0) {   // rendezvous: rank 0 publishes the communicator id
      CHK(ptm_shard_unique_id(id));
      FILE* f = fopen((idfile + ".tmp").c_str(), "wb");
      fwrite(id, 1, sizeof id, f);
      fclose(f);
      rename((idfile + ".tmp").c_str(), idfile.c_str());
    } else {
      for (int tries = 0;; tries++) {
        FILE* f = fopen(idfile.c_str(), "rb");
        if (f) { const size_t n = fread(id, 1, sizeof id, f); fclose(f); if (n == sizeof id) break; }
        if (tries > 3000) { printf("[rank %d] no communicator id after 30 s\n", rank); return 5; }
        usleep(10000);
      }
    }
    std::vector<int32_t> counts(world);
    for (int r = 0; r < world; r++) { int b, c; block_of(Nt, world, r, b, c); counts[r] = c; }
    CHK(ptm_shard_init(e, id, rank, world, counts.data(), 0));
    CHK(ptm_shard_step(e, nsteps));
  }
  CHK(ptm_sync(e));
  std::vector<double> X((size_t)count * W * D);
  CHK(ptm_get_states(e, X.data()));
  double sum = 0, sq = 0;
  for (double v : X) { sum += v; sq += v * v; }
  printf("rank %d/%d rungs %d..%d x %d walkers: %d steps, checksum %.17g %.17g\n", rank, world, begin, begin + count, W, nsteps, sum, sq);
  fflush(stdout);
  CHK(ptm_shard_finalize(e));
  CHK(ptm_engine_destroy(e));
  return 0;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? atoi(argv[1]) : 1, nsteps = argc > 2 ? atoi(argv[2]) : 200, W = argc > 3 ? atoi(argv[3]) : 256, Nt = argc > 4 ? atoi(argv[4]) : 64;
  if (world <= 0) return run_rank(0, 1, true, nsteps, W, Nt, "");
  char tmpl[] = "/tmp/ptm_shard_id_XXXXXX";
  const int fd = mkstemp(tmpl);
  if (fd >= 0) { close(fd); unlink(tmpl); }
  const std::string idfile = tmpl;
  std::vector<pid_t> kids;
  for (int r = 0; r < world; r++) {   // one process per GPU (nothing has touched the GPU yet: fork is safe)
    const pid_t p = fork();
    if (p == 0) _exit(run_rank(r, world, false, nsteps, W, Nt, idfile));
    kids.push_back(p);
  }
  int bad = 0;
  for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad++; }
  unlink(idfile.c_str());
  if (bad) printf("%d rank(s) failed\n", bad);
  return bad ? 1 : 0;
}
