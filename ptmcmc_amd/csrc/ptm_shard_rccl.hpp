// ptm_shard_rccl.hpp -- the sharded PT step driven NATIVELY: contiguous rung blocks, one engine per GPU / process, neighbour
// point-to-point messages over RCCL (ncclSend / ncclRecv inside one group per message round), no collective on the data path.
// The C++ twin of ptmcmc_amd/parallel.py::ShardedLadder (same phases, same overlap), for hosts that stay C++ (north star).
// Included by ptm_engine.hip only.
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first ptm_shard_* call): libptm_engine.so carries no link-time
// dependency on it, single-GPU users never load it, and a process that already holds an RCCL (torch's) gets that one.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <vector>

namespace ptm {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  const char* load() {   // null on success, else what failed
    if (lib) return nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return "librccl.so.1 not found (dlopen)";
#define PTM_SYM(field, name)                                  \
    *(void**)(&field) = dlsym(lib, name);                     \
    if (!field) { lib = nullptr; return "RCCL symbol missing: " name; }
    PTM_SYM(GetUniqueId, "ncclGetUniqueId")
    PTM_SYM(CommInitRank, "ncclCommInitRank")
    PTM_SYM(CommDestroy, "ncclCommDestroy")
    PTM_SYM(GroupStart, "ncclGroupStart")
    PTM_SYM(GroupEnd, "ncclGroupEnd")
    PTM_SYM(Send, "ncclSend")
    PTM_SYM(Recv, "ncclRecv")
    PTM_SYM(AllGather, "ncclAllGather")
    PTM_SYM(GetErrorString, "ncclGetErrorString")
#undef PTM_SYM
    return nullptr;
  }
};
inline RcclApi& rccl() { static RcclApi a; return a; }

// one shard's communication state (owned by the engine)
struct ShardComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, halo = 0, h_recv = 0, h_send = 0;
  int up = -1, down = -1;                  // neighbour ranks or -1
  hipStream_t cstream = nullptr;           // messages travel on their own stream, next to the engine's kernels
  hipEvent_t ev_ready = nullptr, ev_rows = nullptr, ev_halo = nullptr;
  double *ll_top = nullptr, *ll_bottom = nullptr, *ll_below = nullptr, *ll_above = nullptr;   // llike halos
  double *send_up = nullptr, *recv_above = nullptr, *send_down = nullptr, *recv_below = nullptr;   // boundary row messages
  size_t row_doubles = 0;
  bool halos_in_flight = false;
  bool recover = false;                    // runs longer than the halo are decided by a gathered second pass (ptm_set_shard_map)
  // evolving ladders (ptm_set_evolve_temps on a rung shard): the whole ladder's llikes / lpriors, all-gathered each step --
  // gsend [2][maxn * W] this shard's (padded to the largest shard), grecv [world][2][maxn * W], ll_all / lp_all [Nt][W]
  std::vector<int> counts;
  int maxn = 0;
  double *gsend = nullptr, *grecv = nullptr, *ll_all = nullptr, *lp_all = nullptr;
  hipEvent_t ev_gather = nullptr;
};

}  // namespace ptm
