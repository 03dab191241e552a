"""One rank of the CPU (gloo) test of ptmcmc_amd.parallel.preflight_neighbour_messages: prints the all-reduced verdict."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptmcmc_amd.parallel import preflight_neighbour_messages

if __name__ == "__main__":
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    sabotage_rank = int(sys.argv[1])
    dist.init_process_group("gloo")
    bad = preflight_neighbour_messages(dist, torch, rank, world, torch.device("cpu"), None, sabotage=(sabotage_rank == -2 or rank == sabotage_rank), wait_s=4.0)
    flag = torch.tensor([bad], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    print("rank %d bad %d agreed %d" % (rank, bad, int(flag.item())), flush=True)
    dist.barrier()
    dist.destroy_process_group()
