"""One rank of a CPU rehearsal of bench.py's N>1 path, started by bench.launch_ranks (the launcher
under test) through torch.distributed.run: gloo instead of RCCL, per-block lengths from the oracle
instead of the kernel, then the same exchange code (sharding.gather_lengths) and the same report
fields (bench.layout_fields) as a real rank.  TEST INFRASTRUCTURE; prints one JSON line on rank 0."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch
import torch.distributed as dist

import bench
import fuzzgen
import oracle_lib
from pkg_loader import load_package


def main():
    blocks_per_rank = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    load_package()
    from compression_accelerator_amd import sharding
    dist.init_process_group("gloo")
    data = fuzzgen.corpus("mtg")
    first = rank * blocks_per_rank
    lens = torch.tensor([len(oracle_lib.raw_compress(data[(first + k) * 65536:(first + k + 1) * 65536]))
                         for k in range(blocks_per_rank)], dtype=torch.int32)
    n_total = blocks_per_rank * world
    gathered = torch.empty(n_total, dtype=torch.int32)
    offs = torch.empty(n_total + 1, dtype=torch.int64)
    sharding.gather_lengths(lens, n_total, out=gathered, offsets=offs)
    fields = bench.layout_fields(gathered, offs, lens, rank, blocks_per_rank)
    ok = torch.tensor([1 if fields["gathered_layout_ok"] else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        fields["gathered_layout_ok"] = bool(ok.item())
        print(json.dumps({"n_gpus": world, "lens": gathered.tolist(), "total": int(offs[-1]), **fields}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
