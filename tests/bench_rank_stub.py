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
    n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 12  # blocks of the whole job (ragged over the ranks if it does not divide)
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    load_package()
    from compression_accelerator_amd import sharding
    dist.init_process_group("gloo")
    data = fuzzgen.corpus("mtg")
    first, end = sharding.shard_range(n_total, rank, world)
    lens = torch.tensor([len(oracle_lib.raw_compress(data[k * 65536:(k + 1) * 65536])) for k in range(first, end)], dtype=torch.int32)
    if os.environ.get("SZ_STUB_DROP_RANK") == str(rank):  # fault injection: this rank's lengths do not arrive as sent
        sent = torch.zeros_like(lens)
    else:
        sent = lens
    gathered = torch.empty(n_total, dtype=torch.int32)
    offs = torch.empty(n_total + 1, dtype=torch.int64)
    sharding.gather_lengths(sent, n_total, out=gathered, offsets=offs)
    fields = bench.verify_gather(gathered, offs, lens, rank, world, n_total, True)
    ok = torch.tensor([1 if fields["gathered_layout_ok"] else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        fields["gathered_layout_ok"] = bool(ok.item())
        print(json.dumps({"n_gpus": world, "lens": gathered.tolist(), "total": int(offs[-1]), **fields}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
