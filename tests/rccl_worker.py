"""One rank, backend "nccl" (= RCCL on ROCm): the exchange helpers of blp_amd.ranking -- _all_reduce of the replicated query
vectors, _all_gather_into of the (2T, 4) int32 counts, all_gather_rows of a table shard -- run through RCCL on this box's GPU
(tests/test_gpu_shard.py).  A world of one moves no data between devices; what it shows is that the process group comes up on
the device and that the helpers' tensor shapes / dtypes are ones RCCL takes."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blp_amd import ranking  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29611")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
g = torch.Generator(device=dev).manual_seed(3)
vectors = torch.randn((2 * 128, 128), device=dev, generator=g)
want = vectors.clone()
ranking._all_reduce(vectors)
out["all_reduce_exact"] = bool(torch.equal(vectors, want))
counts = torch.randint(0, 1 << 20, (2 * 128, 4), device=dev, dtype=torch.int32, generator=g)
gathered = torch.empty((1, 2 * 128, 4), device=dev, dtype=torch.int32)
ranking._all_gather_into(gathered.view(-1), counts.view(-1))
out["all_gather_exact"] = bool(torch.equal(gathered[0], counts))
shard = torch.randn((1000, 128), device=dev, generator=g)
out["all_gather_rows_exact"] = bool(torch.equal(ranking.all_gather_rows(shard, 1000, 1), shard))
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print(json.dumps(out))
