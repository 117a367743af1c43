"""The reference-shard protocol over RCCL on device buffers (torch.distributed backend "nccl"), as bench.py --gpus N runs it -- here
with the one GPU of the test box, i.e. a process group of one rank: the library loads, the group forms on the device, and
all_to_all_single / all_reduce move the counters the scan wrote (tests/test_refshard_gloo.py covers 2-4 ranks over gloo)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
root, mode, port = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
import bench
import oracle_lib as O
from uvaia_amd import capi, hostlib, refshard
gen = hostlib.Synth(4097, seed=77, preset=0)
qseqs, _ = gen.generate_bytes(bench.QUERY_INDEX0, 70)
qnames = ["q%d" % i for i in range(70)]
pq = hostlib.PreparedQuery(qseqs, qnames, acgt=(mode == "acgt"))
ok = bench.parity_on_refshard(O, capi, refshard, dist, pq, gen, qseqs, qnames, mode, 12, 1, 0, 0, True)
dist.barrier()
dist.destroy_process_group()
print("PARITY", ok)
sys.exit(0 if ok else 1)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["iupac", "acgt"])
def test_reference_shards_over_rccl_with_one_rank(mode, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29600 + (os.getpid() % 300) + (0 if mode == "iupac" else 1)
    r = subprocess.run([sys.executable, str(script), ROOT, mode, str(port)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PARITY True" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
