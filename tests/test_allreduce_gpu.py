"""One-shot IPC all-reduce kernel: two processes share GPU 0 (IPC mapping, flag protocol, epoch double buffering and
hipGraph replay are the same as across GPUs; only the link differs).  Results must equal the fp32 sum of the inputs
and be identical on both ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from bigdl_amd.parallel.xgmi_allreduce import OneShotAllReduce
    ar = OneShotAllReduce(max_bytes=64 * 1024)
    ok = True
    g = torch.Generator().manual_seed(1234)          # same stream of inputs on both ranks
    for it, n in enumerate([8, 4096, 8192, 32768, 4096, 4096, 4096]):
        parts = [torch.randn(n, generator=g).half() for _ in range(WORLD)]
        want = sum(p.float() for p in parts).half()
        t = parts[rank].cuda()
        ar(t)
        torch.cuda.synchronize()
        ar.check()
        ok &= bool(torch.equal(t.cpu(), want))
    # graph replay: the epoch counter advances on the device
    parts = [torch.randn(4096, generator=g).half() for _ in range(WORLD)]
    static = parts[rank].cuda()
    src = static.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            ar(static)
    torch.cuda.current_stream().wait_stream(side)
    want = sum(p.float() for p in parts).half()
    for _ in range(5):
        static.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        ok &= bool(torch.equal(static.cpu(), want))
    ar.check()
    # unsupported shapes fall back to the process group
    odd = torch.ones(5, dtype=torch.float16, device="cuda")
    assert not ar.usable(odd)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag)
    torch.save(flag, os.path.join(out_dir, f"ok{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_one_shot_all_reduce_two_processes_one_gpu(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for r in range(WORLD):
        assert int(torch.load(tmp_path / f"ok{r}.pt")) == WORLD
