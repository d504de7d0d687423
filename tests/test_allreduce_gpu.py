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


def _twoshot_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from bigdl_amd.parallel.xgmi_allreduce import XgmiAllReduce
    ar = XgmiAllReduce(small_bytes=16 * 1024, large_bytes=4 << 20)
    ok = True
    g = torch.Generator().manual_seed(99)
    # below 16 KiB: one-shot; above: two-shot, with chunk sizes that do and do not divide evenly; twice each (both sets)
    for n in [4096, 8 * 1001, 8 * 1001, 1 << 20, 1 << 20, 8 * 65, 2 << 20, 24]:
        parts = [torch.randn(n, generator=g).half() for _ in range(WORLD)]
        want = sum(p.float() for p in parts).half()
        t = parts[rank].cuda()
        ar(t)
        torch.cuda.synchronize()
        ar.check()
        ok &= bool(torch.equal(t.cpu(), want))
    assert ar.large.usable(torch.empty(1 << 20, dtype=torch.float16, device="cuda"))
    assert not ar.usable(torch.empty(4 << 20, dtype=torch.float16, device="cuda"))      # 8 MiB: RCCL / gloo path
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag)
    torch.save(flag, os.path.join(out_dir, f"ok{rank}.pt"))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_two_shot_all_reduce_two_processes_one_gpu(tmp_path):
    port = _free_port()
    mp.spawn(_twoshot_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for r in range(WORLD):
        assert int(torch.load(tmp_path / f"ok{r}.pt")) == WORLD


def _tp_engine_worker(rank, port, out_dir, mla=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.parallel.tensor_parallel import shard_engine
    cfg = DecoderConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2,
                        head_dim=64, max_position=512)
    if mla:     # latent attention (MiniCPM3 layout): 64 rotary + 32 plain channels per head, q and kv latents
        cfg = DecoderConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=8, num_kv_heads=8,
                            head_dim=96, max_position=512, rot_dim=32, rope_interleaved=True, mla_q_lora=128, mla_kv_lora=64,
                            attn_scale=96 ** -0.5)
    ids = torch.randint(3, 320, (1, 20), generator=torch.Generator().manual_seed(0)).cuda()
    full = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3)        # same seed: same weights on both ranks
    want = full.generate(ids, max_new_tokens=10)
    full.ensure_cache(1, 64)
    full.cache.reset()
    want_logits = full.prefill(ids, last_only=False).float().cpu()
    tp = shard_engine(build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3), rank, WORLD, fast_allreduce=True)
    assert tp.tp_fast_allreduce is not None and tp.cfg.num_heads == (4 if mla else 2)
    got = tp.generate(ids, max_new_tokens=10)                                  # graph decode with the one-shot all-reduce
    tp.tp_fast_allreduce.check()
    tp.cache.reset()
    got_logits = tp.prefill(ids, last_only=False).float().cpu()
    rel = ((got_logits - want_logits).abs().mean() / want_logits.abs().mean()).item()
    ok = rel < 2e-2 and (got == want).float().mean().item() >= 0.8
    torch.save(torch.tensor([int(ok), int(rel * 1e6)]), os.path.join(out_dir, f"tp{rank}.pt"))
    dist.barrier()
    tp.tp_fast_allreduce.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mla", [False, True])
def test_tensor_parallel_engine_two_processes_one_gpu(tmp_path, mla):
    """TP = 2 native engine (column / row sharded T16 weights, hipGraph decode with the one-shot all-reduce inside)
    against the unsharded engine; ``mla``: latent-attention layers (heads of q_b / kv_b sharded, latents replicated)."""
    port = _free_port()
    mp.spawn(_tp_engine_worker, args=(port, str(tmp_path), mla), nprocs=WORLD, join=True)
    for r in range(WORLD):
        ok, rel = torch.load(tmp_path / f"tp{r}.pt").tolist()
        assert ok == 1, f"rank {r}: rel diff {rel / 1e6}"


def _ep_engine_worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.parallel.expert_parallel import shard_engine_experts
    cfg = DecoderConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2,
                        head_dim=64, max_position=512, model_type="mixtral", num_experts=4, top_k=2, moe_intermediate_size=256)
    ids = torch.randint(3, 320, (2, 24), generator=torch.Generator().manual_seed(0)).cuda()
    full = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=5)
    want = full.generate(ids, max_new_tokens=8)
    full.cache.reset()
    want_logits = full.prefill(ids, last_only=False).float().cpu()
    # experts split over the ranks, tokens replicated: every rank adds its experts' share, one all-reduce per MoE layer
    ep = shard_engine_experts(build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=5), rank, WORLD, mode="allreduce")
    assert ep.layers[0].moe.n_local == 2 and ep.layers[0].moe.expert_range == (2 * rank, 2 * rank + 2)
    got = ep.generate(ids, max_new_tokens=8)
    ep.cache.reset()
    got_logits = ep.prefill(ids, last_only=False).float().cpu()
    rel = ((got_logits - want_logits).abs().mean() / want_logits.abs().mean()).item()
    ok = rel < 2e-2 and (got == want).float().mean().item() >= 0.8
    torch.save(torch.tensor([int(ok), int(rel * 1e6)]), os.path.join(out_dir, f"ep{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_expert_parallel_engine_two_processes_one_gpu(tmp_path):
    """Expert-parallel engine (experts sharded, partial sums all-reduced inside the decode step) against the whole
    engine.  The all-to-all form of the exchange needs one GPU per rank (RCCL); its arithmetic is covered on the host
    in tests/test_parallel_cpu.py."""
    port = _free_port()
    mp.spawn(_ep_engine_worker, args=(port, str(tmp_path)), nprocs=WORLD, join=True)
    for r in range(WORLD):
        ok, rel = torch.load(tmp_path / f"ep{r}.pt").tolist()
        assert ok == 1, f"rank {r}: rel diff {rel / 1e6}"


@pytest.mark.parametrize("fast", ["1", "0"], ids=["ipc_allreduce_in_graph", "host_staged_eager"])
def test_tensor_parallel_serving(tmp_path, fast):
    """shard_engine + ContinuousBatcher on two ranks (one GPU), schedule broadcast as fixed-shape tensors: staggered
    arrivals, admission limits (2 requests / 64 prompt tokens per step), one abort.  Every request gets the tokens the
    unsharded engine generates for it alone; both ranks run the same steps."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    port = _free_port()
    out = str(tmp_path / "res")
    procs = []
    for r in range(WORLD):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=os.path.dirname(here), HSA_ENABLE_IPC_MODE_LEGACY="0", TP_FAST=fast)
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "tp_serving_worker.py"), out], env=env))
    codes = [p.wait(timeout=240) for p in procs]
    assert codes == [0] * WORLD
    res = [json.load(open(f"{out}.{r}")) for r in range(WORLD)]
    head = res[0]
    assert head["graph"] == (fast == "1")          # IPC all-reduce: captured decode graph; gloo sums: eager steps
    budgets = [10, 7, 12, 6, 9]
    agree = total = 0
    for i, want in enumerate(head["alone"]):
        got = head["served"][i]
        assert len(got) == budgets[i] and head["reasons"][i] == "length"
        assert got[0] == want[0]
        agree += sum(a == b for a, b in zip(got, want))
        total += len(want)
    assert agree / total >= 0.9, (agree, total)
    assert head["reasons"][5] == "abort" and 0 < len(head["served"][5]) < 30
    for key in ("steps", "payload_steps", "batcher_steps", "generated", "left"):
        assert res[0][key] == res[1][key], key
    assert head["left"] == 0
    # payloads travelled only on the steps that admitted or aborted something: 3 arrival steps (the second one needs two
    # steps for its 3 requests at 2 admissions per step) + the abort
    assert head["payload_steps"] == 5, head["payload_steps"]


def test_rccl_all_reduce_inside_the_captured_decode_graph():
    """The fallback when the IPC all-reduce is not available: the engine's tensor-parallel sums are RCCL all-reduces
    captured into the decode hipGraph.  One process, a one-rank RCCL communicator (two ranks cannot share a device under
    RCCL): the collective is really enqueued and captured, and generation equals the engine without a group."""
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        cfg = DecoderConfig(vocab_size=320, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2,
                            head_dim=64, max_position=512)
        ids = torch.randint(3, 320, (1, 20), generator=torch.Generator().manual_seed(0)).cuda()
        plain = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3)
        want = plain.generate(ids, max_new_tokens=12)
        eng = build_random_decoder(cfg, Q.SYM_INT4, device="cuda", seed=3)
        eng.tp_group, eng.tp_world = dist.group.WORLD, 1
        calls = []
        real = dist.all_reduce
        dist.all_reduce = lambda t, *a, **k: (calls.append(torch.cuda.is_current_stream_capturing()), real(t, *a, **k))[1]
        try:
            got = eng.generate(ids, max_new_tokens=12)
        finally:
            dist.all_reduce = real
        assert eng._reduce_capturable() and any(g is not None for g in eng._graphs.values())
        assert any(calls), "no all-reduce was issued while the decode graph was being captured"
        assert torch.equal(got, want)
    finally:
        dist.destroy_process_group()
