"""Multi-process (gloo, world_size 2) checks of tensor parallelism, pipeline parallelism and the collective helpers:
every distributed result must equal the single-process result."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from transformers import LlamaConfig, LlamaForCausalLM

WORLD = int(os.environ.get("BIGDL_TEST_WORLD", "2"))      # rehearse more ranks with BIGDL_TEST_WORLD=4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny(layers=4, hidden=128, inter=256):
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                      num_attention_heads=2 * max(2, WORLD), num_key_value_heads=max(2, WORLD),
                      max_position_embeddings=256,
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    m.generation_config.eos_token_id = None
    m.generation_config.pad_token_id = 0
    return m


def _boot(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD),
                      LOCAL_RANK=str(rank), BIGDL_FORCE_CPU="1")
    torch.set_num_threads(2)
    from bigdl_amd.parallel import init_distributed
    return init_distributed("gloo")


def _run(fn, *args):
    port = _free_port()
    mp.spawn(fn, args=(port, *args), nprocs=WORLD, join=True)


# ------------------------------------------------------------------------------------------------
def _tp_worker(rank, port, low_bit, tmp):
    _boot(rank, port)
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.parallel.tensor_parallel import tensor_parallel
    full = optimize_model(_tiny(2, 256 * WORLD, 512 * WORLD), low_bit=low_bit)     # k-quant blocks need 256-wide column shards
    ids = torch.load(os.path.join(tmp, "ids.pt"))
    with torch.no_grad():
        ref = full(ids).logits
    sharded = tensor_parallel(copy.deepcopy(full))
    att = sharded.model.layers[0].self_attn
    assert att.qkv_proj.out_len == full.model.layers[0].self_attn.qkv_proj.out_len // WORLD
    assert sharded.model.layers[0].mlp.down_proj.in_len == 512
    with torch.no_grad():
        out = sharded(ids).logits
        gen = sharded.generate(ids, max_new_tokens=6, do_sample=False)
        gen_ref = full.generate(ids, max_new_tokens=6, do_sample=False)
    rel = ((out - ref).abs().mean() / ref.abs().mean()).item()
    # same quantized weights, different summation order (bf16 layers round their partial sums before the reduce)
    assert rel < (2e-2 if low_bit == "bf16" else 2e-3), rel
    assert low_bit == "bf16" or torch.equal(gen, gen_ref)
    dist.barrier()
    dist.destroy_process_group()


def _tp_kv_replica_worker(rank, port, tmp):
    """fewer kv heads than ranks: every rank keeps the kv head its query heads belong to"""
    _boot(rank, port)
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.parallel.tensor_parallel import tensor_parallel
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=320, hidden_size=128 * WORLD, intermediate_size=256 * WORLD, num_hidden_layers=2,
                      num_attention_heads=2 * WORLD, num_key_value_heads=1, max_position_embeddings=256,
                      tie_word_embeddings=False)
    full = optimize_model(LlamaForCausalLM(cfg).eval(), low_bit="sym_int8")
    ids = torch.load(os.path.join(tmp, "ids.pt"))
    with torch.no_grad():
        ref = full(ids).logits
        out = tensor_parallel(copy.deepcopy(full))(ids).logits
    rel = ((out - ref).abs().mean() / ref.abs().mean()).item()
    assert rel < 2e-3, rel
    dist.barrier()
    dist.destroy_process_group()


def test_tensor_parallel_replicates_kv_heads_when_ranks_outnumber_them(tmp_path):
    torch.manual_seed(1)
    torch.save(torch.randint(0, 320, (2, 9)), tmp_path / "ids.pt")
    _run(_tp_kv_replica_worker, str(tmp_path))


@pytest.mark.parametrize("low_bit", ["sym_int4", "q4_k", "bf16"])
def test_tensor_parallel_matches_single_process(tmp_path, low_bit):
    torch.manual_seed(1)
    torch.save(torch.randint(0, 320, (2, 9)), tmp_path / "ids.pt")
    _run(_tp_worker, low_bit, str(tmp_path))


# ------------------------------------------------------------------------------------------------
def _pp_worker(rank, port, tmp):
    _boot(rank, port)
    from bigdl_amd.optimize import optimize_model
    from bigdl_amd.transformers.pipeline_parallel import pipeline_parallel
    full = optimize_model(_tiny(2 * WORLD), low_bit="sym_int8")
    ids = torch.load(os.path.join(tmp, "ids.pt"))
    ref = full.generate(ids, max_new_tokens=8, do_sample=False)
    long_ids = ids.repeat(1, 3)                                   # > 10 tokens: lm_head keeps the last position only
    ref_long = full.generate(long_ids, max_new_tokens=4, do_sample=False)
    stage = pipeline_parallel(copy.deepcopy(full), WORLD, torch.float32)
    assert len(stage.model.layers) == 2 and stage.layer_start == 2 * rank
    out = stage.generate(ids, max_new_tokens=8)
    assert torch.equal(out, ref), (out, ref)
    out_long = stage.generate(long_ids, max_new_tokens=4)
    assert torch.equal(out_long, ref_long)
    assert stage.first_token_time > 0 and len(stage.next_token_time) == 3
    # serving worker: two requests of different lengths in one micro-batch
    from bigdl_amd.transformers.pipeline_parallel import PPModelWorker
    import asyncio
    w = PPModelWorker(None, rank, WORLD, "sym_int8", max_num_seqs=4, torch_dtype=torch.float32, model=stage)

    async def drive():
        if rank == 0:
            await w.add_request("a", ids[0].tolist(), 5)
            await w.add_request("b", ids[1, 3:].tolist(), 5)
        res = await w.run_until_idle()
        w.shutdown()
        return res
    res = asyncio.run(drive())
    single = full.generate(ids[1:2, 3:], max_new_tokens=5, do_sample=False)
    if rank == 0:
        got = res["batch_1"]
        assert got.shape == (2, 5)
        assert torch.equal(got[0], ref[0, ids.shape[1]:ids.shape[1] + 5])       # unpadded row equals plain generation
        assert torch.equal(got[1], single[0, -5:])                              # left-padded row too
        toks = []
        while not w.streamer["a"].empty():
            toks.append(w.streamer["a"].get_nowait())
        assert [t for t, _ in toks] == got[0].tolist() and toks[-1][1] is True
    assert not w.kv_caches and not w.positions                                  # FREE reached every stage
    # three micro-batches of one request each: two are in flight at any time (one per stage), the third joins as soon as
    # one retires, and no stage waits for a whole batch to finish before it sees the next
    w2 = PPModelWorker(None, rank, WORLD, "sym_int8", max_num_seqs=1, torch_dtype=torch.float32, model=stage)
    budgets = {"x": 6, "y": 4, "z": 5}
    prompts = {"x": ids[0].tolist(), "y": ids[1].tolist(), "z": ids[0, 2:].tolist()}

    async def drive3():
        if rank == 0:
            for k in budgets:
                await w2.add_request(k, prompts[k], budgets[k])
        res = await w2.run_until_idle()
        w2.shutdown()
        return res
    res = asyncio.run(drive3())
    if rank == 0:
        assert w2.max_inflight == WORLD
        for serial, k in enumerate(budgets, start=1):
            want = full.generate(torch.tensor([prompts[k]]), max_new_tokens=budgets[k], do_sample=False)[0, len(prompts[k]):]
            assert torch.equal(res[f"batch_{serial}"][0], want), k
    # every stage alternates between the two micro-batches in flight: x0 y0 x1 y1 x2 y2 x3 y3, then z takes y's place
    assert w2.trace[:8] == [(1, 0), (2, 0), (1, 1), (2, 1), (1, 2), (2, 2), (1, 3), (2, 3)], w2.trace
    assert w2.trace[8:12] == [(1, 4), (3, 0), (1, 5), (3, 1)], w2.trace
    assert len(w2.trace) == sum(budgets.values())
    dist.barrier()
    dist.destroy_process_group()


def test_pipeline_parallel_generate_matches_single_process(tmp_path):
    torch.manual_seed(2)
    torch.save(torch.randint(1, 320, (2, 7)), tmp_path / "ids.pt")
    _run(_pp_worker, str(tmp_path))


# ------------------------------------------------------------------------------------------------
def _comm_worker(rank, port):
    _boot(rank, port)
    from bigdl_amd.parallel import all_reduce_coalesced_, new_groups
    g = new_groups(tp=2, pp=1)
    assert g["coords"] == (rank // 2, 0, rank % 2)          # data-parallel index, stage, tensor-parallel index
    ts = [torch.full((3, 5), float(rank + 1)), torch.full((7,), float(rank + 1)), torch.ones(4, dtype=torch.float64) * rank]
    all_reduce_coalesced_(ts, bucket_mb=1, average=True)
    mean = (WORLD + 1) / 2                                   # mean of rank + 1 over the world
    assert torch.allclose(ts[0], torch.full((3, 5), mean)) and torch.allclose(ts[1], torch.full((7,), mean))
    assert torch.allclose(ts[2], torch.full((4,), mean - 1, dtype=torch.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_groups_and_bucketed_all_reduce():
    _run(_comm_worker)


# ------------------------------------------------------------------------------------------------
def _ep_worker(rank, port):
    _boot(rank, port)
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.moe import MoEWeights
    torch.manual_seed(0)                                     # same experts on every rank, then sharded
    e, k, h, inter = 4, 2, 64, 96
    router = torch.randn(e, h) * 0.3
    gate, up, down = torch.randn(e, inter, h) * 0.1, torch.randn(e, inter, h) * 0.1, torch.randn(e, h, inter) * 0.1

    def make():
        m = MoEWeights(router, gate.clone(), up.clone(), down.clone(), Q.FP16, e, k, h, inter)
        m.router = router.clone()                            # fp32 on the CPU
        return m
    full = make()
    torch.manual_seed(100)
    x_all = torch.randn(WORLD * 5, h)                        # 5 tokens per rank
    ids, w = full.route(x_all, local_only=False)
    want = full.local_ffn(x_all, ids, w)                     # single-process result for every token
    mine = x_all[rank * 5:(rank + 1) * 5].contiguous()
    # tokens sharded: dispatch / return with all_to_all (prefill form), gather / reduce-scatter (decode form)
    ep = make().shard_experts(rank, WORLD, None, mode="a2a")
    assert ep.n_local == e // WORLD and ep.gate.shape[0] == e // WORLD
    torch.testing.assert_close(ep.forward_a2a(mine), want[rank * 5:(rank + 1) * 5], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ep.forward_gather_scatter(mine), want[rank * 5:(rank + 1) * 5], rtol=1e-4, atol=1e-5)
    # fixed-shape dispatch (no split sizes on the host): exact when the capacity covers the worst case ...
    ep.ep_capacity_factor = float(WORLD)
    torch.testing.assert_close(ep.forward_a2a(mine), want[rank * 5:(rank + 1) * 5], rtol=1e-4, atol=1e-5)
    assert int(ep.ep_dropped) == 0
    # ... and with a tight capacity the rows over it are dropped, counted, and only their terms are missing
    ep.ep_capacity_factor = 0.4                              # capacity = ceil(0.4 * 10 / WORLD) rows per peer
    cap = -(-int(0.4 * 10 * 1000) // (WORLD * 1000))
    got = ep.forward_a2a(mine)
    ids_m, w_m = full.route(mine, local_only=False)
    dest = (ids_m.reshape(-1) // ep.n_local)
    order = torch.argsort(dest, stable=True)
    seen, kept = {}, torch.zeros(10, dtype=torch.bool)
    for j in order.tolist():
        d = int(dest[j])
        seen[d] = seen.get(d, 0) + 1
        kept[j] = seen[d] <= cap
    assert int(ep.ep_dropped) == int((~kept).sum()) > 0
    partial = full.local_ffn(mine, torch.where(kept.view(5, 2), ids_m, torch.full_like(ids_m, -1)), w_m)
    torch.testing.assert_close(got, partial, rtol=1e-4, atol=1e-5)
    ep.ep_capacity_factor = None
    # tokens replicated: partial sums + all-reduce
    rep = make().shard_experts(rank, WORLD, None, mode="allreduce")
    ids_l, w_l = rep.route(x_all, local_only=True)
    part = rep.local_ffn(x_all, ids_l, w_l)
    dist.all_reduce(part)
    torch.testing.assert_close(part, want, rtol=1e-4, atol=1e-5)
    dist.barrier()
    dist.destroy_process_group()


def test_expert_parallel_all_to_all_matches_single_process():
    _run(_ep_worker)


def _zero_worker(rank, port, tmp):
    from bigdl_amd.parallel.zero import ZeroOptimizer
    _boot(rank, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(24, 37), torch.nn.Tanh(), torch.nn.Linear(37, 5))
    ref = copy.deepcopy(net)
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=1e-2)
    data = torch.randn(4, WORLD, 6, 24)
    target = torch.randn(4, WORLD, 6, 5)
    for stage3 in (False, True):
        model = copy.deepcopy(net)
        zo = ZeroOptimizer(model.parameters(), lambda ps: torch.optim.AdamW(ps, lr=1e-2), max_grad_norm=None,
                           shard_params=stage3).attach(model)
        single = copy.deepcopy(ref)
        s_opt = torch.optim.AdamW(single.parameters(), lr=1e-2)
        for it in range(4):
            # each rank sees its own slice of the global batch; the single-process run sees all of it
            loss = torch.nn.functional.mse_loss(model(data[it, rank]), target[it, rank])
            loss.backward()
            zo.step()
            if stage3:
                assert all(p.numel() == 0 for p in model.parameters())     # working copy released between steps
            s_loss = torch.nn.functional.mse_loss(single(data[it].reshape(-1, 24)), target[it].reshape(-1, 5))
            s_opt.zero_grad()
            s_loss.backward()
            s_opt.step()
        zo.gather_params()
        for a, b in zip(model.parameters(), single.parameters()):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
        # optimizer state lives on 1/world of the parameters
        n = sum(p.numel() for p in single.parameters())
        assert zo.master.numel() == (n + WORLD - 1) // WORLD
        # shard checkpoint round trip
        path = os.path.join(tmp, f"zero_{int(stage3)}_rank{rank}.pt")
        torch.save(zo.state_dict(), path)
        zo.load_state_dict(torch.load(path, weights_only=True))
    dist.destroy_process_group()


def test_zero_sharded_optimizer_matches_single_process(tmp_path):
    """ZeRO stage 1/2 (sharded optimizer state + reduce-scattered gradients) and stage 3 (parameters released between
    steps) reproduce single-process AdamW on the global batch."""
    _run(_zero_worker, str(tmp_path))


def _cp_worker(rank, port):
    from bigdl_amd.parallel import context_parallel as cp
    _boot(rank, port)
    torch.manual_seed(5)
    b, hkv, d, n = 1, 2, 8, 3
    total = 2 * WORLD * n
    full_k = torch.randn(b, hkv, total + 4, d)
    full_v = torch.randn(b, hkv, d, total + 4)
    for dtype in (torch.float32, torch.uint8):
        fk = full_k if dtype == torch.float32 else (full_k * 40).to(torch.uint8)
        fv = full_v if dtype == torch.float32 else (full_v * 40).to(torch.uint8)
        kc, vc = torch.zeros_like(fk), torch.zeros_like(fv)
        for c in cp.zigzag_chunks(WORLD, rank):            # this rank only has its own two chunks
            kc[:, :, c * n:(c + 1) * n] = fk[:, :, c * n:(c + 1) * n]
            vc[..., c * n:(c + 1) * n] = fv[..., c * n:(c + 1) * n]
        cp.gather_kv_chunks(kc, vc, n, None)
        assert torch.equal(kc[:, :, :total], fk[:, :, :total]) and torch.equal(vc[..., :total], fv[..., :total])
        assert float(kc[:, :, total:].float().abs().sum()) == 0.0       # nothing past the prompt is touched
    idx = cp.local_token_index(total, WORLD, rank)
    got = [torch.empty_like(idx) for _ in range(WORLD)]
    dist.all_gather(got, idx)
    assert sorted(torch.cat(got).tolist()) == list(range(total))        # the chunks tile the prompt exactly
    t = torch.full((3,), float(rank + 7))
    cp.broadcast_from_last_token_owner(t)
    assert t.tolist() == [7.0] * 3
    dist.destroy_process_group()


def test_context_parallel_kv_gather():
    _run(_cp_worker)


# ------------------------------------------------------------------------------------------------
class _FakeArLib:
    """Stands in for the native library's IPC entry points: buffers are host allocations, handles are their addresses.
    ``fail`` names the step that fails on this rank."""

    def __init__(self, fail=None):
        import ctypes
        self.fail, self.live, self.opened, self._ct = fail, {}, 0, ctypes

        class _Fn:
            def __init__(s, f):
                s.f = f

            def __call__(s, *a):
                return s.f(*a)
        for name in ("bigdl_ar_buffer_bytes", "bigdl_ar_alloc", "bigdl_ar_free", "bigdl_ar_ipc_handle", "bigdl_ar_ipc_open",
                     "bigdl_ar_ipc_close", "bigdl_allreduce_oneshot", "bigdl_ar2_buffer_bytes", "bigdl_allreduce_twoshot",
                     "bigdl_ar2_epoch_words"):
            setattr(self, name, _Fn(getattr(self, "_" + name, lambda *a: 0)))

    def _bigdl_ar_buffer_bytes(self, world, n):
        return world * n + 64

    _bigdl_ar2_buffer_bytes = _bigdl_ar_buffer_bytes

    def _bigdl_ar2_epoch_words(self):
        return 2

    def _bigdl_ar_alloc(self, n, out):
        if self.fail == "alloc":
            return 1
        buf = self._ct.create_string_buffer(64)
        self.live[self._ct.addressof(buf)] = buf
        out._obj.value = self._ct.addressof(buf)
        return 0

    def _bigdl_ar_free(self, p):
        self.live.pop(p.value, None)

    def _bigdl_ar_ipc_handle(self, p, handle):
        return 1 if self.fail == "handle" else 0

    def _bigdl_ar_ipc_open(self, h, out):
        if self.fail == "open":
            return 1
        self.opened += 1
        out._obj.value = 4096 * self.opened
        return 0

    def _bigdl_ar_ipc_close(self, p):
        self.opened -= 1


def _ipc_setup_worker(rank, port, fail):
    _boot(rank, port)
    from bigdl_amd.ops import hip
    from bigdl_amd.parallel import xgmi_allreduce as xa
    fake = _FakeArLib(fail if rank == WORLD - 1 else None)
    hip.lib = lambda: fake
    if fail is None:
        ar = xa.XgmiAllReduce(device="cpu")
        assert len(fake.live) == 2 and fake.opened == 2 * (WORLD - 1)
        ar.close()
    elif fail == "second":                     # one-shot builds everywhere, the two-shot member fails on one rank
        real = xa.TwoShotAllReduce._alloc_and_export

        def broken(self):
            if rank == WORLD - 1:
                raise RuntimeError("no room for the large buffer")
            return real(self)
        xa.TwoShotAllReduce._alloc_and_export = broken
        with pytest.raises(RuntimeError, match="no room for the large buffer"):
            xa.XgmiAllReduce(device="cpu")
    else:                                      # one rank fails a step: every rank raises, naming that rank
        with pytest.raises(RuntimeError, match=f"rank {WORLD - 1}"):
            xa.OneShotAllReduce(device="cpu")
    assert not fake.live and fake.opened == 0, (fake.live, fake.opened)     # nothing left mapped or allocated
    dist.barrier()                             # the group is still in step after the failure
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [None, "alloc", "handle", "open", "second"])
def test_ipc_allreduce_setup_fails_on_every_rank_together(fail):
    """A set-up error on one rank must not leave its peers inside a collective it never enters
    (parallel/xgmi_allreduce.py: every phase ends in an exchange and all ranks raise after it)."""
    _run(_ipc_setup_worker, fail)
