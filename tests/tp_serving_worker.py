"""Worker of tests/test_allreduce_gpu.py::test_tensor_parallel_serving: two ranks of a gloo group share GPU 0, each holds
a tensor-parallel shard of the engine and a ContinuousBatcher; rank 0 feeds requests, the schedule travels as
fixed-shape tensors."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    # a rank that stops making progress says where (all thread stacks) and exits instead of holding the test
    import faulthandler
    import threading
    deadline = int(os.environ.get("TP_WORKER_DEADLINE_S", "100"))
    state = {}

    def report():
        tp = state.get("tp")
        if tp is not None:
            b = tp.batcher
            sys.stderr.write(f"[rank {os.environ['RANK']}] stuck: tp.steps={tp.steps} batcher.steps={b.steps} running="
                             f"{sorted((s, len(r.output_ids)) for s, r in b.running.items())} waiting={len(b.waiting)} "
                             f"prefilling={b._prefilling is not None} header={tp._header.tolist()}\n")
        faulthandler.dump_traceback(all_threads=True)
        sys.stderr.flush()
        os._exit(3)

    timer = threading.Timer(deadline, report)
    timer.daemon = True
    timer.start()
    out_path = sys.argv[1]
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    from bigdl_amd import qtypes as Q
    from bigdl_amd.models.decoder import DecoderConfig, build_random_decoder
    from bigdl_amd.parallel.tensor_parallel import shard_engine
    from bigdl_amd.serving.continuous import ContinuousBatcher, SeqRequest
    from bigdl_amd.serving.tp import TensorParallelBatcher
    cfg = DecoderConfig(vocab_size=2048, hidden_size=512, intermediate_size=1536, num_layers=4, num_heads=8, num_kv_heads=4,
                        head_dim=64, max_position=2048)
    g = torch.Generator().manual_seed(9)
    lens = [33, 9, 21, 40, 17, 12]
    budgets = [10, 7, 12, 6, 9, 30]
    prompts = [torch.randint(3, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
    res = {}
    if rank == 0:       # what one unsharded engine serves for the same arrivals (request 5 is aborted later, not compared)
        whole = build_random_decoder(cfg, Q.SYM_INT4, device="cuda:0", seed=7)
        res["alone"] = [whole.generate(torch.tensor([p]), max_new_tokens=b)[0, len(p):].cpu().tolist()
                        for p, b in zip(prompts[:5], budgets[:5])]
        del whole
    fast = os.environ.get("TP_FAST", "1") == "1"
    eng = shard_engine(build_random_decoder(cfg, Q.SYM_INT4, device="cuda:0", seed=7), rank, world, fast_allreduce=fast)
    tp = TensorParallelBatcher(eng, max_num_seqs=4, max_seq_len=256, max_admit_per_step=2, max_batched_tokens=64)
    state["tp"] = tp
    res["graph"] = bool(eng.use_graph and eng._reduce_capturable())
    if rank == 0:
        reqs = [SeqRequest(f"r{i}", p, max_new_tokens=b) for i, (p, b) in enumerate(zip(prompts, budgets))]
        arrivals = {0: [0, 1], 3: [2, 3, 4], 5: [5]}
        step = 0
        while step < 6 or tp.has_work():
            for i in arrivals.get(step, []):
                tp.submit(reqs[i])
            if step == 13:
                tp.abort("r5")
            tp.step()
            step += 1
            assert step < 200
        tp.shutdown()
        res["served"] = [r.output_ids for r in reqs]
        res["reasons"] = [r.finish_reason for r in reqs]
    else:
        tp.follow()
    res["steps"] = tp.steps
    res["payload_steps"] = tp.payload_steps
    res["batcher_steps"] = tp.batcher.steps
    res["generated"] = tp.batcher.generated
    res["left"] = len(tp.requests) + len(tp.batcher.running) + len(tp.batcher.waiting)
    if eng.tp_fast_allreduce is not None:
        eng.tp_fast_allreduce.check()
    with open(f"{out_path}.{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    if eng.tp_fast_allreduce is not None:
        eng.tp_fast_allreduce.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
