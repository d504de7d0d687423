"""Graph-mode decoding.  The reference keeps its native ops traceable for ``torch.compile``
(python/llm/example/GPU/GraphMode/gpt2-graph-mode-benchmark.py, fake kernels in ipex_llm/transformers/xpu_ops.py); on
MI355X the decode step is captured once into a hipGraph and replayed per token.  This script times both ways of
running the same engine, and shows that the registered ``torch.library`` ops trace under ``torch.compile``."""
import argparse

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.models.decoder import TINY_LLAMA, DecoderConfig, build_random_decoder

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-predict", type=int, default=128)
    a = ap.parse_args()
    eng = build_random_decoder(DecoderConfig(**TINY_LLAMA), Q.SYM_INT4, device="cuda")
    ids = torch.randint(3, 32000, (1, 256)).cuda()
    for graph in (False, True):
        eng.use_graph = graph
        eng._graphs.clear()
        eng.generate(ids, max_new_tokens=8)
        eng.generate(ids, max_new_tokens=a.n_predict)
        print(f"hipGraph={graph}: {eng.rest_cost_mean * 1e3:.3f} ms/token")
    from bigdl_amd.ops import torch_ops
    torch_ops.register()                     # bigdl_amd::* ops with fake kernels
    x = torch.randn(4, 2048, device="cuda", dtype=torch.float16)
    w = torch.ones(2048, device="cuda", dtype=torch.float16)
    fn = torch.compile(lambda t: torch.ops.bigdl_amd.rms_norm(w, t, 1e-5) * 2, backend="aot_eager")
    print("compiled op ok:", torch.allclose(fn(x), torch.ops.bigdl_amd.rms_norm(w, x, 1e-5) * 2))
