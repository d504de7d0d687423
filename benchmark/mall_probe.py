"""Does a weight matrix that was read shortly before (and therefore sits in the 256 MiB memory-side cache) stream
faster into the decode GEMV than one coming from HBM?  Cold: rotate over 600 MB of copies.  Warm: alternate between
two copies (2 x size < 256 MiB, > aggregate L2 for the larger shapes)."""
import json

import torch

from bigdl_amd import qtypes as Q
from bigdl_amd.ops import hip, host
from micro_kernels import timed_graph

qt = Q.SYM_INT4
for name, (n, k) in {"o": (4096, 4096), "qkv": (12288, 4096), "down": (4096, 11008), "gate_up": (11008, 4096)}.items():
    nbytes = host.device_nbytes(qt, n, k)
    for mode, copies in (("cold", max(2, int(600e6 // nbytes))), ("warm", 2)):
        ws = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device="cuda") for _ in range(copies)]
        x = torch.randn(1, k, device="cuda").half()
        out = torch.empty(1, n, device="cuda", dtype=torch.float16)
        st = {"i": 0}

        def fn():
            w = ws[st["i"] % copies]
            st["i"] += 1
            if name == "gate_up":
                hip.gemv_lowbit(x, w, qt, n, k, w2=ws[st["i"] % copies], act="silu", out=out)
            else:
                hip.gemv_lowbit(x, w, qt, n, k, out=out)
        us = timed_graph(fn)
        print(json.dumps({"shape": name, "mode": mode, "us": round(us, 2), "MB": round(nbytes / 1e6, 1)}), flush=True)
        del ws
