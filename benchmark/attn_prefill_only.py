import sys, torch
from bigdl_amd.ops import hip
s = int(sys.argv[1]); d = 128
q, k = (torch.randn(1, 32, s, d, device="cuda").half() for _ in range(2))
vt = torch.randn(1, 32, d, s, device="cuda").half()
for _ in range(5):
    hip.attn_prefill(q, k, vt, scale=d ** -0.5, causal=True, v_transposed=True)
torch.cuda.synchronize()
