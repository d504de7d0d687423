import torch

from bigdl_amd.transformers.galore import GaLoreAdamW, galore_param_groups


def test_galore_trains_with_low_rank_state():
    torch.manual_seed(0)
    net = torch.nn.Sequential()
    net.add_module("attn", torch.nn.Linear(64, 96))
    net.add_module("act", torch.nn.Tanh())
    net.add_module("mlp", torch.nn.Linear(96, 48))
    net.add_module("head", torch.nn.Linear(48, 4))
    groups = galore_param_groups(net, ["attn", "mlp"], rank=8, update_proj_gap=5, scale=1.0)
    assert len(groups[1]["params"]) == 2 and len(groups[0]["params"]) == 4     # two matrices; biases + head stay dense
    opt = GaLoreAdamW(groups, lr=1e-2)
    x, y = torch.randn(256, 64), torch.randn(256, 4)
    losses = []
    for _ in range(60):
        loss = torch.nn.functional.mse_loss(net(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.8 * losses[0]
    # moments of the projected matrices are rank x n (or m x rank), not m x n
    st = opt.state[net.attn.weight]
    assert st["exp_avg"].shape == (96, 8) and st["projector"].ortho.shape == (8, 64)      # [96, 64]: right projection
    st = opt.state[net.mlp.weight]
    assert st["exp_avg"].shape == (8, 96) and st["projector"].ortho.shape == (48, 8)      # [48, 96]: left projection
    dense_adam = 2 * sum(p.numel() for p in net.parameters())
    assert opt.state_numel() < 0.45 * dense_adam


def test_projector_is_the_truncated_svd():
    from bigdl_amd.transformers.galore import _Projector
    torch.manual_seed(1)
    for shape in ((12, 20), (20, 12)):
        g = torch.randn(*shape)
        full = _Projector(rank=12, update_proj_gap=10, scale=1.0, proj_type="std")
        torch.testing.assert_close(full.project_back(full.project(g, 0)), g, rtol=1e-4, atol=1e-4)
        low = _Projector(rank=3, update_proj_gap=10, scale=1.0, proj_type="std")
        u, s_, vh = torch.linalg.svd(g, full_matrices=False)
        best = (u[:, :3] * s_[:3]) @ vh[:3]
        torch.testing.assert_close(low.project_back(low.project(g, 0)), best, rtol=1e-4, atol=1e-4)
        # the basis is kept between refreshes
        basis = low.ortho.clone()
        low.project(torch.randn(*shape), 3)
        assert torch.equal(low.ortho, basis)
        low.project(torch.randn(*shape), 10)
        assert not torch.equal(low.ortho, basis)
