"""Reference-format checkpoints (basicsr/models/base_model.py:214-261 save_network, :299-326 load_network,
:328-373 training state): the file layout the reference's inference script and trainer read / write."""
import torch

import wave_mamba_amd as wm
from wave_mamba_amd import trainer

CFG = dict(in_chn=3, wf=8, n_l_blocks=[1, 1, 1], n_h_blocks=[1, 1, 1], ffn_scale=2.0)


class _Wrapped(torch.nn.Module):
    """State-dict keys carry a 'module.' prefix, as a checkpoint written from inside a DDP wrapper would."""

    def __init__(self, net):
        super().__init__()
        self.module = net


def test_save_load_round_trip(tmp_path):
    torch.manual_seed(0)
    a, b = wm.WaveMamba(**CFG), wm.WaveMamba(**CFG)
    assert any(not torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    path = trainer.save_network(a, str(tmp_path / "models" / "net_g_latest.pth"))
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert set(blob) == {"params", "iter", "epoch"} and blob["iter"] == "latest" and blob["epoch"] == 0
    assert list(blob["params"]) == list(a.state_dict())                           # same keys, same order
    assert all(v.device.type == "cpu" for v in blob["params"].values())
    missing, unexpected, skipped = trainer.load_network(b, path)
    assert (missing, unexpected, skipped) == ([], [], [])
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    # the reference's inference script reads ['params'] straight into the registered arch (strict)
    wm.WaveMamba(**CFG).load_state_dict(blob["params"], strict=True)


def test_module_prefix_and_param_key_fallback(tmp_path):
    torch.manual_seed(1)
    a, b = wm.WaveMamba(**CFG), wm.WaveMamba(**CFG)
    p = str(tmp_path / "w.pth")
    torch.save({"params": _Wrapped(a).state_dict()}, p)                            # 'module.'-prefixed keys
    assert all(k.startswith("module.") for k in torch.load(p, weights_only=False)["params"])
    assert trainer.load_network(b, p, param_key="params_ema") == ([], [], [])     # falls back to 'params'
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))
    p2 = trainer.save_network(a, str(tmp_path / "it.pth"), current_iter=500, epoch=3, param_key="params_ema")
    blob = torch.load(p2, weights_only=False)
    assert blob["iter"] == 500 and blob["epoch"] == 3 and "params_ema" in blob
    torch.save(a.state_dict(), p)                                                  # bare state dict: param_key None
    assert trainer.load_network(b, p, param_key=None) == ([], [], [])


def test_non_strict_skips_shape_mismatch(tmp_path):
    torch.manual_seed(2)
    a = wm.WaveMamba(**CFG)
    b = wm.WaveMamba(**dict(CFG, in_chn=1))
    p = trainer.save_network(a, str(tmp_path / "a.pth"))
    before = {k: v.clone() for k, v in b.state_dict().items()}
    missing, unexpected, skipped = trainer.load_network(b, p, strict=False)
    assert skipped and unexpected == [] and sorted(missing) == skipped
    sa, sb = a.state_dict(), b.state_dict()
    for k in sb:
        if k in skipped:
            assert torch.equal(sb[k], before[k])
        else:
            assert torch.equal(sb[k], sa[k])
    try:
        trainer.load_network(b, p, strict=True)
    except RuntimeError:
        pass
    else:
        raise AssertionError("strict load of mismatching shapes must raise")


def test_training_state_resume(tmp_path):
    torch.manual_seed(3)
    net = torch.nn.Linear(4, 3)
    opt = trainer.make_optimizer(net)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2, 4], gamma=0.5)
    for _ in range(3):
        opt.zero_grad()
        net(torch.randn(5, 4)).square().mean().backward()
        opt.step()
        sched.step()
    assert trainer.save_training_state(str(tmp_path / "x.state"), 1, -1, [opt], [sched]) is None
    path = trainer.save_training_state(str(tmp_path / "training_states" / "3.state"), 1, 3, [opt], [sched])
    st = torch.load(path, weights_only=False)
    assert set(st) == {"epoch", "iter", "optimizers", "schedulers"} and st["iter"] == 3
    net2 = torch.nn.Linear(4, 3)
    net2.load_state_dict(net.state_dict())
    opt2 = trainer.make_optimizer(net2)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=[2, 4], gamma=0.5)
    assert trainer.resume_training(path, [opt2], [sched2]) == (1, 3)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] == 5e-4 * 0.5
    x = torch.randn(5, 4)
    for n, o in ((net, opt), (net2, opt2)):
        o.zero_grad()
        n(x).square().mean().backward()
        o.step()
    assert all(torch.equal(p, q) for p, q in zip(net.parameters(), net2.parameters()))      # moments were restored
    try:
        trainer.resume_training(st, [opt2, opt2], [sched2])
    except ValueError:
        pass
    else:
        raise AssertionError("optimizer count mismatch must raise")


# ---- the optimizer step itself (femasr_model.py:122-135 setup_optimizers, :157-185 optimize_parameters) ---------------
def _opt_golden():
    import os
    import numpy as np
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "train_grads_wf8.npz")), np.load(os.path.join(GOLDEN, "train_opt_wf8.npz"))


def _net_with_golden_grads(device):
    """The wf = 8 model at the reference's weights with the REFERENCE's gradients of step 1 in .grad."""
    g, o = _opt_golden()
    torch.manual_seed(0)
    net = wm.WaveMamba(in_chn=3, wf=8, n_l_blocks=[1, 1, 2], n_h_blocks=[1, 1, 1], ffn_scale=2.0).train()
    net.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}, strict=False)
    net = net.to(device)
    for k, p in net.named_parameters():
        p.grad = torch.from_numpy(g["g." + k]).to(device)
    return net, g, o


def check_optimizer_arithmetic(device):
    """trainer.make_optimizer + step on the reference's own gradients reproduces the reference's parameters after
    optimizer_g.step() (tests/golden/train_opt_wf8.npz, make_golden_grads.py): AdamW lr 5e-4, weight decay 1e-3, betas
    (0.9, 0.99), every parameter in the one group - <= 1e-6 relative per tensor (measured: bit-equal or 1 ulp)."""
    net, g, o = _net_with_golden_grads(device)
    opt = trainer.make_optimizer(net)
    assert len(opt.param_groups) == 1 and len(opt.param_groups[0]["params"]) == len(list(net.parameters()))
    opt.step()
    worst = 0.0
    for k, p in net.named_parameters():
        ref = torch.from_numpy(o["p1." + k]).double()
        worst = max(worst, float((p.detach().cpu().double() - ref).norm() / ref.norm().clamp_min(1e-30)))
    assert worst <= 1e-6, f"parameters after the optimizer step deviate by {worst:.3e}"
    return worst


def test_optimizer_step_matches_reference_on_reference_gradients():
    check_optimizer_arithmetic("cpu")
