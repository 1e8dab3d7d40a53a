"""Whole-model parity of the HIP product against goldens produced by the reference (and the oracle).

fp32 "parity mode": tensors within 1e-4 relative, decoded payload bits exact (north_star).
bf16 mode: payload bits exact vs the fp32 oracle, deviation of tensors reported and bounded.
"""
import importlib

import numpy as np
import pytest
import torch

from oracle import uformer_wm as O
from synth import load_synth, synth_bits, synth_tile

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def wm():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return importlib.import_module("image-in-speech-watermarking_amd")


@pytest.fixture(scope="module")
def net(wm):
    m = wm.UformerAudio(img_size=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff",
                        depths=[1, 2, 8, 8, 2, 8, 8, 2, 1], modulator=True, dd_in=2, in_chans=2, audio_scale="0")
    load_synth(m)
    return m.cuda()


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def test_state_dict_is_reference_compatible(net, golden):
    ref = O.UformerAudio()
    sd = net.state_dict()
    assert set(sd) == set(ref.state_dict())
    ref.load_state_dict({k: v.cpu() for k, v in sd.items()})      # shapes and names line up both ways
    assert all(p.is_cuda for p in net.parameters())


# the two modes held to north_star's bound (payload bits exact, 1e-4 relative on the spectrogram / loss tensors): "fp32" (fp32 MFMA,
# exact accumulation) and "split" (fp32 storage, every GEMM as three bf16 MFMA products of hi / lo halves, round 3)
PARITY_MODES = ["fp32", "split"]


@pytest.mark.parametrize("pmode", PARITY_MODES)
def test_g3_eval_forward_matches_reference_golden(net, golden, pmode):
    g = golden("g3_model_eval")
    load_synth(net)
    net.eval()
    net.compute_mode = pmode
    x, msg = synth_tile(2, "g3_x").cuda(), synth_bits("g3_msg", (2, 1, 32, 32)).cuda()
    with torch.no_grad():
        out = net(x, msg)
    for o, k in zip(out, ("stft_new", "noise", "wm_pred", "wm")):
        assert rel(o, g[k]) <= 1e-4, (k, rel(o, g[k]))
    tot, terms = O.train_losses([o.cpu() for o in out], x.cpu(), msg.cpu())
    np.testing.assert_allclose(torch.stack(terms).numpy(), g["terms"], rtol=1e-4)
    bits = out[3].round().clamp(0, 1).to(torch.uint8).cpu().numpy()
    assert np.array_equal(bits, g["bits"])                         # decoded payload: bit exact
    with torch.no_grad():
        assert rel(net.wm_decode(x), g["wm_decode_x"]) <= 1e-4
        y, wmp = net.feature_extract(x, msg)
    assert rel(y[:, :, :8, :8], g["fe_y"]) <= 1e-4 and rel(wmp, g["fe_wm"]) <= 1e-4
    net.compute_mode = "fp32"


def _scales_from_golden_masks(net, masks):
    """Golden masks are in DropPath call order and skip rate-0 blocks; the product wants [nslots, B] scales."""
    net._ensure_runtime(torch.device("cuda"))
    rates = net._rates_host
    out = torch.ones(len(rates), masks.shape[1])
    it = iter(masks)
    for s, r in enumerate(rates):
        if r > 0:
            out[s] = next(it) / (1.0 - r)
    assert next(it, None) is None
    return out.cuda()


@pytest.mark.parametrize("pmode", PARITY_MODES)
def test_g4_train_step_gradients_match_reference_golden(net, golden, wm, pmode):
    g = golden("g4_model_train")
    load_synth(net)
    net.train()
    net.compute_mode = pmode
    x, msg = synth_tile(2, "g4_x").cuda(), synth_bits("g4_msg", (2, 1, 32, 32)).cuda()
    scales = _scales_from_golden_masks(net, T(g["masks"]))
    net.zero_grad(set_to_none=True)
    out = net(x, msg, keep_scales=scales)
    tot, terms = O.train_losses(out, x, msg)                       # the trainer's own torch loss code
    np.testing.assert_allclose(torch.stack(terms).detach().cpu().numpy(), g["terms"], rtol=1e-4)
    assert rel(out[0][:, :, :16, :16], g["stft_new"]) <= 1e-4
    tot.backward()
    names = [str(n) for n in g["names"]]
    gn = np.array([net.get_parameter(n).grad.norm().item() for n in names])
    bad = [(n, a, b) for n, a, b in zip(names, gn, g["gnorm"]) if abs(a - b) > 2e-3 * b + 1e-6]
    assert not bad, bad[:10]
    for k in [k for k in g.files if k.startswith("g_")]:
        name = k[2:].replace("__", ".")
        assert rel(net.get_parameter(name).grad, g[k]) <= 2e-3, (name, rel(net.get_parameter(name).grad, g[k]))
    # one fused AdamW step reproduces torch.optim.AdamW (audio_uformer_stft.py:252-255)
    # AdamW's first step moves an element by lr * g / (|g| + eps): where |g| is within a few orders of eps = 1e-8 the step is a
    # steep function of g (a gradient that is a rounding error away from zero lands anywhere in +-lr), so those elements are only
    # held to the step bound; everywhere else the parameter is compared elementwise
    grads = {n: net.get_parameter(n).grad.detach().clone() for n in (k[2:].replace("__", ".") for k in g.files if k.startswith("p_"))}
    opt = wm.FusedAdamW(net, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    opt.step()
    for k in [k for k in g.files if k.startswith("p_")]:
        name = k[2:].replace("__", ".")
        got, want = net.get_parameter(name).detach().cpu().numpy(), g[k]
        firm = (grads[name].abs() > 1e-5).cpu().numpy()                    # |g| >= 1000 eps: the step is lr * sign(g) to 1e-3
        np.testing.assert_allclose(got[firm], want[firm], rtol=1e-4, atol=2e-6)
        assert firm.mean() > 0.5 and np.abs(got - want).max() <= 4.1e-4, (name, firm.mean(), np.abs(got - want).max())
    unused = net.get_parameter("encoder.input_proj.proj.0.weight")
    assert torch.equal(unused.detach().cpu(), load_synth(O.UformerAudio()).encoder.input_proj.proj[0].weight.detach())
    net.compute_mode = "fp32"


def test_bf16_mode_bits_and_deviation(net, golden):
    g = golden("g3_model_eval")
    load_synth(net)
    net.eval()
    net.compute_mode = "bf16"
    x, msg = synth_tile(2, "g3_x").cuda(), synth_bits("g3_msg", (2, 1, 32, 32)).cuda()
    try:
        with torch.no_grad():
            out = net(x, msg)
    finally:
        net.compute_mode = "fp32"
    dev = {k: rel(o, g[k]) for o, k in zip(out, ("stft_new", "noise", "wm_pred", "wm"))}
    bits = out[3].round().clamp(0, 1).to(torch.uint8).cpu().numpy()
    ber_vs_ref = float(np.mean(bits != g["bits"]))
    _record("r02_parity_bf16_g3_b2.json", {"max_rel_dev_vs_reference_golden": dev, "payload_bits_differing": ber_vs_ref})
    assert max(dev.values()) <= 3e-2, dev                          # measured 1.2e-2 (B = 6 twin below)
    assert ber_vs_ref <= 0.005, ber_vs_ref  # untrained synthetic net: sigmoid outputs within bf16 noise of 0.5 may flip


def test_attacked_forward_backward_runs_and_matches_oracle(net, wm):
    """Config 3 graph (attacks in the loop, extractor on the attacked spectrogram) against the oracle with the
    attack chain expressed through the same linear operators (noise-free so it is deterministic)."""
    from oracle import attacks as OA
    load_synth(net)
    net.eval()
    cfg = wm.attacks.AttackConfig(resample=True, snr_db=None, lowpass=True, amp=0.8)
    net.attack, net.extract_from = cfg, "attacked"
    x, msg = synth_tile(2, "g3_x").cuda(), synth_bits("g3_msg", (2, 1, 32, 32)).cuda()
    try:
        out = net(x, msg)
        loss, _ = O.train_losses(out, x, msg)
        net.zero_grad(set_to_none=True)
        loss.backward()
        gsum = float(net._flat_grad.abs().sum())
        assert np.isfinite(gsum) and gsum > 0
        ref = load_synth(O.UformerAudio()).eval()

        def attack(w):
            return torch.from_numpy(OA.attack_chain(w.detach().numpy(), resample=True, lowpass=True, amp=0.8)).float()
        with torch.no_grad():
            rout = ref(x.cpu(), msg.cpu(), attack=attack, extract_from="attacked")
        for o, r, k in zip(out, rout, ("stft_new", "noise", "wm_pred", "wm")):
            assert rel(o, r) <= 2e-4, (k, rel(o, r))
    finally:
        net.attack, net.extract_from = None, "y"


def test_train_step_graph_replay_equals_eager_and_learns(wm):
    """TrainStep (captured hipGraph) == the same launches issued eagerly, and the loss goes down."""
    def make():
        m = wm.UformerAudio(modulator=True, drop_path_rate=0.0)
        load_synth(m)
        m = m.cuda().train()
        return m, wm.FusedAdamW(m, lr=2e-4)
    x, msg = synth_tile(2, "ts_x").cuda(), synth_bits("ts_msg", (2, 1, 32, 32)).cuda()
    (m1, o1), (m2, o2) = make(), make()
    s1 = wm.TrainStep(m1, o1, 2, use_graph=True)
    s2 = wm.TrainStep(m2, o2, 2, use_graph=False)
    l1, l2 = [], []
    for _ in range(4):
        s1.step(x, msg)
        s2.step(x, msg)
        l1.append(s1.losses.cpu().clone())
        l2.append(s2.losses.cpu().clone())
    for a, b in zip(l1, l2):
        assert torch.allclose(a[:5], b[:5], rtol=2e-3, atol=1e-5), (a, b)     # atomics: order-dependent last bits
    assert all(torch.isfinite(a).all() for a in l1) and not torch.equal(l1[0], l1[-1])
    assert rel(m1._flat[:m1.optimized_numel], m2._flat[:m2.optimized_numel].cpu()) < 1e-3
    assert int(o1.step_t.item()) == 4


def test_train_step_segmented_graphs_equal_single_graph(wm):
    """The multi-rank form of the step (graphs cut at the gradient-bucket boundaries, collectives issued between the
    replays) run on one rank: same numbers as the single captured graph; a cut wherever a ~25 MB bucket completes or a
    segment ends, every element of the optimised gradient arena in exactly one planned collective."""
    def make():
        m = wm.UformerAudio(modulator=True, drop_path_rate=0.1)
        load_synth(m)
        m = m.cuda().train()
        return m, wm.FusedAdamW(m, lr=2e-4)
    x, msg = synth_tile(2, "sg_x").cuda(), synth_bits("sg_msg", (2, 1, 32, 32)).cuda()
    (m1, o1), (m2, o2) = make(), make()
    s1 = wm.TrainStep(m1, o1, 2, use_graph=True)
    s2 = wm.TrainStep(m2, o2, 2, use_graph=True, segmented=True)
    l1, l2 = [], []
    for _ in range(5):
        s1.step(x, msg)
        s2.step(x, msg)
        l1.append(s1.losses.cpu().clone())
        l2.append(s2.losses.cpu().clone())
    graphs, launches = s2._segments
    sends = sorted(r for sl in launches for r in sl)
    assert len(graphs) == len(launches) >= 10 and launches[-1] == []           # the last graph (AdamW) is followed by nothing
    assert sends[0][0] == 0 and sends[-1][1] == m2.optimized_numel
    assert all(a[1] == b[0] for a, b in zip(sends, sends[1:]))                 # disjoint, no gaps
    assert max(hi - lo for lo, hi in sends) * 4 < 45 << 20                     # ~25 MB buckets (whole blocks: up to ~38 MB)
    for a, b in zip(l1, l2):
        assert torch.allclose(a[:5], b[:5], rtol=2e-3, atol=1e-5), (a, b)
    assert not torch.equal(l2[0], l2[-1])
    assert rel(m1._flat[:m1.optimized_numel], m2._flat[:m2.optimized_numel].cpu()) < 1e-3
    assert int(o2.step_t.item()) == 5


def test_resume_from_reference_format_checkpoint(wm, tmp_path):
    """A checkpoint in the reference's format (torch AdamW state after one oracle step, DataParallel key prefix) is
    loaded into the model + fused optimizer; the next step matches the oracle's next step (nonzero moments, step = 2)."""
    from collections import OrderedDict
    ref = load_synth(O.UformerAudio()).train()
    opt = torch.optim.AdamW(ref.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    x, msg = synth_tile(2, "rs_x"), synth_bits("rs_msg", (2, 1, 32, 32))

    def ref_step():
        opt.zero_grad()
        loss, _ = O.train_losses(ref(x, msg), x, msg)
        loss.backward()
        opt.step()
        return loss.item()
    ref_step()
    path = wm.checkpoint.save_checkpoint(str(tmp_path), {
        "epoch": 1, "optimizer": opt.state_dict(),
        "state_dict": OrderedDict(("module." + k, v) for k, v in ref.state_dict().items())}, "resume")
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.0).cuda().train()
    m.compute_mode = "fp32"
    fused = wm.FusedAdamW(m, lr=1e-3)
    wm.checkpoint.load_checkpoint(m, path)
    assert wm.checkpoint.load_optim(fused, path) == 2e-4
    step = wm.TrainStep(m, fused, 2, use_graph=False)
    loss2 = ref_step()
    step.step(x.cuda(), msg.cuda())
    assert abs(step.losses[4].item() - loss2) <= 1e-4 * abs(loss2)
    assert int(fused.step_t.item()) == 2
    # Adam normalises the update to ~lr whatever the gradient's size, so an element whose gradient is a rounding-level
    # cancellation can move by +-lr on either side: bound every element by the two steps' worth of lr and ask all but a
    # sliver of them to agree tightly
    bad = total = 0
    for k, v in ref.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        d = (m.state_dict()[k].cpu().double() - v.double()).abs()
        assert d.max().item() <= 2 * 2 * 2e-4 + 1e-6, (k, d.max().item())
        bad += int((d > 1e-4 * v.double().abs() + 5e-6).sum())
        total += v.numel()
    assert bad <= 1e-2 * total, (bad, total)                   # (a wrong moment mapping moves every element)


# ------------------------------------------------------------------------------------------------------------------
# round 2: the benched mode (bf16) and the benched configurations (3 and 5) against the oracle
# ------------------------------------------------------------------------------------------------------------------
def test_reference_lr_schedule_drives_fused_adamw_under_graph_replay(wm):
    """the reference trainer's schedule (audio_uformer_stft.py:347-356: CosineAnnealingLR behind a
    GradualWarmupScheduler, stepped once per epoch, :545) on FusedAdamW while TrainStep REPLAYS its captured graph: the learning
    rate lives in device memory, so a replay must pick each epoch's value up.  A stock torch.optim.AdamW under the same two
    schedulers is fed the gradients the HIP step produced; the parameter arenas must agree after every step of five epochs."""
    import warnings
    tr = importlib.import_module("image-in-speech-watermarking_amd.train")
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode="bf16")
    load_synth(m)
    m = m.cuda().train()
    n = m.optimized_numel
    hyper = dict(lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)          # audio_uformer_stft.py:252-255, options.py
    opt = wm.FusedAdamW(m, **hyper)
    shadow = torch.nn.Parameter(m._flat[:n].detach().clone())
    ref = torch.optim.AdamW([shadow], **hyper)
    nepoch, warm, per_epoch = 5, 2, 2

    def schedule(o):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cosine = torch.optim.lr_scheduler.CosineAnnealingLR(o, nepoch - warm, eta_min=1e-6)
            sch = tr.GradualWarmupScheduler(o, multiplier=1, total_epoch=warm, after_scheduler=cosine)
            sch.step()
        return sch
    s_hip, s_ref = schedule(opt), schedule(ref)
    st = wm.TrainStep(m, opt, 1, use_graph=True)
    x, msg = synth_tile(1, "lr_x").cuda(), synth_bits("lr_msg", (1, 1, 32, 32)).cuda()
    lrs = []
    for epoch in range(nepoch):
        assert opt.param_groups[0]["lr"] == pytest.approx(ref.param_groups[0]["lr"], rel=1e-12)
        lrs.append(opt.param_groups[0]["lr"])
        for _ in range(per_epoch):
            before = m._flat[:n].detach().clone()
            st.step(x, msg)                                         # call 1: eager warm-up pass + capture; afterwards graph replays
            torch.cuda.synchronize()
            shadow.grad = m._flat_grad[:n].detach().clone()
            ref.step()
            moved = (m._flat[:n] - before).abs().mean().item()
            err = (m._flat[:n] - shadow.detach()).abs().max().item()
            # (AdamW moves an element by lr * m / (sqrt(v) + eps): ~lr at the first step, a fraction of it once the gradient signs
            # fluctuate; a replay that kept an OLD learning rate would be off by the difference of two epochs' rates, >= 3e-5)
            assert moved > 0.05 * lrs[-1] and err <= 2e-6, (epoch, lrs[-1], moved, err)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s_hip.step()
            s_ref.step()
    assert st._graph is not None                                    # the steps above were replays of ONE captured graph
    assert lrs[0] == pytest.approx(1e-4) and lrs[1] == pytest.approx(2e-4) and len(set(lrs)) >= 4 and lrs[-1] < lrs[2]
    assert float(opt.lr_t.item()) == pytest.approx(lrs[-1], rel=1e-6)


def _record(name, payload):
    """Measured deviations go to gpurun_out/ (merged back from the GPU box; the judged copies live under profiles/)."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            json.dump(payload, fh, indent=1)
    except OSError:
        pass


def _bucket_names(net):
    a, b, c = net._seg_sizes[:3]
    return (("encoder+payload AE", 0, a), ("decoder", a, a + b), ("extractor", a + b, a + b + c))


def test_bf16_train_step_gradients_against_fp32_mode(wm):
    """The benched path: one un-captured TrainStep in bf16 mode and one in fp32 parity mode from the same weights
    (drop_path_rate = 0): loss terms and every gradient bucket of the flat arena, norm-wise."""
    x, msg = synth_tile(2, "bf_x").cuda(), synth_bits("bf_msg", (2, 1, 32, 32)).cuda()
    grads, losses = {}, {}
    for mode in ("fp32", "bf16"):
        m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode=mode)
        load_synth(m)
        m = m.cuda().train()
        st = wm.TrainStep(m, wm.FusedAdamW(m, lr=0.0, weight_decay=0.0), 2, use_graph=False)
        st.step(x, msg)
        torch.cuda.synchronize()
        grads[mode], losses[mode] = m._flat_grad.double().cpu(), st.losses.double().cpu()
        names = _bucket_names(m)
    rec = {"loss_terms_rel": ((losses["bf16"][:5] - losses["fp32"][:5]).abs() / losses["fp32"][:5].abs()).tolist(), "buckets": {}}
    for name, lo, hi in names:
        a, b = grads["bf16"][lo:hi], grads["fp32"][lo:hi]
        rec["buckets"][name] = {"norm_rel_err": ((a - b).norm() / b.norm()).item(), "cosine": (a @ b / (a.norm() * b.norm())).item()}
    _record("r02_parity_bf16_trainstep_b2.json", rec)
    # measured on MI355X (profiles/r02_parity_bf16_trainstep_b2.json): loss terms <= 2.4e-3, buckets <= 2.1e-2 / cosine >= 0.9997
    assert max(rec["loss_terms_rel"]) <= 1e-2, rec
    for name, e in rec["buckets"].items():
        assert e["norm_rel_err"] <= 4e-2 and e["cosine"] >= 0.999, (name, e)


def test_bf16_droppath_backward_against_fp32_mode(wm):
    """the bf16 DropPath backward (the benched default: bf16, drop_path_rate = 0.1) had no test.  In bf16 mode
    layernorm_bwd writes the gradient copies pre-multiplied by the DropPath factor of the branch that reads them (dx1 * s1; dx * the
    consumer block's s2) and the fc2 / proj dX and dW launches then take NO scale; a wrong or stale factor still trains.  Here both
    modes run the reference-style loop with the SAME keep masks (some samples dropped in every block, some kept) and every gradient
    bucket is compared norm-wise; the chains block -> down -> block and up -> block (where the copy is reset) are part of the net."""
    B = 3
    x, msg = synth_tile(B, "dp_x").cuda(), synth_bits("dp_msg", (B, 1, 32, 32)).cuda()
    grads, terms = {}, {}
    for mode in ("fp32", "bf16"):
        m = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode=mode)
        load_synth(m)
        m = m.cuda().train()
        m._ensure_runtime(torch.device("cuda"))
        rates = m._rates_host
        scales = torch.ones(len(rates), B)
        for s_, r in enumerate(rates):
            if r > 0:
                for b in range(B):
                    scales[s_, b] = 0.0 if (5 * s_ + 3 * b) % 4 == 0 else 1.0 / (1.0 - r)
        assert (scales == 0).any(dim=1).sum() >= 60 and (scales > 1).any(dim=1).sum() >= 100     # every stage drops and keeps
        m.zero_grad(set_to_none=True)
        out = m(x, msg, keep_scales=scales.cuda())
        tot, tl = O.train_losses(out, x, msg)
        tot.backward()
        torch.cuda.synchronize()
        grads[mode], terms[mode] = m._flat_grad.double().cpu(), torch.stack(tl).detach().double().cpu()
        names = _bucket_names(m)
    rec = {"loss_terms_rel": ((terms["bf16"] - terms["fp32"]).abs() / terms["fp32"].abs()).tolist(), "buckets": {}}
    for name, lo, hi in names:
        a, b = grads["bf16"][lo:hi], grads["fp32"][lo:hi]
        rec["buckets"][name] = {"norm_rel_err": ((a - b).norm() / b.norm()).item(), "cosine": (a @ b / (a.norm() * b.norm())).item()}
    _record("r04_parity_bf16_droppath_b3.json", rec)
    # measured on MI355X (profiles/r04_parity_bf16_droppath_b3.json): encoder 1.4e-2, decoder 8e-3, extractor 6.2e-2 / cosine 0.9981
    # (the extractor sees the bf16 deviation of the whole embedder through its INPUT; without DropPath it is 3.2e-2 / 0.9995, with
    # the fused LeFF kernels switched off the same 6.0e-2 / 0.9982): bf16-vs-fp32 bounds cannot see a mis-placed factor in one
    # block, the additivity test below can
    assert max(rec["loss_terms_rel"]) <= 1e-2, rec
    for name, e in rec["buckets"].items():
        assert e["norm_rel_err"] <= 9e-2 and e["cosine"] >= 0.997, (name, e)


def test_bf16_droppath_backward_is_additive_over_samples(wm):
    """Sharp form of the test above.  With a loss that is LINEAR in the outputs (sum of outputs x fixed weights), the gradient of a
    batch is the sum of its samples' gradients, and in bf16 mode every per-row operation rounds the same way alone or in a batch.
    So a batched backward with per-sample DropPath factors must equal the three single-sample backwards with their own factor
    columns up to fp32 summation order — a factor applied to the wrong sample, applied twice, or missing from one of the scaled
    bf16 gradient copies (layernorm_bwd copy_scale, the consumer block's s2, the reset behind down / up-sampling) breaks it at
    the 1e-1 level in that block's parameters."""
    B = 3
    x, msg = synth_tile(B, "da_x").cuda(), synth_bits("da_msg", (B, 1, 32, 32)).cuda()
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.1, compute_mode="bf16")
    load_synth(m)
    m = m.cuda().train()
    m._ensure_runtime(torch.device("cuda"))
    rates = m._rates_host
    scales = torch.ones(len(rates), B)
    for s_, r in enumerate(rates):
        if r > 0:
            for b in range(B):
                scales[s_, b] = 0.0 if (5 * s_ + 3 * b) % 4 == 0 else 1.0 / (1.0 - r)
    scales = scales.cuda()
    gen = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(B, 2, 128, 128), (B, 2, 128, 128), (B, 1, 32, 32), (B, 1, 32, 32)]
    G = [torch.randn(sh, device="cuda", generator=gen) for sh in shapes]

    def grads(sl):
        m.zero_grad(set_to_none=True)
        out = m(x[sl], msg[sl], keep_scales=scales[:, sl].contiguous())
        sum((o * g[sl]).sum() for o, g in zip(out, G)).backward()
        torch.cuda.synchronize()
        return m._flat_grad.double().cpu()
    whole = grads(slice(0, B))
    parts = sum(grads(slice(b, b + 1)) for b in range(B))
    rec = {}
    for name, lo, hi in _bucket_names(m):
        a, b = whole[lo:hi], parts[lo:hi]
        rec[name] = ((a - b).norm() / b.norm()).item()
    # per owner (LeWin block / convolution): the largest relative deviation of any one block's parameters
    worst = max((((whole[lo:hi] - parts[lo:hi]).norm() / parts[lo:hi].norm().clamp_min(1e-30)).item(), name)
                for name, (lo, hi) in m._owner_range.items())
    rec["worst_owner"] = worst
    _record("r04_droppath_additivity_b3.json", rec)
    assert all(v <= 2e-3 for k, v in rec.items() if k != "worst_owner"), rec
    assert worst[0] <= 1e-2, worst


def test_bf16_whole_model_b6_against_cpu_oracle(wm):
    """bf16 mode at B = 6 — enough rows for the 128x128 and N = 192 GEMM instances and the multi-tile strip kernels the
    bench uses — eval forward and train-mode (drop_path_rate = 0) gradients against the fp32 CPU oracle on the same
    weights; deviations and BER are recorded (profiles/r02_parity_bf16_b6.json is the committed copy)."""
    B = 6
    x, msg = synth_tile(B, "b6_x"), synth_bits("b6_msg", (B, 1, 32, 32))
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode="bf16")
    load_synth(m)
    m = m.cuda().eval()
    ref = load_synth(O.UformerAudio(drop_path_rate=0.0)).eval()
    with torch.no_grad():
        out = [o.float().cpu() for o in m(x.cuda(), msg.cuda())]
        rout = ref(x, msg)
    keys = ("stft_new", "noise", "wm_pred", "wm")
    rec = {"batch": B, "eval_max_rel_dev": {k: rel(o, r) for k, o, r in zip(keys, out, rout)}}
    bits_h, bits_r = out[3].round().clamp(0, 1), rout[3].round().clamp(0, 1)
    rec["ber_hip_bf16"], rec["ber_cpu_fp32"] = float((bits_h - msg).abs().mean()), float((bits_r - msg).abs().mean())
    rec["payload_bits_differing"] = float((bits_h != bits_r).float().mean())
    lh, lr = O.train_losses(out, x, msg)[1], O.train_losses(rout, x, msg)[1]
    rec["loss_terms_rel"] = [abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(lh, lr)]
    # gradients
    m.train()
    ref.train()
    m.zero_grad(set_to_none=True)
    o2 = m(x.cuda(), msg.cuda())
    O.train_losses(o2, x.cuda(), msg.cuda())[0].backward()
    O.train_losses(ref(x, msg), x, msg)[0].backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}
    rg = {n: p.grad.detach().double() for n, p in ref.named_parameters() if p.grad is not None}
    worst = {}
    for pre in ("encoder.", "decoder.", "decoder_wm.", "encoder_wm.", "input_proj.", "output_proj.", "stft_layer."):
        a = torch.cat([g[n].flatten() for n in rg if n.startswith(pre)])
        b = torch.cat([rg[n].flatten() for n in rg if n.startswith(pre)])
        worst[pre] = {"norm_rel_err": ((a - b).norm() / b.norm()).item(), "cosine": (a @ b / (a.norm() * b.norm())).item()}
    rec["grad_groups"] = worst
    _record("r02_parity_bf16_b6.json", rec)
    # measured on MI355X (profiles/r02_parity_bf16_b6.json): tensors <= 1.2e-2, 0.15 % of the (untrained, near-0.5) payload
    # bits differ, loss terms <= 1.4e-2, gradient groups <= 1.9e-2 / cosine >= 0.9998
    assert max(rec["eval_max_rel_dev"].values()) <= 3e-2, rec
    assert rec["payload_bits_differing"] <= 0.005 and abs(rec["ber_hip_bf16"] - rec["ber_cpu_fp32"]) <= 0.005, rec
    assert max(rec["loss_terms_rel"]) <= 3e-2, rec
    for pre, e in worst.items():
        assert e["norm_rel_err"] <= 4e-2 and e["cosine"] >= 0.999, (pre, e)


def _full_attack_oracle(wm, net, B):
    """The oracle's differentiable attack chain with the HIP kernels' own AWGN realisation for the CURRENT step."""
    from oracle import attacks_torch as AT
    cfg = wm.AttackConfig.full_stack()
    z = wm.ops.awgn_noise(wm.attacks.rank_seed(cfg.seed), net._step_t, B, 8002).cpu()
    return cfg, (lambda w: AT.attack_chain(w.double(), resample=True, snr_db=20.0, z=z.double(), lowpass=True, amp=0.8).float())


@pytest.mark.parametrize("pmode", PARITY_MODES)
def test_config3_full_attack_stack_forward_and_gradients(wm, pmode):
    """BASELINE config 3: resample + AWGN (20 dB, replayed) + low-pass + x0.8 between iSTFT and STFT, extractor on the
    attacked spectrogram; train mode (drop_path_rate = 0), fp32 parity mode: outputs, loss terms and gradients of
    parameters on both sides of the attack against the oracle (autograd through its torch restatement of the chain)."""
    B = 2
    x, msg = synth_tile(B, "c3_x"), synth_bits("c3_msg", (B, 1, 32, 32))
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode=pmode)
    load_synth(m)
    m = m.cuda().train()
    m._ensure_runtime(torch.device("cuda"))
    wm.ops.step_inc(m._step_t)                                   # forward() advances the counter once more: replay step 2
    ref = load_synth(O.UformerAudio(drop_path_rate=0.0)).train()
    m.attack, m.extract_from = wm.AttackConfig.full_stack(), "attacked"
    m.zero_grad(set_to_none=True)
    out = m(x.cuda(), msg.cuda())
    cfg, attack = _full_attack_oracle(wm, m, B)                  # after the forward: the counter now holds the step it used
    loss, terms = O.train_losses(out, x.cuda(), msg.cuda())
    loss.backward()
    rout = ref(x, msg, attack=attack, extract_from="attacked")
    rloss, rterms = O.train_losses(rout, x, msg)
    rloss.backward()
    for o, r, k in zip(out, rout, ("stft_new", "noise", "wm_pred", "wm")):
        assert rel(o, r.detach()) <= 2e-4, (k, rel(o, r.detach()))
    np.testing.assert_allclose(torch.stack(terms).detach().cpu().numpy(), torch.stack(rterms).detach().numpy(), rtol=2e-4)
    assert torch.equal(out[3].detach().round().clamp(0, 1).cpu(), rout[3].detach().round().clamp(0, 1))
    names = ["decoder_wm.encoderlayer_0.blocks.0.attn.qkv.to_q.weight", "decoder_wm.conv2.weight", "stft_layer.0.weight",
             "output_proj.proj.0.weight", "decoder.decoderlayer_3.blocks.0.mlp.linear2.0.weight",
             "decoder.decoderlayer_0.blocks.3.attn.relative_position_bias_table", "decoder.upsample_0.deconv.0.weight",
             "encoder.encoderlayer_2.blocks.5.mlp.dwconv.0.weight", "encoder.dowsample_1.conv.0.weight",
             "encoder.conv.blocks.1.norm1.weight", "input_proj.proj.0.weight", "encoder_wm.conv1.weight"]
    for n in names:
        e = rel(m.get_parameter(n).grad, ref.get_parameter(n).grad)
        assert e <= 3e-3, (n, e)


_C5_ORACLE = {}


@pytest.mark.parametrize("pmode", PARITY_MODES)
def test_config5_ten_tile_clips_quadrant_payload(wm, pmode):
    """BASELINE config 5 / row A14: 2 clips x 10 tiles (10 s at 16 kHz through the device data-prep STFT), a 64 x 64
    grayscale payload per clip cut into quadrants over the tiles, full attack stack (AWGN replayed), train mode, fp32
    parity mode, against the oracle: outputs, loss, selected gradients; decoded quadrants re-assembled bit-exactly."""
    from synth import chirp_wave
    clips, tpc = 2, 10
    B = clips * tpc
    waves = torch.stack([chirp_wave(10 * 16384 - 128, f"c5_{i}") for i in range(clips)]).cuda()
    tiles, n_per = wm.dataprep.stft_tiles(waves)
    x = tiles.view(clips, n_per, 2, 128, 128)[:, :tpc].reshape(B, 2, 128, 128).contiguous()
    img = torch.rand(clips, 1, 64, 64, generator=torch.Generator().manual_seed(5)).cuda()
    msg = wm.dataprep.payload_for_tiles(img, tpc).contiguous()
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, compute_mode=pmode)
    load_synth(m)
    m = m.cuda().train()
    m.attack, m.extract_from = wm.AttackConfig.full_stack(), "attacked"
    m.zero_grad(set_to_none=True)
    out = m(x, msg)
    cfg, attack = _full_attack_oracle(wm, m, B)
    loss, terms = O.train_losses(out, x, msg)
    loss.backward()
    names = ("decoder_wm.conv.blocks.0.mlp.linear1.0.weight", "decoder.decoderlayer_2.blocks.0.attn.proj.weight",
             "encoder.encoderlayer_0.blocks.0.norm2.bias", "encoder_wm.t_conv2.weight")
    # the oracle's pass over the 20 tiles (a minute of CPU) is the same for both parity modes — same tiles, same payload, same
    # replayed AWGN stream (seed, step 1, B) — and is computed once
    key = (x.cpu().double().sum().item(), msg.cpu().double().sum().item(), int(m._step_t.item()))
    if _C5_ORACLE.get("key") != key:
        ref = load_synth(O.UformerAudio(drop_path_rate=0.0)).train()
        rout = ref(x.cpu(), msg.cpu(), attack=attack, extract_from="attacked")
        rloss, rterms = O.train_losses(rout, x.cpu(), msg.cpu())
        rloss.backward()
        _C5_ORACLE.update(key=key, rout=[r.detach() for r in rout], rterms=torch.stack(rterms).detach(),
                          grads={n: ref.get_parameter(n).grad.clone() for n in names})
    rout, rterms, rgrads = _C5_ORACLE["rout"], _C5_ORACLE["rterms"], _C5_ORACLE["grads"]
    for o, r, k in zip(out, rout, ("stft_new", "noise", "wm_pred", "wm")):
        assert rel(o, r) <= 2e-4, (k, rel(o, r))
    np.testing.assert_allclose(torch.stack(terms).detach().cpu().numpy(), rterms.numpy(), rtol=2e-4)
    for n in names:
        e = rel(m.get_parameter(n).grad, rgrads[n])
        assert e <= 3e-3, (n, e)
    # payload: per-tile quadrants -> 64 x 64 image, thresholded (grayscale payload: compare the rounded images)
    got = wm.dataprep.payload_from_tiles(out[3].detach().round().clamp(0, 1), tpc).cpu()
    want = wm.dataprep.payload_from_tiles(rout[3].round().clamp(0, 1).cuda(), tpc).cpu()
    assert torch.equal(got, want)


def test_training_forward_advances_the_random_streams(wm):
    """Reference-style loop (net(x, msg) in train mode, no TrainStep): consecutive forwards draw different DropPath masks
    and AWGN realisations (the Philox step counter used to advance only inside TrainStep)."""
    m = wm.UformerAudio(modulator=True, drop_path_rate=0.3)
    load_synth(m)
    m = m.cuda().train()
    x, msg = synth_tile(4, "rs_x").cuda(), synth_bits("rs_m", (4, 1, 32, 32)).cuda()
    with torch.no_grad():
        a = m(x, msg)[1].clone()
        b = m(x, msg)[1].clone()
        assert not torch.equal(a, b)
        m.eval()
        c, d = m(x, msg)[1].clone(), m(x, msg)[1].clone()
        assert torch.equal(c, d)                                 # eval: no stochastic depth, nothing advances
        m.train()
        m.arch.dp_rate = 0.0
    m2 = wm.UformerAudio(modulator=True, drop_path_rate=0.0, attack=wm.AttackConfig(snr_db=10.0))
    load_synth(m2)
    m2 = m2.cuda().train()
    with torch.no_grad():
        e, f = m2(x, msg)[0].clone(), m2(x, msg)[0].clone()
    assert not torch.equal(e, f)                                 # AWGN realisation differs per step


def test_backward_replays_its_own_forwards_noise_draw(wm):
    """Two training forwards, then their two backwards (gradient accumulation over two graphs, or net(x2) before
    loss1.backward()): every backward must regenerate the AWGN / jitter draws of ITS forward, not of the latest one.  The sum
    of the two gradients equals the sum from the interleaved order forward-backward-forward-backward."""
    cfg = wm.AttackConfig(snr_db=10.0, jitter=200, amp=0.9)

    def grads(interleaved):
        m = wm.UformerAudio(modulator=True, drop_path_rate=0.0, attack=cfg, extract_from="attacked", compute_mode="fp32")
        load_synth(m)
        m = m.cuda().train()
        x1, m1 = synth_tile(2, "nd_x1").cuda(), synth_bits("nd_m1", (2, 1, 32, 32)).cuda()
        x2, m2 = synth_tile(2, "nd_x2").cuda(), synth_bits("nd_m2", (2, 1, 32, 32)).cuda()
        m._flat_grad.zero_()
        o1 = m(x1, m1)
        l1 = ((o1[0] - x1) ** 2).mean() + ((o1[3] - m1) ** 2).mean()
        if interleaved:
            l1.backward()
        o2 = m(x2, m2)
        l2 = ((o2[0] - x2) ** 2).mean() + ((o2[3] - m2) ** 2).mean()
        if not interleaved:
            l1.backward()
        l2.backward()
        return m._flat_grad.clone(), o1[0].detach().clone(), o2[0].detach().clone()
    ga, a1, a2 = grads(True)
    gb, b1, b2 = grads(False)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)            # same draws in both orders
    err = ((ga - gb).norm() / ga.norm()).item()
    assert err <= 1e-5, err                                      # fp32 atomics reorder sums; a wrong draw gives O(1e-1)
