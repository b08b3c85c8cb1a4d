"""GPU: the operand ranges of the split-f16 kernels are LOUD (VERDICT r2 weak #5 / ADVICE r2).

At G = 64 the conv stack and the K-dominated linears run on the f16 matrix pipe with fp32 operands written as f16 hi + lo
halves under fixed power-of-two scalings (csrc/conv_split.h, csrc/linear.hip): relu(bn1(conv1)) <= 253.9, |W2| < 63.4,
fc inputs <= 1015, |W_fc| < 15.8.  Outside those ranges the kernels would clamp (or overflow to inf).  Contract tested here,
per violation: either the result is EXACT (the parameter pre-check of Hybrid_Encoder.check_operand_ranges moved the encoder to
the fp32-MFMA kernels before anything was computed) or GennbvHipError is raised (a kernel reached an activation bound) and a
repeat of the call is exact.  Reference = the same torch modules in fp64 on the CPU, tolerances of tests/test_encoder_gpu.py."""
import numpy as np
import pytest
import torch

from tests import policy_util as pu
from tests.test_encoder_gpu import _obs, _obs_clear_of_the_relu_threshold

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G, B = 64, 4


def _policies(mutate):
    hip, _, _ = pu.make_policy(g=G, device=DEV, backend="hip", det_weights=True)
    ref, _, _ = pu.make_policy(g=G, device="cpu", backend="torch", det_weights=True)
    with torch.no_grad():
        mutate(ref.features_extractor)
    hip.load_state_dict({k: v.to(DEV) for k, v in ref.state_dict().items()})
    ref = ref.double()
    ref.extract_features = lambda x: ref.features_extractor(x)
    return hip, ref


def _int8_batch(obs):
    from gennbv_amd.ops.encoder_ops import RowGather, input_autocorr
    grid_i8 = obs[:, 600:600 + G ** 3].to(torch.int8).contiguous()
    return RowGather(obs, torch.arange(obs.shape[0], device=DEV), grid_i8, None, input_autocorr(grid_i8, G))


def _forward_backward(pol, x, training=True):
    pol.set_training_mode(training)
    pol.zero_grad()
    f = pol.features_extractor(x)
    dev, dt = f.device, f.dtype
    (f * torch.linspace(0.5, 1.5, f.shape[1], device=dev, dtype=dt)).sum().backward()
    return f.detach().double().cpu(), {k: p.grad.detach().double().cpu() for k, p in pol.features_extractor.named_parameters() if p.grad is not None}


def _assert_close(got, ref, what, rel=2e-5):
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert np.isfinite(err) and err <= rel * scale + 1e-6, (what, err, scale)


def _compare(hip, ref, obs, training=True):
    if not training:  # inference (rollout): forward only -- the kernels' backward is the train-mode BatchNorm backward
        for pol in (hip, ref):
            pol.set_training_mode(False)
        with torch.no_grad():
            _assert_close(hip.features_extractor(_int8_batch(obs)).double().cpu(), ref.features_extractor(obs.cpu().double()), "features (eval)")
        return
    f_r, g_r = _forward_backward(ref, obs.cpu().double(), training)
    f_h, g_h = _forward_backward(hip, _int8_batch(obs), training)
    _assert_close(f_h, f_r, "features")
    for k, r in g_r.items():
        if float(r.abs().max()) > 1e-9:  # (conv biases in front of a train-mode BatchNorm: analytically zero)
            _assert_close(g_h[k], r, k)


def test_default_weights_stay_on_the_split_kernels():
    hip, ref = _policies(lambda enc: None)
    enc = hip.features_extractor
    info = enc.check_operand_ranges()
    assert not info["force_fp32"] and info["flag"] == 0 and info["w2_max"] < 1.0 and info["z1_bound"] < 60.0, info
    _compare(hip, ref, _obs_clear_of_the_relu_threshold(ref, B, G))
    assert not enc.check_operand_ranges()["force_fp32"]  # nothing raised a flag either


def test_w2_beyond_the_f16_scaling_runs_exact_on_the_fp32_kernels():
    """|W2| = 100: x 2^10 overflows f16 (inf / NaN in the split weight images); the pre-check selects the fp32-MFMA kernels."""
    def mutate(enc):
        w = enc.naive_encoder_grid[3].weight
        w.mul_(100.0 / float(w.abs().max()))
    hip, ref = _policies(mutate)
    enc = hip.features_extractor
    info = enc.check_operand_ranges()
    assert info["force_fp32"] and abs(info["w2_max"] - 100.0) < 1e-3 and enc.force_fp32
    obs = _obs_clear_of_the_relu_threshold(ref, B, G)
    _compare(hip, ref, obs)
    _compare(hip, ref, obs, training=False)


def test_fc_weight_beyond_its_clamp_runs_exact_on_the_fp32_kernels():
    """|W_fc| = 20 > 15.8 (fc_grid) -- the split linear kernels would clamp the weight."""
    def mutate(enc):
        w = enc.output_layer_grid[0].weight
        w[3, 17] = 20.0
        w[200, 5000] = -20.0
    hip, ref = _policies(mutate)
    enc = hip.features_extractor
    assert enc.check_operand_ranges()["force_fp32"] and getattr(enc.output_layer_grid[0], "_fp32_arith", False)
    _compare(hip, ref, _obs_clear_of_the_relu_threshold(ref, B, G))


def test_bn1_gamma_x300_precheck_and_device_flag():
    """BatchNorm-1 gamma x 300: relu(bn1(conv1)) reaches ~10^3 > 253.9.  (a) with the pre-check the encoder runs exact on the fp32
    kernels; (b) WITHOUT it (a caller that skips the check) the kernel that computes BN1's scale raises bit 2 of the range flag,
    the next check raises GennbvHipError, and the repeated call is exact."""
    from gennbv_amd._lib import GennbvHipError

    def mutate(enc):
        enc.naive_encoder_grid[1].weight.mul_(300.0)
    hip, ref = _policies(mutate)
    obs = _obs_clear_of_the_relu_threshold(ref, B, G)
    enc = hip.features_extractor
    info = enc.check_operand_ranges()
    assert info["force_fp32"] and info["z1_bound"] > 253.0
    _compare(hip, ref, obs)
    # (b) the split kernels run (nobody looked at the parameters) -> flag -> error -> exact repeat
    hip2, ref2 = _policies(mutate)
    enc2 = hip2.features_extractor
    assert not enc2.force_fp32
    _forward_backward(hip2, _int8_batch(obs))           # training mode: k_bn1_analytic sees the batch statistics
    assert int(enc2._range_flag.item()) & 2
    with pytest.raises(GennbvHipError):
        enc2.check_operand_ranges()
    assert enc2.force_fp32 and int(enc2._range_flag.item()) == 0
    hip2.load_state_dict({k: v.float().to(DEV) for k, v in ref2.state_dict().items()})  # (the flagged call advanced the running statistics)
    _compare(hip2, ref2, obs)
    # eval mode raises the same bit from the running statistics (k_bn_finalize)
    hip3, _ = _policies(mutate)
    hip3.set_training_mode(False)
    with torch.no_grad():
        hip3.features_extractor(_int8_batch(obs))
    assert int(hip3.features_extractor._range_flag.item()) & 2


def test_features_above_the_fc_input_clamp_raise():
    """BatchNorm-2 gamma x 3000: fc_grid's inputs exceed 1000 -> bit 4 of the flag (fc_grid's operand load, or k_bn_relu_apply when the fold is off) -> GennbvHipError; the
    repeat (fp32 linear kernel, no clamp) is exact."""
    from gennbv_amd._lib import GennbvHipError

    def mutate(enc):
        enc.naive_encoder_grid[4].weight.mul_(3000.0)
    hip, ref = _policies(mutate)
    obs = _obs_clear_of_the_relu_threshold(ref, B, G)
    enc = hip.features_extractor
    assert not enc.check_operand_ranges()["force_fp32"]  # no parameter is out of range: only the activations are
    _forward_backward(hip, _int8_batch(obs))
    assert int(enc._range_flag.item()) & 4
    with pytest.raises(GennbvHipError):
        enc.check_operand_ranges()
    assert enc.force_fp32
    hip.load_state_dict({k: v.float().to(DEV) for k, v in ref.state_dict().items()})  # (the flagged call advanced the running statistics)
    _compare(hip, ref, obs)


def test_train_call_checks_ranges_and_recaptures():
    """PPO_Grid_Obs.train(): parameters pushed out of range between two calls move the captured minibatch graph to the fp32
    kernels (re-capture) and the update still matches the fp64 loop (tests/test_ppo_g64_gpu.py's recording and oracle)."""
    from tests import test_ppo_g64_gpu as t64
    rec = t64._Recorded(n_envs=16, t=16, epochs=1)
    with torch.no_grad():
        w = rec.state["features_extractor.naive_encoder_grid.3.weight"]
        w.mul_(80.0 / float(w.abs().max()))
    ref = rec.oracle(None)
    hip = t64._fresh_hip(rec, None, True)
    hip.train()
    assert hip.policy.features_extractor.force_fp32 and hip._hip["force_fp32"] is True
    s_h, s_r = hip.last_train_stats, ref.last_train_stats
    assert len(s_h) == len(s_r) == 2
    d = np.abs(s_h[:, :6] - s_r[:, :6]) / np.maximum(1.0, np.abs(s_r[:, :6]))
    assert float(d.max()) <= 1e-4, d


def test_train_call_flagged_by_an_activation_is_replayed_on_the_fp32_kernels():
    """a range flag is only known AFTER train() applied every Adam step and BatchNorm update, so raising there aborts
    learn() with possibly clamped results applied.  Now train() snapshots its update state and REPEATS a flagged call on the fp32-MFMA
    kernels: BatchNorm-2 gamma x 400 puts fc_grid's inputs above 1000 (bit 4; no parameter is out of range, so the pre-check passes),
    the call must warn, finish, and equal the fp64 loop run ONCE from the same initial state."""
    from tests import test_ppo_g64_gpu as t64
    rec = t64._Recorded(n_envs=16, t=16, epochs=1)
    with torch.no_grad():
        rec.state["features_extractor.naive_encoder_grid.4.weight"].mul_(400.0)
    ref = rec.oracle(None)
    hip = t64._fresh_hip(rec, None, True)
    hip.policy.features_extractor.force_fp32 = False
    with pytest.warns(UserWarning, match="repeated on the"):
        hip.train()
    enc = hip.policy.features_extractor
    assert enc.force_fp32 and hip.range_replays == 1 and int(enc._range_flag.item()) == 0
    s_h, s_r = hip.last_train_stats, ref.last_train_stats
    assert len(s_h) == len(s_r) == 2 and int(hip._hip["opt"].step_count.item()) == 2 and hip._n_updates == ref._n_updates
    d = np.abs(s_h[:, :6] - s_r[:, :6]) / np.maximum(1.0, np.abs(s_r[:, :6]))
    assert float(d.max()) <= 1e-4, d
    sd_h = hip.policy.state_dict()
    for k, v in ref.policy.state_dict().items():
        if "running" in k:
            assert float((sd_h[k].double().cpu() - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max())), k
        if "num_batches" in k:
            assert int(sd_h[k]) == int(v), k  # the flagged pass's BatchNorm updates were rolled back
    hip.train()  # and the next call runs without a replay
    assert hip.range_replays == 1
