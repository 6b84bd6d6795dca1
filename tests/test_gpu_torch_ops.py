"""GPU: the PyTorch-ROCm custom ops torch.ops.asw.* (csrc/torch_ops.cpp) -- they run on the
tensor's device and torch's current stream, validate their arguments (RuntimeError, never a
silent truncation) and give the results of the C ABI / the oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asw():
    from acousticswarms_speech_amd import native
    return native.torch_ops()


def test_energies_and_pair_sisdr_ops(asw):
    from oracle import spot_ref
    y = torch.randn(5, 6000, generator=torch.Generator().manual_seed(1)) * 0.1
    en = asw.energies(y.cuda(), 1000)
    assert en.dtype == torch.float64 and en.shape == (5, 2)
    np.testing.assert_allclose(en.cpu().numpy(), spot_ref.candidate_energies(y.numpy(), 1000), rtol=1e-4)
    s = asw.pair_sisdr(y.cuda()).cpu().numpy()
    for i in range(5):
        for j in range(5):
            if i != j:
                assert abs(s[i, j] - spot_ref.si_sdr(y[i].numpy(), y[j].numpy())) < 1e-3
    z = y.cuda().clone()
    out = asw.center_rows_(z)
    assert out.data_ptr() == z.data_ptr() and float(z.mean(1).abs().max()) < 1e-7
    # on a side stream: the op must enqueue on torch's CURRENT stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        en2 = asw.energies(y.cuda(), 1000)
    side.synchronize()
    assert torch.equal(en2, en)


def test_spot_ops_match_the_ctypes_abi_and_the_oracle(asw):
    from acousticswarms_speech_amd.config import SMALL
    from acousticswarms_speech_amd.scenes import make_scene
    from acousticswarms_speech_amd.spot import SpotModel
    from acousticswarms_speech_amd.weights import make_spot_state_dict
    from oracle import spot_ref
    sd = make_spot_state_dict(SMALL, seed=3)
    model = SpotModel(SMALL, sd, batch_size=4).to("cuda")
    mix = torch.from_numpy(make_scene(7, 2, 7, 4000).mix)
    offs = np.array([[0, 0, 0, 0, 0, 0], [3, -5, 8, -13, 21, -34], [-40, 30, -20, 10, -5, 2]], dtype=np.int32)
    wave, en = asw.spot_shift_and_sep(model._h.value, mix.cuda(), torch.from_numpy(offs).cuda(), 1, True, True, True, 1000)
    want = spot_ref.shift_and_sep(sd, SMALL, mix, list(offs), strict=1)
    snr = 10 * np.log10(np.sum(want.astype(np.float64) ** 2) / np.sum((wave.cpu().numpy().astype(np.float64) - want) ** 2))
    assert snr > 80.0
    np.testing.assert_allclose(en.cpu().numpy(), spot_ref.candidate_energies(want, 1000), rtol=1e-4)
    # the class surface goes through the same op
    np.testing.assert_array_equal(model.shift_and_sep(mix, list(offs), Strict=1), wave.cpu().numpy())
    # fused front end as an op of its own
    w = torch.from_numpy(sd["preproc.weight"][:, :, 0]).cuda().contiguous()
    b = torch.from_numpy(sd["preproc.bias"]).cuda()
    x0, refn, mean, std = asw.shift_norm_preproc(mix.cuda(), torch.from_numpy(offs).cuda(), w, b, 4096, True)
    data = torch.stack([spot_ref.roll_channels(mix, o) for o in offs])
    dn, mu, sg = spot_ref.normalize_input(data)
    np.testing.assert_allclose(mean.cpu().numpy(), mu.flatten().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(std.cpu().numpy(), sg.flatten().numpy(), rtol=2e-6)
    assert x0.shape == (3, 4096, 64) and refn.shape == (3, 4096)
    np.testing.assert_allclose(refn[:, 96:].cpu().numpy(), dn[:, 0].numpy(), rtol=1e-5, atol=1e-6)


def test_ops_validate_their_arguments(asw):
    y = torch.zeros(3, 100, device="cuda")
    with pytest.raises(RuntimeError, match="Float"):
        asw.pair_sisdr(y.double())
    with pytest.raises(RuntimeError, match="contiguous"):
        asw.pair_sisdr(torch.zeros(100, 3, device="cuda").t())
    with pytest.raises(RuntimeError, match="dimensions"):
        asw.energies(torch.zeros(100, device="cuda"), 10)
    with pytest.raises(RuntimeError, match="null model"):
        asw.spot_shift_and_sep(0, torch.zeros(7, 100, device="cuda"), torch.zeros(1, 6, dtype=torch.int32, device="cuda"),
                               0, True, True, False, 10)
    with pytest.raises(RuntimeError, match="M-1"):
        asw.sep_infer(1, torch.zeros(7, 100, device="cuda"), torch.zeros(2, 5, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError, match="same device|HIP"):
        asw.segment_sisdr(y, torch.zeros(3, 1, 2, dtype=torch.int32), torch.zeros(3, dtype=torch.int32, device="cuda"))
