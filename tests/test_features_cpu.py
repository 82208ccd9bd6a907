"""CPU checks of the feature-stage oracle: the reference's own smoke test (feature_extractor_test.py:24-34)
restated, and independent cross-checks (direct DFT, scipy)."""
import numpy as np
import torch

from oracle import features_oracle as FO


def test_reference_smoke_shapes():
    wav = np.zeros((4, 32000), np.float32)
    foa = FO.extract_features(wav, 16000, mode="foa")
    assert foa.ndim == 3 and foa.shape[-1] == 7 and foa.shape[1] == 64
    mic = FO.extract_features(wav, 16000, mode="mic")
    assert mic.ndim == 3 and mic.shape[-1] == 10
    # zeros -> power 0 -> dB floor 10*log10(1e-10) = -100 (top_db clamp is inactive), IV = 0
    assert np.all(foa[..., :4] == -100.0) and np.all(foa[..., 4:] == 0.0)


def test_stft_matches_direct_dft():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 2000))
    spec = FO.complex_spec(torch.tensor(x), n_fft=256, win_length=240, hop_length=120).numpy()
    assert spec.shape == (2, 129, 1 + 2000 // 120)
    win = np.zeros(256)
    win[8:248] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(240) / 240)
    xp = np.pad(x, ((0, 0), (128, 128)), mode="reflect")
    for t in (0, 5, 16):
        frame = xp[:, t * 120:t * 120 + 256] * win
        np.testing.assert_allclose(spec[:, :, t], np.fft.rfft(frame, axis=-1), atol=1e-10)


def test_mel_filterbank_properties():
    fb = FO.mel_filterbank(513, 64, 24000)
    assert fb.shape == (513, 64) and fb.min() >= 0 and fb.max() <= 1.0
    assert ((fb > 0).sum(1) <= 2).all()                       # a bin feeds at most two triangles
    for m in range(64):                                       # each triangle is one contiguous bin range
        nz = np.nonzero(fb[:, m])[0]
        assert nz.size > 0 and (np.diff(nz) == 1).all()


def test_gcc_is_irfft_of_unit_phase():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 4096))
    spec = FO.complex_spec(torch.tensor(x), n_fft=512)
    g = FO.gcc_features(spec, 64).numpy()
    assert g.shape == (6, 64, spec.shape[2])
    R = np.conj(spec[0].numpy()) * spec[1].numpy()
    cc = np.fft.irfft(R / np.abs(R), axis=0)
    np.testing.assert_allclose(g[0], np.concatenate([cc[-32:], cc[:32]], 0), atol=1e-10)
    assert np.abs(g).max() <= 1.0 + 1e-9


def test_stockham_model_matches_numpy_fft():
    rng = np.random.default_rng(2)
    for n in (64, 512, 1024):
        x = rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))
        np.testing.assert_allclose(FO.stockham_fft(x), np.fft.fft(x), atol=1e-10)


def test_radix4_stockham_model_matches_numpy_fft():
    """Index model of the wave-per-frame FFT (features.hip::wave_fft): in-place radix-4 passes + a final radix-2 pass for odd log2 n."""
    rng = np.random.default_rng(3)
    for n in (256, 512, 1024, 2048):
        x = rng.standard_normal((2, n)) + 1j * rng.standard_normal((2, n))
        np.testing.assert_allclose(FO.stockham_fft_radix4(x), np.fft.fft(x), atol=1e-10)


def test_pruned_inverse_transform_model_matches_irfft():
    """The mic kernel's GCC-PHAT inverse transform (two stages of 32-point sums, only the kept lags: features_oracle.gcc_pruned_inverse_model)
    against numpy's irfft on unit-modulus half spectra, as feature_extractor.gcc_features forms them (feature_extractor.py:196-214)."""
    rng = np.random.default_rng(0)
    for n_lags in (64, 40, 32):
        ph = rng.uniform(-np.pi, np.pi, 513)
        ph[0], ph[512] = 0.0, np.pi                          # real end bins (+1 and -1), as the spectra of real signals have
        R = np.exp(1j * ph)
        cc = np.fft.irfft(R, 1024)
        want = np.concatenate([cc[-n_lags // 2:], cc[:n_lags // 2]])
        got = FO.gcc_pruned_inverse_model(R, n_lags)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-13
