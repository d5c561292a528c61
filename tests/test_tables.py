"""Host-side table builders of the product (exported through the C ABI) vs the reference
fixtures and vs the oracle."""
import os

import numpy as np

import mel_spec_amd as M
from conftest import GOLDEN


def test_mel_matches_fixture_and_oracle(oracle):
    want = np.load(os.path.join(GOLDEN, "mel_filters.npz"))["mel_80"].astype(np.float64)
    got = M.mel(16000.0, 400, 80)
    assert got.shape == (80, 201)
    assert np.abs(got - want).max() <= 1e-7                      # src/mel.rs:838-850
    for (sr, n_fft, n_mels) in ((16000.0, 400, 80), (16000.0, 400, 128), (16000.0, 512, 80), (22050.0, 1024, 40)):
        assert np.abs(M.mel(sr, n_fft, n_mels) - oracle.mel_filterbank(sr, n_fft, n_mels)).max() <= 1e-12
    assert np.abs(M.mel(16000.0, 512, 64, 50.0, 7000.0, True, False)
                  - oracle.mel_filterbank(16000.0, 512, 64, 50.0, 7000.0, True, False)).max() <= 1e-12


def test_nemo_fixture():
    want = np.load(os.path.join(GOLDEN, "nemo_mel_filters.npz"))["banks"][0].astype(np.float64)
    assert np.abs(M.mel(16000.0, 512, 80) - want).max() <= 1e-7  # src/mel.rs:853-871


def test_sparse_structure_of_whisper_filters():
    # 391 non-zeros @80 mels, column 200 all-zero, contiguous support per row
    w = M.mel(16000.0, 400, 80)
    # column 200 (Nyquist) is dropped by project_stft_log10 (bin < n_fft/2, src/mel.rs:155-163);
    # in f64 it holds one ~2e-17 residue in the last row.
    assert int((w[:, :200] != 0).sum()) == 391 and np.abs(w[:, 200]).max() < 1e-15
    w = w[:, :200]
    for row in w:
        nz = np.nonzero(row)[0]
        assert nz.size and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))
    assert int((w != 0).sum()) < w.size // 10                    # src/mel.rs:910


def test_hann_and_kaldi_tables(oracle):
    assert np.array_equal(M.hann_window(400), oracle.hann_window(400))
    a = M.kaldi_mel_filterbank(16000.0, 512, 80, 20.0, 8000.0)
    b = oracle.kaldi_mel_filterbank(16000.0, 512, 80, 20.0, 8000.0)
    assert np.abs(a - b).max() <= 1e-12 and int((a != 0).sum()) == 501
