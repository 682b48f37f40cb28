"""GPU tests of the large 2-D FFT (xh_fft2d_*, SURVEY.md section 8f rank 3: the transform FlexAlign's movie frames need)
against numpy.fft: sizes whose lines need the four-step form, and sizes that take one LDS transform per line."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import xmipp3_amd as xa
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return xa, xa.Context(0), torch


@pytest.mark.parametrize("ny,nx", [(64, 96), (100, 36), (2048, 384), (4096, 90), (90, 4096), (3000, 2304), (4096, 5760)])
def test_fft2d_against_numpy(gpu, ny, nx):
    """Forward transform within fp32 rounding of numpy's double transform (1e-5 of the largest coefficient: log2(N) = 24
    butterfly levels of 6e-8 each and the Bluestein lines' three transforms), inverse(forward(x)) = x to 2e-5 of max |x|.
    4096 x 5760 is a K3 movie frame (movie_alignment_correlation_gpu.cpp:633-725)."""
    xa, ctx, torch = gpu
    rng = np.random.default_rng(ny * 7 + nx)
    x = (rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))).astype(np.complex64)
    f = xa.Fft2D(ctx, ny, nx)
    big = max(ny, nx) > 2048 or (not (nx & (nx - 1)) == 0 and nx > 1024) or (not (ny & (ny - 1)) == 0 and ny > 1024)
    assert (f.factors[1] > 1 or f.factors[3] > 1) == big
    d = torch.from_numpy(x).cuda()
    f(d)
    got = d.cpu().numpy()
    exp = np.fft.fft2(x.astype(np.complex128))
    assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    f(d, inverse=True)
    back = d.cpu().numpy()
    assert np.abs(back - x).max() <= 2e-5 * np.abs(x).max()
    f.close()


def test_fft2d_refuses_a_length_it_cannot_split(gpu):
    xa, ctx, torch = gpu
    with pytest.raises(xa.XhError):
        xa.Fft2D(ctx, 64, 2 * 1031)        # 2062 = 2 x 1031 (prime): neither one LDS line nor two
