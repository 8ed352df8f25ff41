"""The host's baseline-JPEG reader (the format the earth texture arrives in, worlds.rs:180,441) against PIL/libjpeg on
small synthetic fixtures (tests/golden/jpeg_*.jpg, made with PIL from generated content)."""
import os

import numpy as np
import pytest

PIL_Image = pytest.importorskip("PIL.Image")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,chroma_tol", [("jpeg_444.jpg", 3), ("jpeg_gray.jpg", 2), ("jpeg_420.jpg", None), ("jpeg_422_restart.jpg", None)])
def test_decoder_matches_libjpeg(mlrt, name, chroma_tol):
    data = open(os.path.join(GOLD, name), "rb").read()
    mine = mlrt.decode_jpeg(data).astype(int)
    ref = np.asarray(PIL_Image.open(os.path.join(GOLD, name)).convert("RGB")).astype(int)
    assert mine.shape == ref.shape
    d = np.abs(mine - ref)
    if chroma_tol is not None:
        assert d.max() <= chroma_tol, d.max()  # full-resolution chroma: only IDCT / colour rounding differs
    else:
        # subsampled chroma is replicated here and interpolated by libjpeg: same picture, small local differences
        assert d.mean() < 2.5 and np.percentile(d, 99) <= 24, (d.mean(), np.percentile(d, 99))


def test_progressive_is_refused_with_a_message(mlrt):
    data = open(os.path.join(GOLD, "jpeg_progressive.jpg"), "rb").read()
    with pytest.raises(mlrt.MlrtError) as e:
        mlrt.decode_jpeg(data)
    assert e.value.code == mlrt.MLRT_ERR_UNSUPPORTED and "baseline" in str(e.value)
    with pytest.raises(mlrt.MlrtError):
        mlrt.decode_jpeg(b"not a jpeg at all")


@pytest.mark.skipif(not os.path.exists("/root/reference/earthmap.jpg"), reason="reference assets only exist in the build container")
def test_reference_earthmap_decodes_like_the_committed_fixture(mlrt, earthmap):
    """earthmap.jpg through the library's reader vs assets/earthmap.ppm (decoded by PIL/libjpeg in make_golden.py)."""
    mine = mlrt.decode_jpeg(open("/root/reference/earthmap.jpg", "rb").read()).astype(int)
    assert mine.shape == earthmap.shape
    assert np.abs(mine - earthmap.astype(int)).max() <= 3
