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


def test_truncated_and_lying_segments_fail_cleanly(mlrt, tmp_path):
    """Marker segments shorter than the fields they announce (a truncated file, or a segment length that lies) must be
    refused with a message - never read past the buffer (an ASAN build of the reader found exactly that in round 1) and
    never let an exception escape the C ABI (a header announcing 65535 x 65535 pixels)."""
    import subprocess
    cases = {
        "sof_len2": bytes.fromhex("FFD8 FFC0 0002"),                                   # SOF with no payload at all
        "sof_no_components": bytes.fromhex("FFD8 FFC0 0008 08 0010 0010 03"),           # announces 3 components, holds none
        "dqt_truncated": bytes.fromhex("FFD8 FFDB 0003 00"),                            # table id, no table
        "dqt_16bit_short": bytes.fromhex("FFD8 FFDB 0043 10") + bytes(64),              # 16-bit table needs 128 bytes
        "dht_truncated": bytes.fromhex("FFD8 FFC4 0005 00 0102"),                       # 2 of 16 counts
        "dri_short": bytes.fromhex("FFD8 FFDD 0003 00"),
        "sos_short": bytes.fromhex("FFD8 FFC0 000B 08 0008 0008 01 01 11 00 FFDA 0003 01"),
        "huge": bytes.fromhex("FFD8 FFC0 000B 08 FFFF FFFF 01 01 11 00 FFDA 0008 01 01 00 00 3F 00"),
    }
    # every prefix of a real file is a truncated file
    good = open(os.path.join(GOLD, "jpeg_444.jpg"), "rb").read()
    for cut in (3, 5, 20, 100, len(good) // 2):
        cases["prefix_%d" % cut] = good[:cut]
    # (a file cut inside the entropy-coded scan decodes, as with libjpeg: the missing bits read as zeros - what must not
    # happen is a read past the buffer, which the sanitizer run below would report)
    may_decode = {"prefix_%d" % (len(good) // 2)}
    for name, data in cases.items():
        if len(data) < 4:
            data = data + b"\x00" * (4 - len(data))
        try:
            img = mlrt.decode_jpeg(data)
            assert name in may_decode and img.shape == (70, 93, 3), name
        except mlrt.MlrtError as e:
            assert e.code in (mlrt.MLRT_ERR_UNSUPPORTED, mlrt.MLRT_ERR_INVALID), name
    # the same inputs through a host-only AddressSanitizer build of the reader (gcc; no GPU code involved)
    root = os.path.dirname(os.path.dirname(GOLD))
    exe = str(tmp_path / "jpeg_asan")
    main = tmp_path / "main.cpp"
    main.write_text('#include <cstdio>\n#include <cstdint>\n#include <string>\n#include "mlrt_host.h"\n'
                    'namespace mlrt { void set_last_error(const std::string&) {} }\n'
                    'int main(int argc, char** argv) { for (int i = 1; i < argc; i++) { uint8_t* px = nullptr; int32_t w = 0, h = 0; '
                    'int rc = mlrt_decode_jpeg_file(argv[i], &px, &w, &h); if (rc == 0) mlrt_free(px); std::printf("%d\\n", rc); } return 0; }\n')
    host = os.path.join(root, "mu-lambda-raytracer_amd", "host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(root, "include"), str(main),
                    os.path.join(host, "jpeg_baseline.cpp"), "-o", exe], check=True)
    paths = []
    for name, data in cases.items():
        q = tmp_path / (name + ".jpg")
        q.write_bytes(data)
        paths.append(str(q))
    out = subprocess.run([exe] + paths + [os.path.join(GOLD, "jpeg_444.jpg")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    codes = [int(x) for x in out.stdout.split()]
    names = list(cases)
    assert codes[-1] == 0 and all(c != 0 for c, n in zip(codes[:-1], names) if n not in may_decode), codes
