"""The device RNG contract differs from the reference in ONE documented place: uniform points in the unit ball / unit
disk are drawn by inverse CDF from exactly 3 / 2 Philox words instead of the reference's rejection loops
(vec.rs:23-30, 45-52).  The distributions must be the same; these tests check that on the samplers themselves and at
image level (oracle in device-contract mode vs oracle in the reference's own PCG64 + rejection mode)."""
import ctypes as C

import numpy as np
import pytest

from oracle_binding import RNG_PCG_ROW, RNG_PHILOX_SAMPLE


def _samples(oracle, fn, dim, mode, n, seed=7):
    out = np.zeros((n, dim))
    getattr(oracle.L, fn)(C.c_uint64(seed), mode, n, out.ctypes.data_as(C.c_void_p))
    return out


def _ks_uniform(u):
    u = np.sort(u)
    n = len(u)
    return max(np.max(np.arange(1, n + 1) / n - u), np.max(u - np.arange(0, n) / n))


def test_ball_samplers_have_the_same_distribution(oracle):
    n = 200_000
    crit = 1.95 / np.sqrt(n)  # KS critical value at alpha ~ 0.001
    for mode in (0, 1):
        p = _samples(oracle, "orc_sample_ball", 3, mode, n)
        r = np.linalg.norm(p, axis=1)
        assert r.max() <= 1.0 and r.min() > 0.0
        assert _ks_uniform(r ** 3) < crit                      # P(|p| < rho) = rho^3
        assert _ks_uniform((p[:, 2] / r + 1) / 2) < crit       # cos(theta) uniform: direction uniform on the sphere
        assert _ks_uniform((np.arctan2(p[:, 1], p[:, 0]) / (2 * np.pi)) % 1.0) < crit
        assert np.all(np.abs(p.mean(axis=0)) < 4 * np.sqrt(0.2 / n))          # E[x] = 0, Var[x] = 1/5
        assert np.all(np.abs((p ** 2).mean(axis=0) - 0.2) < 0.003)
        octants = np.bincount((p[:, 0] > 0) * 4 + (p[:, 1] > 0) * 2 + (p[:, 2] > 0), minlength=8)
        assert np.all(np.abs(octants - n / 8) < 5 * np.sqrt(n / 8))


def test_disk_samplers_have_the_same_distribution(oracle):
    n = 200_000
    crit = 1.95 / np.sqrt(n)
    for mode in (0, 1):
        p = _samples(oracle, "orc_sample_disk", 2, mode, n)
        r = np.linalg.norm(p, axis=1)
        assert r.max() <= 1.0
        assert _ks_uniform(r ** 2) < crit                      # P(|p| < rho) = rho^2
        assert _ks_uniform((np.arctan2(p[:, 1], p[:, 0]) / (2 * np.pi)) % 1.0) < crit
        assert np.all(np.abs(p.mean(axis=0)) < 4 * np.sqrt(0.25 / n))


def test_closed_form_uses_exactly_three_and_two_draws(oracle):
    # draw k of the stream (pixel 0, sample i) is word k: the ball sample must consume words 0..2 only
    u = oracle.philox_uniform(7, 0, 5, 3)
    p = _samples(oracle, "orc_sample_ball", 3, 1, 6)[5]
    z = u[0] * 2.0 - 1.0
    assert abs(p[2] - np.cbrt((round(u[2] * 2 ** 24) + 1) / 2 ** 24) * z) < 1e-15
    assert abs(np.arctan2(p[1], p[0]) % (2 * np.pi) - 2 * np.pi * u[1]) < 1e-12


@pytest.mark.parametrize("name,aspect,width,height,spp,lens", [("random", 1.5, 144, 96, 96, True), ("cornell_smoke", 1.0, 96, 96, 192, False),
                                                              ("final_scene", 1.0, 96, 96, 128, False)])
def test_device_contract_renders_the_same_picture_as_the_reference_algorithm(oracle, earthmap, name, aspect, width, height, spp, lens):
    """oracle[Philox streams + inverse-CDF samplers + one block per scatter] vs oracle[the reference's PCG64 row streams + rejection
    loops]: 8x8 block means of the linear radiance must agree as well as two PCG renders with different render seeds do.  `random`
    with a lens covers the disk sampler; cornell_smoke and final_scene cover the MEDIUM draws (volumes.rs:47), which the
    block-per-scatter rule of the device contract moves to other words of the stream (oracle_core.hpp begin_scatter / end_scatter)."""
    w = oracle.world(name, 42, earthmap if name == "final_scene" else None)
    lookfrom, lookat, fov = w.camera()
    focus = 10.0 if lens else float(np.linalg.norm(np.array(lookat) - np.array(lookfrom)))
    cam = oracle.camera_new(lookfrom, lookat, (0, 1, 0), fov, aspect, 0.1 if lens else 0.0, focus)
    W, H = width, height

    def blocks(mode, seed):
        _, acc, _, _ = w.render(cam, W, H, spp, seed=seed, rng_mode=mode, want_accum=True)
        lin = np.minimum(acc / spp, 4.0)
        return lin.reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))

    pcg_a, pcg_b, dev = blocks(RNG_PCG_ROW, 1), blocks(RNG_PCG_ROW, 2), blocks(RNG_PHILOX_SAMPLE, 1)
    noise = np.sqrt(np.mean((pcg_a - pcg_b) ** 2))
    diff = np.sqrt(np.mean((dev - pcg_a) ** 2))
    assert diff < 1.35 * noise, (name, diff, noise)
    assert abs(dev.mean() / pcg_a.mean() - 1.0) < (0.01 if name == "random" else 0.03), (name, dev.mean() / pcg_a.mean())
    w.close()


def test_contract_arithmetic_of_the_samplers_is_accurate_over_all_inputs(oracle):
    """cbrt24 and turn_cos_sin (oracle_core.hpp; the same lines in mlrt_kernel.hip) ARE the contract's cube root and cos / sin of a
    24-bit draw - defined as mul / fma sequences so that oracle and device agree to the bit.  Here: all 2^24 inputs of each
    against long-double libm, and exact values at the ends and on the axes."""
    out = (C.c_double * 10)()
    oracle.L.orc_contract_function_errors(out)
    assert out[0] < 6e-16, out[0]                      # cbrt24: relative
    assert out[1] < 8e-16 and out[2] < 8e-16, (out[1], out[2])   # cos, sin: absolute (measured 6.4e-16, 6.9e-16)
    assert out[3] == 2.0 ** -8 and out[4] == 1.0       # cbrt(2^-24), cbrt(1)
    assert (out[5], out[6]) == (1.0, 0.0) and (out[7], out[8]) == (0.0, 1.0)
