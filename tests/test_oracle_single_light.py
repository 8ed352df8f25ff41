"""The oracle's restatement of SingleLightSourceRayTracer (raytrace.rs:110-135), the Renderer's other RayTracer.

The reference never runs this tracer (it is constructed only in a comment, main.rs:153: light (14, 3, 3), intensity 1) and holds no
output of it: parity unpinned beyond what is checked here - the formula of raytrace.rs:119-126 recomputed in numpy for rays whose hit
is known in closed form, the emitter and the background branches, and powi(100) as compiler-rt's square-and-multiply."""
import numpy as np
import pytest

LIGHT, INTENSITY = (14.0, 3.0, 3.0), 1.0


def unit(a):
    return a / np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def powi(x, n):  # __powidf2
    a, r = x, 1.0
    while True:
        if n & 1:
            r *= a
        n //= 2
        if n == 0:
            return r
        a *= a


def expected(attenuation, p, normal, direction, light=LIGHT, intensity=INTENSITY):
    l = unit(np.array(light) - p)
    v = -unit(direction)
    h = unit(l + v)
    lambertian = np.array(attenuation) * intensity * max(float(np.dot(l, normal)), 0.0)
    blinn_phong = 0.5 * np.ones(3) * intensity * powi(max(float(np.dot(h, normal)), 0.0), 100)
    return lambertian + blinn_phong


@pytest.mark.parametrize("target,radius,albedo", [((0.0, 0.0, -1.0), 0.5, (0.1, 0.3, 0.5)), ((1.0, 0.0, -1.0), 0.5, (0.8, 0.6, 0.2))])
def test_lambert_and_blinn_phong_terms_of_a_known_hit(oracle, target, radius, albedo):
    """`simple` (worlds.rs:42-59): a ray aimed at the centre of the blue Lambertian sphere / the gold Metal sphere from a point from
    which nothing is in the way.  Hit point, normal and attenuation are known in closed form; the scattered ray is dropped."""
    w = oracle.world("simple", 42)
    for k, origin in enumerate([(0.3, 1.5, 1.0), (2.5, 0.7, 0.4), (0.9, 0.2, 2.0)]):
        o, c = np.array(origin), np.array(target)
        d = (c - o) * 1.7  # not normalised
        a, hb, cc = np.dot(d, d), np.dot(o - c, d), np.dot(o - c, o - c) - radius * radius
        t = (-hb - np.sqrt(hb * hb - a * cc)) / a
        p = o + t * d
        n = (p - c) / radius
        got = w.trace_ray_single_light(origin, d, LIGHT, INTENSITY, seed=7, pixel=11 + k, sample=3)
        assert np.allclose(got, expected(albedo, p, n, d), rtol=0, atol=1e-13), (got, expected(albedo, p, n, d))
    w.close()


def test_a_miss_is_the_background_and_an_emitter_is_its_emission(oracle):
    w = oracle.world("simple", 42)
    d = np.array([0.0, 1.0, 0.0])  # straight up: GradientBackground at t = 1 -> the top colour (raytrace.rs:29-35)
    assert np.allclose(w.trace_ray_single_light((0.0, 2.0, 5.0), d, LIGHT, INTENSITY, 1, 0, 0), [0.5, 0.7, 1.0])
    w.close()
    w = oracle.world("simple_light", 42)  # worlds.rs:227-239: a red DiffuseLight (7, 0, 0) sphere at (0, 6, 0), a green one on a rect at z = -2
    got = w.trace_ray_single_light((0.0, 6.0, 10.0), (0.0, 0.0, -1.0), LIGHT, INTENSITY, 1, 0, 0)
    assert np.array_equal(got, [7.0, 0.0, 0.0])  # scatter() is None: the emitter branch, no shading term
    got = w.trace_ray_single_light((30.0, 30.0, 30.0), (1.0, 1.0, 1.0), LIGHT, INTENSITY, 1, 0, 0)
    assert np.array_equal(got, [0.0, 0.0, 0.0])  # BlackBackground
    w.close()


def test_the_frame_loop_uses_the_tracer(oracle, mlrt):
    """render_single_light = the Renderer loop (raytrace.rs:172-198) around this tracer: a pixel's colour sum is the sum of its
    samples' traces (Philox stream per (pixel, sample), ray from Camera::get_ray), and the picture differs from the path tracer's."""
    w = oracle.world("simple", 42)
    pw = mlrt.World("simple", 42)
    W, H, spp = 48, 27, 3
    cam = mlrt.default_camera(pw, 16.0 / 9.0)
    cam19 = list(cam.origin) + list(cam.lower_left_corner) + list(cam.horizontal) + list(cam.vertical) + list(cam.u) + list(cam.v) + [cam.lens_radius]
    rgb, acc = w.render_single_light(cam19, W, H, spp, LIGHT, INTENSITY, seed=5, want_accum=True)
    again, _ = w.render_single_light(cam19, W, H, spp, LIGHT, INTENSITY, seed=5)
    assert np.array_equal(rgb, again)
    path, _, _, _ = w.render(cam19, W, H, spp, seed=5)
    assert not np.array_equal(rgb, path)
    assert acc.min() >= 0.0 and rgb.max() <= 255
    # twice the intensity: every shaded sample doubles, background samples do not (`simple` has no emitter)
    rgb2, acc2 = w.render_single_light(cam19, W, H, spp, LIGHT, 2.0 * INTENSITY, seed=5, want_accum=True)
    pixel_sky = np.all(acc2 == acc, axis=2)       # every sample of the pixel left the scene
    pixel_lit = np.all(acc2 == 2.0 * acc, axis=2)  # every sample of the pixel was shaded (powers of two scale exactly)
    assert pixel_sky.any() and pixel_lit.any() and np.all(acc2 >= acc)
    w.close(); pw.close()
