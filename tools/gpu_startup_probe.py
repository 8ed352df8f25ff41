"""Where the CLI's `Rendered in` goes on a small render: runs the C consumer-style sequence in-process with timestamps.
python tools/gpu_startup_probe.py"""
import ctypes as C, importlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.perf_counter()
pkg = importlib.import_module("mu-lambda-raytracer_amd")
L = pkg.lib()
t1 = time.perf_counter()
n = L.mlrt_device_count()
t2 = time.perf_counter()
w = pkg.World("random", 42)
cam = pkg.default_camera(w, 1.5)
params = pkg.make_params(w, 1200, 800, 500)
t3 = time.perf_counter()
r = pkg.Renderer(w, cam, params, precision=pkg.PRECISION_FP64)
t4 = time.perf_counter()
img = r.render()
t5 = time.perf_counter()
img = r.render()
t6 = time.perf_counter()
print("dlopen %.1f ms | first HIP call (device count) %.1f ms | world build %.1f ms | context create %.1f ms | first render %.1f ms | second render %.1f ms" % (
    (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3))
for args in (["--image_width=1200", "--samples_per_pixel=500"], ["--image_width=400", "--samples_per_pixel=8"]):
    for k in range(2):
        t = time.perf_counter()
        out = subprocess.run([pkg.CLI_PATH, "--world=random", "--seed=42", "--aspect_ratio=3:2", "--format=p6"] + args, capture_output=True)
        print(args, "process wall %.3f s |" % (time.perf_counter() - t), out.stderr.decode().strip().splitlines()[-1])
