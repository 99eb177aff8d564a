"""The reference's own main.cpp, unmodified, links against librt_shim.so in
place of renderer.cpp + maths.cpp: g++ main.cpp -lrt_shim, nothing else changed
(main.cpp:104-106 calls render(), renderer.hpp:16-23 declares both functions).
Build-container test: needs /root/reference (absent on the GPU box) and g++.
Nothing is run: the run-time check of the shim's symbols is the GPU test
test_gpu_parity.py::test_reference_signature_shim."""
import os
import shutil
import subprocess

import pytest

import rtamd

REF = "/root/reference"
RENDER = "_Z6renderPhiPfS_iS0_S0_S_iS0_i"
CHECK = "_Z18check_intersectionPfiS_S_iS_PiS_S_"
HOST = os.path.join(rtamd.PKG_DIR, "host")


def test_shim_exports_the_reference_symbols():
    out = subprocess.check_output(["nm", "-D", "--defined-only",
                                   os.path.join(HOST, "librt_shim.so")], text=True)
    assert f" T {RENDER}" in out and f" T {CHECK}" in out


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "main.cpp")) or not shutil.which("g++"),
                    reason="needs the reference sources and g++ (build container only)")
def test_reference_main_links_against_the_shim(tmp_path):
    exe = tmp_path / "raytracer"
    cmd = ["g++", "-std=c++11", "-w", os.path.join(REF, "main.cpp"), "-o", str(exe),
           f"-L{HOST}", "-lrt_shim", f"-Wl,-rpath,{HOST}", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", str(exe)], text=True)
    assert RENDER in undefined, "main.cpp should take render() from the shim"
    needed = subprocess.check_output(["readelf", "-d", str(exe)], text=True)
    assert "librt_shim.so" in needed
    # nothing of the reference's renderer or math is in the executable
    defined = subprocess.check_output(["nm", "--defined-only", str(exe)], text=True)
    assert "raySphereIntersects" not in defined and CHECK not in defined
